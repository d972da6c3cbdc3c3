"""Local-affine photo smoothing -- the reference's alternative post-process, /root/reference/smooth_filter.py
(dead code there: nothing on the swap path calls it).  Same two entry points and argument
meaning: ``smooth_local_affine(output, input, epsilon, patch, h, w, f_r, f_e)`` (:332-378) and
``smooth_filter(initImg, contentImg, f_radius=15, f_edge=1e-1)`` (:381-405); the arithmetic runs in
csrc/smooth_filter.hip (two launches instead of the reference's three NVRTC kernels)."""
import numpy as np
import torch

from . import ops
from ._lib import lib, check

_p, _stream = ops._p, ops._stream


def smooth_local_affine_tensor(output, input_, patch=3, f_r=15, f_e=1e-1, return_model=False):
    """output (stylised), input_ (content / guide): CUDA fp32 (B,3,H,W) or (3,H,W), planar, channel order as the reference
    passes them (BGR in, RGB out: the kernels index channels in reverse).  Returns the smoothed image, same shape."""
    for t, n in ((output, "output"), (input_, "input")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError("smooth_local_affine needs CUDA fp32 tensors (%s)" % n)
    squeeze = output.dim() == 3
    o = (output[None] if squeeze else output).contiguous()
    i = (input_[None] if input_.dim() == 3 else input_).contiguous()
    if o.shape != i.shape or o.dim() != 4 or o.shape[1] != 3:
        raise RuntimeError("smooth_local_affine: output %s and input %s must both be (B,3,H,W)" % (tuple(o.shape), tuple(i.shape)))
    B, _, H, W = o.shape
    res = torch.empty_like(o)
    ws = torch.empty(lib.ppst_smooth_local_affine_ws(B, H, W), device=o.device, dtype=torch.uint8)
    filt = torch.empty((B, H * W, 12), device=o.device, dtype=torch.float32) if return_model else None
    radius = int((patch - 1) / 2)
    check(lib.ppst_smooth_local_affine(_p(o), _p(i), _p(res), _p(ws), _p(filt), B, H, W, radius, int(f_r), float(f_r) / 3.0,
                                       float(f_e), _stream()), "ppst_smooth_local_affine")
    res = res[0] if squeeze else res
    if return_model:
        return res, (filt[0] if squeeze else filt), ws.view(torch.float32).view(B, H * W, 12)
    return res


def smooth_local_affine(output_cpu, input_cpu, epsilon, patch, h, w, f_r, f_e):
    """smooth_filter.py:332-378: numpy (3,h,w) float32 arrays in, numpy array out (``epsilon`` is accepted and unused, as in
    the reference's kernel)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    o = torch.from_numpy(np.ascontiguousarray(output_cpu, dtype=np.float32)).to(dev)
    i = torch.from_numpy(np.ascontiguousarray(input_cpu, dtype=np.float32)).to(dev)
    if tuple(o.shape) != (3, h, w):
        raise RuntimeError("smooth_local_affine: arrays are %s, h/w say (3,%d,%d)" % (tuple(o.shape), h, w))
    return smooth_local_affine_tensor(o, i, patch, f_r, f_e).cpu().numpy()


def smooth_filter(initImg, contentImg, f_radius=15, f_edge=1e-1):
    """smooth_filter.py:381-405: PIL images (or paths) in, PIL image out."""
    from PIL import Image
    if isinstance(initImg, str):
        initImg = Image.open(initImg).convert("RGB")
    best = np.array(initImg, dtype=np.float32)
    bH, bW, _ = best.shape
    best = best[:, :, ::-1].transpose((2, 0, 1))
    if isinstance(contentImg, str):
        contentImg = Image.open(contentImg).convert("RGB")
    content = np.array(contentImg.resize((bW, bH)), dtype=np.float32)[:, :, ::-1].transpose((2, 0, 1))
    input_ = np.ascontiguousarray(content, dtype=np.float32) / np.float32(255.)
    output_ = np.ascontiguousarray(best, dtype=np.float32) / np.float32(255.)
    r = smooth_local_affine(output_, input_, 1e-7, 3, bH, bW, f_radius, f_edge).transpose(1, 2, 0)
    return Image.fromarray(np.uint8(np.clip(r * 255., 0, 255.)))
