"""Image preprocessing on the device.

Mirrors the evaluators' transform (data/base_dataset.py:85-171 ``get_transform`` with
``preprocess = scale_shortside``): PIL ``Image.resize(..., Image.BICUBIC)`` to the short side,
``__make_power_2`` to a multiple of 16, ``ToTensor`` and ``Normalize(0.5, 0.5)`` -- on uint8 HWC
tensors that are already in HBM (e.g. decoded by the host once, or the output of a previous swap).

Host logic here = sizes (Python ``round``: half to even, like the reference) and Pillow's
coefficient tables (Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc``: double-precision
bicubic weights, normalised, 22-bit fixed point); the pixel arithmetic is integer HIP kernels
(csrc/imageio.hip), bit-identical to Pillow.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import check, lib

PRECISION_BITS = 32 - 8 - 2
_TABLES = {}


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def resample_tables(in_size, out_size, device):
    """(ksize, bounds int32 [out][2], coef int32 [out][ksize]) on ``device`` for the bicubic filter."""
    key = (in_size, out_size, str(device))
    hit = _TABLES.get(key)
    if hit is not None:
        return hit
    support = 2.0
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        k = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in k:            # Pillow accumulates the weights left to right in double
            ww += float(v)
        if ww != 0.0:
            k = k / ww
        fx = k * float(1 << PRECISION_BITS)
        coef[xx, :xmax] = np.where(k < 0, np.trunc(-0.5 + fx), np.trunc(0.5 + fx)).astype(np.int64)
        bounds[xx] = (xmin, xmax)
    out = (ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(coef).to(device))
    _TABLES[key] = out
    return out


def resize_bicubic_u8(img, out_h, out_w):
    """img (B,H,W,C) uint8 CUDA -> (B,out_h,out_w,C) uint8 == PIL Image.resize((out_w,out_h), BICUBIC) per image."""
    if not img.is_cuda or img.dtype != torch.uint8:
        raise RuntimeError("resize_bicubic_u8 needs a CUDA uint8 tensor (no CPU fallback)")
    img = img.contiguous()
    B, H, W, C = img.shape
    x = img
    if out_w != W:
        ks, bnd, cf = resample_tables(W, out_w, img.device)
        y = torch.empty((B, H, out_w, C), device=img.device, dtype=torch.uint8)
        check(lib.ppst_resample_u8(ops._p(x), ops._p(y), B, H, W, C, out_w, 1, ops._p(bnd), ops._p(cf), ks, ops._stream()), "ppst_resample_u8")
        x = y
    if out_h != H:
        ks, bnd, cf = resample_tables(H, out_h, img.device)
        y = torch.empty((B, out_h, x.shape[2], C), device=img.device, dtype=torch.uint8)
        check(lib.ppst_resample_u8(ops._p(x), ops._p(y), B, H, x.shape[2], C, out_h, 0, ops._p(bnd), ops._p(cf), ks, ops._stream()), "ppst_resample_u8")
        x = y
    return x


def to_tensor_normalized(img, mean=0.5, std=0.5):
    """(B,H,W,C) uint8 -> (B,C,H,W) float32, ToTensor + Normalize(mean, std)."""
    if not img.is_cuda or img.dtype != torch.uint8:
        raise RuntimeError("to_tensor_normalized needs a CUDA uint8 tensor (no CPU fallback)")
    img = img.contiguous()
    B, H, W, C = img.shape
    y = torch.empty((B, C, H, W), device=img.device, dtype=torch.float32)
    check(lib.ppst_u8_to_tensor(ops._p(img), ops._p(y), B, H, W, C, float(mean), float(std), ops._stream()), "ppst_u8_to_tensor")
    return y


def scale_shortside_size(ow, oh, target_width):
    """__scale_shortside (base_dataset.py:164-168)."""
    scale = target_width / min(ow, oh)
    return round(ow * scale), round(oh * scale)


def make_power_2_size(ow, oh, base=16):
    """__make_power_2 (base_dataset.py:141-149)."""
    return int(round(ow / base) * base), int(round(oh / base) * base)


def preprocess(img, load_size=512):
    """The evaluators' transform for ``preprocess=scale_shortside``: (B,H,W,3) uint8 -> (B,3,h,w) float32 in [-1,1]."""
    B, H, W, C = img.shape
    w1, h1 = scale_shortside_size(W, H, load_size)
    x = resize_bicubic_u8(img, h1, w1) if (w1, h1) != (W, H) else img
    w2, h2 = make_power_2_size(w1, h1)
    if (w2, h2) != (w1, h1):
        x = resize_bicubic_u8(x, h2, w2)
    return to_tensor_normalized(x)


def preprocess_resize(img, load_size=512):
    """The training transform of the reference's launcher (experiments/CelebA_launcher.py:17-18 ``preprocess="resize"``,
    base_dataset.py:91-95): ``transforms.Resize([load_size, load_size], BICUBIC)`` (a square, whatever the aspect), then
    ``__make_power_2``, ToTensor, Normalize: (B,H,W,3) uint8 -> (B,3,h,w) float32 in [-1,1]."""
    B, H, W, C = img.shape
    x = resize_bicubic_u8(img, load_size, load_size) if (W, H) != (load_size, load_size) else img
    w2, h2 = make_power_2_size(load_size, load_size)
    if (w2, h2) != (load_size, load_size):
        x = resize_bicubic_u8(x, h2, w2)
    return to_tensor_normalized(x)
