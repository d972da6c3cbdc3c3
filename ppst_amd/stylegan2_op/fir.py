"""``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` on ppst_upfirdn2d.

Public contract = models/networks/stylegan2_op/upfirdn2d.py:150-159 (NCHW in,
NCHW out, pad applied to both axes).  Differentiation uses the fact that the
adjoint of an upfirdn2d is again an upfirdn2d (flipped taps, up and down
exchanged, complementary padding -- the g_pad of upfirdn2d.py:116-121), so a
single autograd node that differentiates into itself yields first, second
(R1 penalty, ppst_model.py:140-159) and any higher derivative.
"""
import torch

from .. import ops


def _adjoint_pad(in_hw, out_hw, k_hw, up, down, pad):
    (ih, iw), (oh, ow), (kh, kw) = in_hw, out_hw, k_hw
    (ux, uy), (dx, dy) = up, down
    px0, _, py0, _ = pad
    return (kw - px0 - 1, iw * ux - ow * dx + px0 - ux + 1,
            kh - py0 - 1, ih * uy - oh * dy + py0 - uy + 1)


class _UpFirDn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps, up, down, pad):
        n, c, ih, iw = x.shape
        y = ops.upfirdn2d_raw(x.reshape(n * c, ih, iw, 1), taps, up[0], up[1], down[0], down[1], *pad)
        y = y.view(n, c, y.shape[1], y.shape[2])
        ctx.save_for_backward(taps)
        ctx.geom = ((ih, iw), (y.shape[2], y.shape[3]), up, down, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        taps, = ctx.saved_tensors
        in_hw, out_hw, up, down, pad = ctx.geom
        gpad = _adjoint_pad(in_hw, out_hw, tuple(taps.shape), up, down, pad)
        gx = _UpFirDn.apply(gy.contiguous(), torch.flip(taps, [0, 1]), down, up, gpad)
        assert gx.shape[2:] == in_hw
        return gx, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
