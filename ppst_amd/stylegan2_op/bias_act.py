"""``fused_leaky_relu`` / ``FusedLeakyReLU`` on ppst_fused_bias_act.

Public contract = models/networks/stylegan2_op/fused_act.py:77-96:
``y = leaky_relu(x + bias[c], slope) * scale`` with bias broadcast over dim 1,
module parameter ``bias`` of shape (channel,), defaults slope 0.2, scale sqrt 2.
The derivative is gated by the sign of the saved *output*
(fused_bias_act_kernel.cu:43, act*10+grad == 31), not of the input; it is a
diagonal linear map of the incoming gradient, so one self-differentiating
node covers backward and double backward (fused_act.py:23-53), and the bias
gradient is an ordinary (differentiable) reduction of it.
"""
import torch
from torch import nn

from .. import ops


class _GateByOutput(torch.autograd.Function):
    """g -> g * (out > 0 ? 1 : slope) * scale   (kernel mode act=3, grad=1)."""

    @staticmethod
    def forward(ctx, g, out, slope, scale):
        ctx.save_for_backward(out)
        ctx.cfg = (slope, scale)
        return ops.fused_bias_act_raw(g, None, out, 3, 1, slope, scale)

    @staticmethod
    def backward(ctx, gg):
        out, = ctx.saved_tensors
        slope, scale = ctx.cfg
        # d/dg is the same diagonal map; d/d(out) is zero almost everywhere (mode 32)
        return _GateByOutput.apply(gg.contiguous(), out, slope, scale), None, None, None


class _BiasLeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        out = ops.fused_bias_act_raw(x, bias, None, 3, 0, slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (slope, scale)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return out

    @staticmethod
    def backward(ctx, gy):
        out, = ctx.saved_tensors
        slope, scale = ctx.cfg
        gx = _GateByOutput.apply(gy.contiguous(), out, slope, scale)
        gb = None
        if ctx.has_bias:
            gb = gx.sum([0] + list(range(2, gx.ndim))).to(ctx.bias_dtype)
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
