"""Discriminator D on HIP kernels (reference: models/networks/discriminator.py:5-30 ->
stylegan2_layers.py:582-646): 1x1 3->64 + lrelu, 7 ResBlocks with [1,3,3,1] blur (zero
padding) 512^2 -> 4^2, 3x3 conv + lrelu, flatten in NCHW order, EqualLinear 8192->512
(fused lrelu) -> 1.  forward(x NCHW) -> (B,1)."""
import math

from .. import ops, weights
from .base_network import BaseNetwork, to_nhwc


class StyleGAN2Discriminator(BaseNetwork):
    prefix = "D."

    def __init__(self, opt=None, size=512, seed=0):
        super().__init__(opt, size=size, seed=seed)
        self.size = size

    def get_features(self, x):
        p = "stylegan2_D."
        x = self.from_rgb(to_nhwc(x), p + "convs.0.")
        for name in weights.discriminator_block_names(self.size):
            x = self.res_block(x, p + "convs.%s." % name, ops.PAD_ZERO, norm=False)
        cin = x.shape[3]
        return self.plan(p + "final_conv.Conv.weight", scale=1.0 / math.sqrt(cin * 9))(
            x, bias=self.p(p + "final_conv.Act.bias"), act=ops.ACT_LRELU)

    def forward(self, x):
        p = "stylegan2_D."
        f = ops.nhwc_to_nchw(self.get_features(x))
        f = f.reshape(f.shape[0], -1)
        w0 = self.p(p + "final_linear.0.weight")
        h = ops.linear(f, w0, self.p(p + "final_linear.0.bias"), wscale=1.0 / math.sqrt(w0.shape[1]), act=ops.ACT_LRELU)
        w1 = self.p(p + "final_linear.1.weight")
        return ops.linear(h, w1, self.p(p + "final_linear.1.bias"), wscale=1.0 / math.sqrt(w1.shape[1]))
