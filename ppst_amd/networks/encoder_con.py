"""Content encoder E1 on HIP kernels (reference: models/networks/encoder_con.py:12-92,
class StyleGAN2ResnetEncodercon): FromRGB 1x1 3->32, three ResBlocks with reflection
padding, [1,2,1] blur and InstanceNorm (32->64->128->256), ToSpatialCode = two 1x1 convs +
InstanceNorm.  forward(x NCHW) -> sp (B,256,H/8,W/8)."""
import math

import torch

from .. import ops
from .base_network import BaseNetwork, as_nchw, to_nhwc


class StyleGAN2ResnetEncodercon(BaseNetwork):
    prefix = "E1."

    def forward(self, x, extract_features=False, patch_ids=None):
        x = self.from_rgb(to_nhwc(x), "FromRGB.", out_dtype=ops.act_dtype())
        for i in range(3):
            x = self.res_block(x, "DownToSpatialCode.ResBlockDownBy%d." % (2 ** i), ops.PAD_REFLECT, norm=True)
        B, H, W, C = x.shape
        sc = 1.0 / math.sqrt(C)
        y, st = self.plan("ToSpatialCode.0.Conv.weight", scale=sc)(x, stats=True)
        if ops.FUSE_TAIL["value"]:
            # the norm + leaky ReLU of ToSpatialCode.0 has ONE consumer, the 1x1 conv behind it: applied while that conv loads (round 5)
            ss = ops.in_finalize(st, H * W, post_bias=self.p("ToSpatialCode.0.Act.bias"))
            y, st = self.plan("ToSpatialCode.1.Conv.weight", scale=sc)(y, bias=self.p("ToSpatialCode.1.Conv.bias"), stats=True,
                                                                       in_ss=ss, in_act=ops.ACT_LRELU)
        else:
            x, _ = self._norm_act(y, st, H * W, self.p("ToSpatialCode.0.Act.bias"), ops.ACT_LRELU)
            y, st = self.plan("ToSpatialCode.1.Conv.weight", scale=sc)(x, bias=self.p("ToSpatialCode.1.Conv.bias"), stats=True)
        sp, _ = self._norm_act(y, st, H * W, out_dtype=torch.float32)    # the spatial code leaves the network as fp32
        return as_nchw(sp)
