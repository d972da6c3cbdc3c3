"""Colour encoder E2 on HIP kernels (reference: models/networks/encoder_col.py:13-251,
class StyleGAN2ResnetEncodercol).  Trunk = E1's without InstanceNorm; at the four levels
(32@H, 64@H/2, 128@H/4, 256@H/8) cat(GAP, GMP) -> conv1x1 -> 3-layer projector ->
F.normalize gives four (B,2048) codes.  With ``corrmatrix`` the level features are warped
(``warp`` :100-138: pool to 64x64, corr @ feat, bilinear back) before the same heads; the
four corr @ feat products run as ONE fp32-MFMA GEMM over the concatenated 480 channels, so
the 64 MB correspondence matrix is read once instead of four times.  With ``mask`` the
per-class masked heads of :171-245 are produced as well."""
import torch

from .. import ops
from .base_network import BaseNetwork, to_nhwc

TAGS = ["9", "0", "1", "2"]
CH = [32, 64, 128, 256]


class StyleGAN2ResnetEncodercol(BaseNetwork):
    prefix = "E2."

    def _head(self, tag, x, mask=None):
        v = ops.gap_gmp(x, mask)
        w = self.p("conv1x1_%s.weight" % tag)
        v = ops.linear(v, w.reshape(w.shape[0], -1), self.p("conv1x1_%s.bias" % tag))
        q = "projector%s." % tag
        v = ops.linear(v, self.p(q + "1.weight"), self.p(q + "1.bias"), relu_in=True)
        v = ops.linear(v, self.p(q + "3.weight"), self.p(q + "3.bias"), relu_in=True)
        v = ops.linear(v, self.p(q + "5.weight"), self.p(q + "5.bias"), relu_in=True)
        return ops.l2norm_rows(v, 1e-12, 1)

    def trunk(self, x, dtype=torch.float32):
        feats = [self.from_rgb(to_nhwc(x), "FromRGB.", out_dtype=dtype)]
        for i in range(3):
            feats.append(self.res_block(feats[-1], "DownToGlobalCode1.ResBlockDownBy%d." % (2 ** i), ops.PAD_REFLECT, norm=False))
        return feats

    def warp_levels(self, feats, corr):
        """E2.warp for all four levels with one GEMM.  corr (B,4096,4096)."""
        B = feats[0].shape[0]
        V = torch.empty((B, 64, 64, sum(CH)), device=corr.device, dtype=torch.float32)
        off = 0
        for f, c in zip(feats, CH):
            assert f.shape[1] == f.shape[2] and f.shape[1] % 64 == 0, "correspondence needs square inputs (64x64 code grid)"
            ops.avgpool(f, f.shape[1] // 64, out=V[..., off:off + c])
            off += c
        Wv = ops.gemm_nn(corr, V.view(B, 4096, sum(CH)), mode="x3").view(B, 64, 64, sum(CH))
        out, off = [], 0
        for f, c in zip(feats, CH):
            sl = Wv[..., off:off + c]
            out.append(sl if f.shape[1] == 64 else ops.bilinear(sl, f.shape[1], f.shape[2]))
            off += c
        return out

    @staticmethod
    def _mask_planes(mask):
        """NCHW one-hot mask -> list over pyramid levels of (B,H,W,3) NHWC (MaxPool2d(2), :218)."""
        m = to_nhwc(mask)
        if not m.is_contiguous():
            m = m.contiguous()
        levels = [m]
        for _ in range(3):
            levels.append(ops.maxpool2(levels[-1]))
        return levels

    def forward(self, x=None, extract_features=False, mask=None, corrmatrix=None):
        # half-precision activation storage (ops.HALF_STORE) for the plain code pass; the warp / masked heads (pooling,
        # bilinear resize, the correspondence GEMM) read fp32 features
        feats = self.trunk(x, ops.act_dtype() if (corrmatrix is None and mask is None) else torch.float32)
        vectors = [self._head(t, f) for t, f in zip(TAGS, feats)]
        vectors_w, pm, pmw = [], [], []
        warped = None
        if corrmatrix is not None:
            if isinstance(corrmatrix, (list, tuple)):  # simple_swapping_evaluator.py:53 wraps it in a list
                corrmatrix = corrmatrix[0]
            warped = self.warp_levels(feats, corrmatrix.detach())
            vectors_w = [self._head(t, f) for t, f in zip(TAGS, warped)]
        if mask is not None:
            from .. import glue
            levels = self._mask_planes(mask)
            sw_levels = self._mask_planes(glue.swap(mask)) if warped is not None else None
            for lvl, (t, f) in enumerate(zip(TAGS, feats)):
                for i in range(3):
                    pm.append(self._head(t, f, levels[lvl][..., i].contiguous()))
                    if warped is not None:
                        pmw.append(self._head(t, warped[lvl], sw_levels[lvl][..., i].contiguous()))
            return vectors, pm, vectors_w, pmw
        return vectors, vectors_w
