"""Generator G on HIP kernels (reference: models/networks/generator.py:104-281, class
StyleGAN2ResnetGenerator, built from StyledConv / ToRGB of stylegan2_layers.py:439-495).

forward(spatial_code (B,256,h,w), [4 x (B,2048)], extract_features=False, noise=None)
  -> rgb (B,3,8h,8w)            | (rgb, feat (B,256,h,w), feat1 (B,64,256,256))

Every StyledConv = one fused MFMA conv launch (bias + noise + leaky-relu epilogue and
instance-norm tile statistics in the same kernel) + a (B,2C) style GEMV + a tiny finalize
+ one apply pass that also carries the residual add / 1/sqrt2 of the resnet blocks.  The
upsampling StyledConv is the 4x4 stride-2 transposed conv of stylegan2_layers.py:312-321
executed as four 2x2 output-phase convs inside the same kernel.

``noise``: dict '<Block>.<conv>' -> (B,1,H,W) tensors for the 14 NoiseInjection layers
(stylegan2_layers.py:376-399).  The reference draws them on the fly; here they must be
given explicitly whenever a noise weight is non-zero (parity needs identical noise);
``noise='random'`` draws them with torch.randn like the reference.
"""
import math

import torch

from .. import ops
from .base_network import BaseNetwork, as_nchw, to_nhwc, INV_SQRT2

HEAD_CH = [(256, 256), (256, 256), (256, 384), (384, 512)]
UP = [(16, 512, 512), (32, 512, 256), (64, 256, 128)]


class StyleGAN2ResnetGenerator(BaseNetwork):
    prefix = "G."

    # -- StyledConv (stylegan2_layers.py:439-475) -----------------------------
    def _styled_consts(self, p):
        """folded per-channel bias (conv.bias + bias + activate.bias) and the noise weight."""
        ts = [self.p(p + "conv.bias"), self.p(p + "bias"), self.p(p + "activate.bias"), self.p(p + "noise.weight")]

        def build():
            b = (ts[0] + ts[1].reshape(-1) + ts[2]).contiguous()
            return b, float(ts[3].item())
        return self.cached(("sc", p), ts, build)

    # -- all StyleMod GEMVs that share a style code run as ONE weight-streaming launch -----
    def _style_groups(self):
        groups = [["HeadResnetBlock%d.%s." % (i, c) for i in range(4) for c in ("conv1", "conv2")]]
        groups += [["UpsamplingResBlock%d.%s." % (key, c) for c in ("conv1", "conv2")] for key, _, _ in UP]
        groups[-1].append("ToRGB.")  # ToRGB is modulated by codes[0] like the last upsampling block
        return groups

    def _style_table(self, codes):
        """{layer prefix: (B, 2C) StyleMod output} -- EqualizedLinear(2048 -> 2C, wscale) per layer
        (stylegan2_layers.py:361-374), batched per style code: groups use codes[-1], [-2], [-3], [-4]."""
        out = {}
        for gi, names in enumerate(self._style_groups()):
            ws = [self.p(n + "epi1.style_mod.lin.weight") for n in names]
            bs = [self.p(n + "epi1.style_mod.lin.bias") for n in names]
            W, Bv = self.cached(("stylecat", gi), ws + bs, lambda: (torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()))
            y = ops.linear(codes[-1 - gi], W, Bv, wscale=W.shape[1] ** -0.5)
            off = 0
            for n, w in zip(names, ws):
                out[n] = y[:, off:off + w.shape[0]]
                off += w.shape[0]
        return out

    def styled_conv(self, x, p, style, key, noise, upsample=False, res=None, out_scale=1.0, in_ss=None, defer=False,
                    out_stats=None, res_up2=False):
        """in_ss: (scale, shift) of the producer StyledConv, applied while this conv stages its
        input ("normalise on load").  defer=True returns (raw conv output, its scale/shift)
        instead of running the apply pass."""
        B, H, W, cin = x.shape
        bias, nw = self._styled_consts(p)
        kind = "conv"
        if upsample:
            if min(H, W) * 2 >= 128:
                kind = "convT"
            else:  # the <128 px branch: nearest x2 + 3x3 conv (stylegan2_layers.py:322-323)
                assert in_ss is None
                x = ops.upsample_nearest2(x)
        nz = None
        if nw != 0.0:
            if noise is None:
                raise RuntimeError("noise weight of %s is non-zero: pass noise tensors (or noise='random')" % p)
            nz = noise[key]
            if nz.shape[0] < B and B % nz.shape[0] == 0:
                # one fixed row applies to the whole batch, like NoiseInjection.fixed_noise; several calls' batches run as one
                # (content and style feature passes concatenated): the fixture's rows repeat, so image i of every sub-batch
                # sees row i exactly as in separate calls
                nz = self._batch_noise(nz, B)
            nz = nz.contiguous()
        y, st = self.plan(p + "conv.weight", kind)(x, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, stats=True,
                                                   in_ss=in_ss)
        ss = ops.in_finalize(st, y.shape[1] * y.shape[2], style=style[p])
        if defer:
            return y, ss
        if out_stats is not None:  # 'rep' | 'plain': also emit the IN partials of the block output (feature heads)
            return ops.affine_act_stats(y, ss, res=res, out_scale=out_scale, rep_pad=(out_stats == "rep"), res_up2=res_up2)
        return ops.affine_act(y, ss, res=res, out_scale=out_scale, res_up2=res_up2)

    def _batch_noise(self, nz, B):
        """``nz`` (rows, 1, H, W) repeated to B rows, kept while the source tensor is unchanged (a pinned noise dict is
        expanded once, not by 14 copy kernels per generator pass)."""
        cache = self.__dict__.setdefault("_nz_cache", {})
        k = (id(nz), B)
        hit = cache.get(k)
        if hit is not None and hit[0] is nz and hit[1] == nz._version:
            return hit[2]
        out = nz.repeat(B // nz.shape[0], 1, 1, 1).contiguous()
        if len(cache) >= 64:
            cache.clear()
        cache[k] = (nz, nz._version, out)       # holding ``nz`` keeps its id from being reused
        return out

    # -- correspondence feature heads (generator.py:174-238) ------------------
    def _feat_head(self, x, p, k, out, st=None, tail=None):
        B, H, W, C = x.shape
        pad = ops.PAD_REPLICATE
        # InstanceNorm runs on the ReplicationPad2d(1)-padded tensor for the 3x3 heads
        if st is None:
            st = ops.in_stats(x, rep_pad=(k == 3))
        cnt = (H + 2) * (W + 2) if k == 3 else H * W
        # both inner norms (+PReLU) are applied by the consuming conv while it stages its input
        y, st = self.plan(p + "2.weight")(x, bias=self.p(p + "2.bias"), stats=True, pad_mode=pad, in_ss=ops.in_finalize(st, cnt))
        y, st = self.plan(p + "6.weight")(y, bias=self.p(p + "6.bias"), stats=True, pad_mode=pad,
                                          in_ss=ops.in_finalize(st, H * W), in_act=ops.ACT_PRELU, in_prelu=self.p(p + "4.weight"))
        ss = ops.in_finalize(st, H * W)
        if tail is not None:   # (feat slice, feat1 slice): pooled / resized copies only, f itself is not stored
            ops.head_tail(y, ss, tail[0], tail[1], act=ops.ACT_PRELU, prelu=self.p(p + "8.weight"))
            return None
        return ops.affine_act(y, ss, act=ops.ACT_PRELU, prelu=self.p(p + "8.weight"), out=out)

    def _residual_block(self, x, p, defer=False):
        """generator.py:10-32.  defer=True: returns (conv2 output, its (a, s), x, slope) instead of running the merge pass
        prelu(IN(conv2) + x) -- the caller's 1x1 conv applies it while it loads (ConvPlan in_res)."""
        B, H, W, C = x.shape
        a = self.p(p + "prelu.weight")
        y, st = self.plan(p + "conv1.weight")(x, bias=self.p(p + "conv1.bias"), stats=True, pad_mode=ops.PAD_REPLICATE)
        y, st = self.plan(p + "conv2.weight")(y, bias=self.p(p + "conv2.bias"), stats=True, pad_mode=ops.PAD_REPLICATE,
                                              in_ss=ops.in_finalize(st, H * W), in_act=ops.ACT_PRELU, in_prelu=a)
        ss = ops.in_finalize(st, H * W)
        if defer:
            return y, ss, x, a
        return ops.affine_act(y, ss, res=x, res_before_act=True, act=ops.ACT_PRELU, prelu=a)

    def _feat1_tail(self, feat1):
        """layert1 = ResidualBlock(256) + Conv2d(256, 64, 1) (generator.py:235-238).  The block's merged output has ONE consumer,
        the 1x1 conv: in the fp32-class mode that conv applies the merge while it loads its fragments (round 5: the 2-GB tensor of
        the batch-16 feature pass is neither written nor read back; ops.FUSE_TAIL)."""
        plan = self.plan("layert1.1.weight")
        if ops.FUSE_TAIL["value"] and feat1.dtype == torch.float32 and plan.takes_in_res(feat1.shape[1], feat1.shape[2]):
            y, ss, x, a = self._residual_block(feat1, "layert1.0.", defer=True)
            return plan(y, bias=self.p("layert1.1.bias"), in_ss=ss, in_act=ops.ACT_PRELU, in_prelu=a, in_res=x)
        return plan(self._residual_block(feat1, "layert1.0."), bias=self.p("layert1.1.bias"))

    def make_noise(self, B, S, device):
        """the fourteen N(0,1) planes NoiseInjection draws in one pass (stylegan2_layers.py:388-390), from ONE torch.randn."""
        sizes = [("HeadResnetBlock%d.%s" % (i, c), S) for i in range(4) for c in ("conv1", "conv2")]
        sizes += [("UpsamplingResBlock%d.%s" % (key, c), S << (j + 1)) for j, (key, _, _) in enumerate(UP) for c in ("conv1", "conv2")]
        flat = torch.randn(B * sum(s * s for _, s in sizes), device=device)
        out, off = {}, 0
        for name, s in sizes:
            out[name] = flat[off:off + B * s * s].view(B, 1, s, s)
            off += B * s * s
        return out

    def forward(self, spatial_code, global_codes, extract_features=False, noise=None, want_rgb=True):
        """want_rgb=False (only with extract_features): skip ToRGB -- `extract_feat_from_image` (ppst_model.py:255-262)
        throws the image of its feature passes away; the reference computes it all the same.  Returns (None, feat, feat1)."""
        sp = to_nhwc(spatial_code)
        B, S = sp.shape[0], sp.shape[1]
        if isinstance(noise, str) and noise == "random":
            noise = self.make_noise(B, S, sp.device)
        codes = [ops.l2norm_rows(c, 1e-8, 0) for c in global_codes]  # util.normalize (generator.py:246)
        g = codes[-1]
        styles = self._style_table(codes)
        ws = self.p("SpatialCodeModulation.scale.weight")
        inv = 1.0 / math.sqrt(ws.shape[1])
        scale = ops.linear(g, ws, self.p("SpatialCodeModulation.scale.bias"), wscale=inv)
        shift = ops.linear(g, self.p("SpatialCodeModulation.bias.weight"), self.p("SpatialCodeModulation.bias.bias"), wscale=inv)
        # half-precision activation storage (ops.HALF_STORE): the plain image pass runs in ops.act_dtype(); the feature-extraction
        # pass (correspondence heads, pooled / resized copies) keeps fp32 storage
        adt = torch.float32 if extract_features else ops.act_dtype()
        x = ops.spatial_modulation(sp if sp.is_contiguous() else sp.contiguous(), scale, shift, out_dtype=adt)
        for i, (ci, co) in enumerate(HEAD_CH):
            q = "HeadResnetBlock%d." % i
            skip = x if ci == co else self.plan(q + "skip.Conv.weight", scale=1.0 / math.sqrt(ci))(x)
            r, rss = self.styled_conv(x, q + "conv1.", styles, "HeadResnetBlock%d.conv1" % i, noise, defer=True)
            want = "rep" if (extract_features and i == len(HEAD_CH) - 1) else None
            x = self.styled_conv(r, q + "conv2.", styles, "HeadResnetBlock%d.conv2" % i, noise, res=skip, out_scale=INV_SQRT2, in_ss=rss,
                                 out_stats=want)
            if want:
                x, xst = x
        feat = feat1 = last_merge = None
        if extract_features:
            h, w = x.shape[1], x.shape[2]
            feat = torch.empty((B, h, w, 256), device=x.device, dtype=torch.float32)
            feat1 = torch.empty((B, 256, 256, 256), device=x.device, dtype=torch.float32)
            f = self._feat_head(x, "layer32.", 3, out=feat[..., 0:64], st=xst)
            ops.bilinear(f, 256, 256, out=feat1[..., 0:64])
        for j, (key, ci, co) in enumerate(UP):
            q = "UpsamplingResBlock%d." % key
            g = codes[-2 - j]
            if ci == co:
                skip = x
            else:
                skip = self.plan(q + "skip.Conv.weight", scale=1.0 / math.sqrt(ci))(x, bias=self.p(q + "skip.Act.bias"), act=ops.ACT_LRELU)
            # (the x2 bilinear upsample of the skip is sampled on the fly by the apply pass)
            r, rss = self.styled_conv(x, q + "conv1.", styles, "UpsamplingResBlock%d.conv1" % key, noise, upsample=True, defer=True)
            want = ("rep" if j < 2 else "plain") if extract_features else None
            if j == len(UP) - 1 and not extract_features and ops.FUSE_TAIL["value"] and ops.FUSE_TAIL["torgb"]:
                # image pass: the last block's merged output is read by ToRGB only -- its 1x1 conv applies the merge on load
                x, xss = self.styled_conv(r, q + "conv2.", styles, "UpsamplingResBlock%d.conv2" % key, noise, in_ss=rss, defer=True)
                last_merge = (xss, skip)
                break
            x = self.styled_conv(r, q + "conv2.", styles, "UpsamplingResBlock%d.conv2" % key, noise, res=skip, out_scale=INV_SQRT2, in_ss=rss,
                                 out_stats=want, res_up2=True)
            if extract_features:
                x, xst = x
                c0 = 64 * (j + 1)
                name, kk = "layer%d." % (2 ** (j + 6)), 3 if j < 2 else 1
                if x.shape[1] in (256, 512) and x.shape[1] % h == 0:
                    self._feat_head(x, name, kk, out=None, st=xst, tail=(feat[..., c0:c0 + 64], feat1[..., c0:c0 + 64]))
                else:
                    f = self._feat_head(x, name, kk, out=None, st=xst)
                    ops.avgpool(f, f.shape[1] // h, out=feat[..., c0:c0 + 64])
                    ops.bilinear(f, 256, 256, out=feat1[..., c0:c0 + 64])
        # ToRGB (stylegan2_layers.py:477-495): 1x1 conv + biases -> InstanceNorm(3) -> StyleMod
        if extract_features and not want_rgb:
            for i in range(3):
                feat = self._residual_block(feat, "layert.%d." % i)
            feat1 = self._feat1_tail(feat1)
            return None, as_nchw(feat), as_nchw(feat1)
        wr = self.p("ToRGB.conv.weight")
        brgb = self.cached(("rgbb",), [self.p("ToRGB.conv.bias"), self.p("ToRGB.bias")],
                           lambda: (self.p("ToRGB.conv.bias") + self.p("ToRGB.bias").reshape(-1)).contiguous())
        if last_merge is not None:
            y = ops.torgb_apply(x, last_merge[0], last_merge[1], INV_SQRT2, wr, brgb, 1.0 / math.sqrt(wr.shape[1]))
        else:
            y = ops.conv1x1_small_cout(x, wr, brgb, 1.0 / math.sqrt(wr.shape[1]))
        ss = ops.in_finalize(ops.in_stats(y), y.shape[1] * y.shape[2], style=styles["ToRGB."])
        rgb = ops.nhwc_to_nchw(ops.affine_act(y, ss))
        if not extract_features:
            return rgb
        for i in range(3):
            feat = self._residual_block(feat, "layert.%d." % i)
        feat1 = self._feat1_tail(feat1)
        return rgb, as_nchw(feat), as_nchw(feat1)
