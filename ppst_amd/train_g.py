"""Generator / encoder update of the GAN train step on HIP kernels.

Reference: PPSTOptimizer.train_generator_one_step (optimizers/ppst_optimizer.py:73-94) ->
PPSTModel.compute_generator_losses (models/ppst_model.py:161-235): ``sum(v.mean()).backward()`` through E1, E2, G
(D frozen: only d/d(image) flows through it), then three Adam optimisers (lr 1e-3, betas (0, 0.99)).

The differentiable forward below is the training-mode twin of ppst_amd/networks/*: the same kernels, unfused where
the backward needs the intermediate (the inference path merges norms into the consumer's loads).  Every block is an
autograd Function over HIP kernels (ppst_amd/autograd.py); parameters of each network live in ONE flat buffer so
the data-parallel gradient average is one all-reduce per network and Adam one launch.
"""
import math

import torch

from . import autograd as A
from . import glue, ops
from .networks.base_network import to_nhwc
from .train import DiscriminatorTrainer, ddp_average_

SQRT2 = math.sqrt(2.0)
INV_SQRT2 = 1.0 / SQRT2
# tuning switches of the differentiable generator (tests/train_ab.py flips them for same-process A/B timing)
# "gate": StyledConv's leaky-ReLU gate rides on the norm backward's apply pass (round 5; off while the gate tape records / replays)
# "gmp_multi": the four poolings of a feature map (plain + three class masks) as one multi-head launch, forward and backward
TRAIN_FUSE = {"merge": True, "res_up2": True, "gate": True, "gmp_multi": True}
HEAD_CH = [(256, 256), (256, 256), (256, 384), (384, 512)]
UP = [(16, 512, 512), (32, 512, 256), (64, 256, 128)]
TAGS = ["9", "0", "1", "2"]
CH = [32, 64, 128, 256]


class FlatParams:
    """Parameters of one network as views into a flat buffer, with flat gradient and Adam state."""

    def __init__(self, net, lr, beta1, beta2):
        self.net = net
        self.names = [n for n, _ in net.named_parameters()]
        params = [p for _, p in net.named_parameters()]
        self.params = params
        self.sizes = [p.numel() for p in params]
        flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        self.flat, self.grad = flat, torch.zeros_like(flat)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.offsets, off = {}, 0
        for n_, p, sz in zip(self.names, params, self.sizes):
            p.data = flat[off:off + sz].view_as(p)
            p.requires_grad_(True)
            p.grad = self.grad[off:off + sz].view_as(p)     # autograd accumulates in place into the flat gradient
            p._ppst_direct = True                           # ... and the backward kernels add straight into it (autograd._direct)
            p._ppst_on_grad = self._on_grad
            self.offsets[n_] = (off, sz)
            off += sz
        self.on_event = None                                # set by the trainer: called once per gradient contribution
        net._flat.clear(); net._cache.clear()
        self.lr, self.b1, self.b2, self.eps, self.step_count = lr, beta1, beta2, 1e-8, 0

    def _on_grad(self):
        if self.on_event is not None:
            self.on_event()

    def g(self, name):
        off, sz = self.offsets[name]
        return self.grad[off:off + sz]

    def scalar(self, name):
        """Host value of a one-element parameter (the 14 NoiseInjection weights: the conv kernel takes them by value).  All
        of them come over in ONE device-to-host copy per parameter update -- `float(param)` per StyledConv call was a
        stream synchronisation each (50-80 per train step, each draining the launch queue)."""
        key = (self.step_count, self.flat._version)
        if getattr(self, "_scalar_key", None) != key:
            names = [n_ for n_, sz in zip(self.names, self.sizes) if sz == 1]
            if not hasattr(self, "_scalar_idx"):
                self._scalar_idx = torch.tensor([self.offsets[n_][0] for n_ in names], dtype=torch.long, device=self.flat.device)
            vals = self.flat.index_select(0, self._scalar_idx).cpu().tolist() if names else []
            self._scalars, self._scalar_key = dict(zip(names, vals)), key
        return self._scalars[name]

    def derived(self, key, build):
        """A value computed from the parameters under no_grad (the folded StyledConv bias), kept until the parameters change --
        same lifetime as the host scalars: one parameter update.  (The three-bias sum alone was 2 aten launches per StyledConv
        call, ~170 per train step.)"""
        ver = (self.step_count, self.flat._version)
        if getattr(self, "_derived_key", None) != ver:
            self._derived, self._derived_key = {}, ver
        hit = self._derived.get(key)
        if hit is None:
            hit = self._derived[key] = build()
        return hit

    def invalidate(self):
        """Forget host-side copies of parameter values (the noise-weight scalars) and the network's packed weights: call after
        ANY write to the parameters that did not go through adam() -- load_state_dict / PPSTModel.load copy into the parameter
        views, which bumps their version counters, not the flat buffer's."""
        self._scalar_key = self._derived_key = None
        self.net._cache.clear()

    def owns_parameters(self):
        off = 0
        for p, n in zip(self.params, self.sizes):
            if p.data_ptr() != self.flat.data_ptr() + 4 * off or p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                return False
            off += n
        return True

    def zero_grad(self):
        self.grad.zero_()

    def adam(self):
        if not self.owns_parameters():
            raise RuntimeError("parameters / gradients no longer alias the flat buffers of this trainer")
        self.step_count += 1
        ops.adam_step_(self.flat, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self.step_count)
        self._scalar_key = self._derived_key = None      # host scalars / folded biases are stale; the conv plans are re-packed in place (two launches)
        self.net.refresh_plans()


class GeneratorTrainer:
    """Differentiable E1 / E2 / G forward + the generator iteration of PPSTOptimizer."""

    @classmethod
    def for_model(cls, model, **kw):
        """The trainer that owns ``model`` (created on first use and kept on it, like DiscriminatorTrainer.for_network): its
        constructor rebinds the parameters of E1 / E2 / G into flat buffers, a second one would orphan the first one's state."""
        tr = model.__dict__.get("_trainer")
        if tr is None or not all(f.owns_parameters() for f in tr.fp.values()):
            tr = cls(model, **kw)
        return tr

    def invalidate(self):
        for f in self.fp.values():
            f.invalidate()
        if self.d_trainer is not None:
            self.d_trainer.D._cache.clear()

    def __init__(self, model, lr=1e-3, beta1=0.0, beta2=0.99, world=1):
        self.model = model
        model.__dict__["_trainer"] = self
        self.opt = model.opt
        self.world = world
        self.G, self.E1, self.E2 = model.G, model.E1, model.E2
        self.fp = {"G": FlatParams(model.G, lr, beta1, beta2), "E2": FlatParams(model.E2, lr, beta1, beta2),
                   "E1": FlatParams(model.E1, lr, beta1, beta2)}
        self.d_trainer = DiscriminatorTrainer.for_network(model.D) if hasattr(model, "D") else None

    # ------------------------------------------------------------ ConvLayer / ResBlock (stylegan2_layers.py:497-579)
    def _res_block(self, net, x, p, norm):
        P = net.p
        cin = x.shape[3]
        ks = P(p + "conv2.Blur.kernel").shape[0]
        sc1, scs = 1.0 / math.sqrt(cin * 9), 1.0 / math.sqrt(cin)
        pad_c, pad_s = (ks - 2) + 2, (ks - 2) + 0
        w1, w2, ws = p + "conv1.Conv.weight", p + "conv2.Conv.weight", p + "skip.Conv.weight"
        xs = A.BlurDownFn.apply(x, net, p + "skip.Blur.kernel", (pad_s + 1) // 2, pad_s // 2)
        if not norm:
            y1 = A.conv(x, P(w1), net, w1, bias=P(p + "conv1.Act.bias"), scale=sc1, pad_mode=A.REFLECT, act=A.LRELU)
            a2 = A.blur_conv(y1, P(w2), net, w2, p + "conv2.Blur.kernel", bias=P(p + "conv2.Act.bias"), scale=sc1,
                             p0=(pad_c + 1) // 2, p1=pad_c // 2, pad_mode=A.REFLECT, act=A.LRELU)
            skip = A.conv(xs, P(ws), net, ws, scale=scs)
            return A.AddScaleFn.apply(a2, skip, INV_SQRT2)
        y1, st1 = A.conv(x, P(w1), net, w1, scale=sc1, pad_mode=A.REFLECT, stats=True)
        a1 = A.instance_norm(y1, st1, post_bias=P(p + "conv1.Act.bias"), act=A.LRELU)
        y2, st2 = A.blur_conv(a1, P(w2), net, w2, p + "conv2.Blur.kernel", scale=sc1, p0=(pad_c + 1) // 2, p1=pad_c // 2,
                              pad_mode=A.REFLECT, stats=True)
        a2 = A.instance_norm(y2, st2, post_bias=P(p + "conv2.Act.bias"), act=A.LRELU)
        ys, sts = A.conv(xs, P(ws), net, ws, scale=scs, stats=True)
        return A.AddScaleFn.apply(a2, A.instance_norm(ys, sts), INV_SQRT2)

    def _from_rgb(self, net, img, p):
        x = A.ToNHWCFn.apply(img) if img.requires_grad else to_nhwc(img)     # img is NCHW like the reference's
        x = x if x.is_contiguous() else x.contiguous()
        w = net.p(p + "Conv.weight")
        return A.FromRGBFn.apply(x, w, net.p(p + "Act.bias"), 1.0 / math.sqrt(w.shape[1]))

    # ------------------------------------------------------------ E1 (encoder_con.py:82-92)
    def encoder_con(self, img):
        """img NCHW -> sp NHWC (B, H/8, W/8, 256)."""
        net, P = self.E1, self.E1.p
        x = self._from_rgb(net, img, "FromRGB.")
        for i in range(3):
            x = self._res_block(net, x, "DownToSpatialCode.ResBlockDownBy%d." % (2 ** i), norm=True)
        sc = 1.0 / math.sqrt(x.shape[3])
        n0, n1 = "ToSpatialCode.0.Conv.weight", "ToSpatialCode.1.Conv.weight"
        y, st = A.conv(x, P(n0), net, n0, scale=sc, stats=True)
        x = A.instance_norm(y, st, post_bias=P("ToSpatialCode.0.Act.bias"), act=A.LRELU)
        y, st = A.conv(x, P(n1), net, n1, bias=P("ToSpatialCode.1.Conv.bias"), scale=sc, stats=True)
        return A.instance_norm(y, st)

    # ------------------------------------------------------------ E2 (encoder_col.py:150-251)
    def _e2_heads(self, tag, items, pooled=None):
        """The projection head of level ``tag`` (encoder_col.py:162-168, 217-245: GAP || GMP of x * mask -> conv1x1 as a linear ->
        three ReLU + linear projectors -> F.normalize) for several (feature map, mask) pairs at once: the pooled vectors are
        stacked and go through the shared linears as ONE batch -- the same arithmetic per row (the weight gradients sum over rows
        either way), a chain of launches per level instead of one per head (76 heads per generator iteration)."""
        P = self.E2.p
        if pooled is not None:          # (stacked pooled vectors of n heads, from the multi-head pooling)
            v, n = pooled
            vs = [None] * n
        else:
            vs = [A.GapGmpFn.apply(x, mask) for x, mask in items]
            v = vs[0] if len(vs) == 1 else torch.cat(vs, 0)
        v = A.linear(v, P("conv1x1_%s.weight" % tag), P("conv1x1_%s.bias" % tag))
        q = "projector%s." % tag
        for i in (1, 3, 5):
            v = A.linear(v, P(q + "%d.weight" % i), P(q + "%d.bias" % i), relu_in=True)
        v = A.L2NormFn.apply(v, 1e-12, 1)
        if len(vs) == 1:
            return [v]
        B = v.shape[0] // len(vs)
        return [v[i * B:(i + 1) * B] for i in range(len(vs))]

    def encoder_col(self, img, mask=None, corrmatrix=None):
        net = self.E2
        feats = [self._from_rgb(net, img, "FromRGB.")]
        for i in range(3):
            feats.append(self._res_block(net, feats[-1], "DownToGlobalCode1.ResBlockDownBy%d." % (2 ** i), norm=False))
        warped = self._warp_levels(feats, corrmatrix) if corrmatrix is not None else None
        levels = net._mask_planes(mask) if mask is not None else None
        sw = net._mask_planes(glue.swap(mask)) if (mask is not None and warped is not None) else None
        vectors, vectors_w, pm, pmw = [], [], [], []
        multi = TRAIN_FUSE["gmp_multi"] and mask is not None      # (under the gate tape its backward walks the heads one by one)
        for lvl, (t, f) in enumerate(zip(TAGS, feats)):
            # every head of this level in one batch, in the reference's order of use
            if multi and (f.shape[1] * f.shape[2]) % 16 == 0:
                # the four poolings of a map (plain + three class masks) in ONE read of it, their gradients written as one sum
                B = f.shape[0]
                vf = A.GapGmpMultiFn.apply(f, levels[lvl])                       # rows [plain | mask 0 | mask 1 | mask 2] x B
                if warped is not None:
                    # the reference's order of use alternates map / warped map per head: one interleaving copy, no per-head slices
                    vw = A.GapGmpMultiFn.apply(warped[lvl], sw[lvl])
                    vf = torch.stack((vf.view(4, B, -1), vw.view(4, B, -1)), 1).reshape(8 * B, -1)
                out = self._e2_heads(t, None, pooled=(vf, vf.shape[0] // B))
            else:
                items = [(f, None)]
                if warped is not None:
                    items.append((warped[lvl], None))
                if mask is not None:
                    for i in range(3):
                        items.append((f, levels[lvl][..., i].contiguous()))
                        if warped is not None:
                            items.append((warped[lvl], sw[lvl][..., i].contiguous()))
                out = self._e2_heads(t, items)
            vectors.append(out[0])
            k = 1
            if warped is not None:
                vectors_w.append(out[k]); k += 1
            if mask is not None:
                for i in range(3):
                    pm.append(out[k]); k += 1
                    if warped is not None:
                        pmw.append(out[k]); k += 1
        if mask is not None:
            return vectors, pm, vectors_w, pmw
        return vectors, vectors_w

    def _warp_levels(self, feats, corr):
        """E2.warp (encoder_col.py:100-138) for the four levels with one GEMM; the correspondence matrix receives a
        gradient through the first level only (the others use corrmatrix.detach(), :197)."""
        B = feats[0].shape[0]
        feats = [f.float() for f in feats]          # (the correspondence branch is fp32: a half-stored trunk is widened here, differentiably)
        pooled = [f if f.shape[1] == 64 else A.AvgPoolFn.apply(f, f.shape[1] // 64) for f in feats]
        V = torch.cat(pooled, dim=3).reshape(B, 4096, sum(CH))
        Wv = A.WarpGemmFn.apply(corr, V, CH[0]).view(B, 64, 64, sum(CH))
        out, off = [], 0
        for f, c in zip(feats, CH):
            sl = Wv[..., off:off + c]
            out.append(sl if f.shape[1] == 64 else A.BilinearFn.apply(sl, f.shape[1], f.shape[2]))
            off += c
        return out

    # ------------------------------------------------------------ G (generator.py:244-281)
    def _styled_conv(self, x, p, code, key, noise, upsample=False, res=None, out_scale=1.0, res_up2=False):
        net, P = self.G, self.G.p
        parts = (P(p + "conv.bias"), P(p + "bias"), P(p + "activate.bias"))                # three biases of StyledConv collapse
        bias_params = parts if all(A._direct(q) is not None for q in parts) else None
        if bias_params is not None:       # the sum is a constant of the graph; the block adds d/d(bias) to all three gradients itself
            def fold():
                with torch.no_grad():
                    return parts[0] + parts[1].reshape(-1) + parts[2]
            bias = self.fp["G"].derived(("styled_bias", p), fold)
        else:
            bias = parts[0] + parts[1].reshape(-1) + parts[2]
        kind = "conv"
        if upsample:
            if min(x.shape[1], x.shape[2]) * 2 < 128:
                raise NotImplementedError("training below 64x64 feature maps (the nearest-upsample branch) is not on the path")
            kind = "convT"
        nz = noise[key] if isinstance(noise, dict) else None
        if nz is None:
            B, H, W = x.shape[0], x.shape[1] * (2 if upsample else 1), x.shape[2] * (2 if upsample else 1)
            nz = torch.randn(B, 1, H, W, device=x.device)          # NoiseInjection draws N(0,1) (stylegan2_layers.py:388-390)
        wn = p + "conv.weight"
        fuse_gate = TRAIN_FUSE["gate"]     # (`a` has exactly one consumer: the norm below; under the gate tape the norm's backward runs the gate pass itself)
        a, st = A.conv(x, P(wn), net, wn, bias=bias, kind=kind, act=A.LRELU, noise_w=P(p + "noise.weight"), noise=nz.contiguous(), stats=True,
                       noise_w_host=self.fp["G"].scalar(p + "noise.weight"), bias_params=bias_params, gate_downstream=fuse_gate)
        wl = P(p + "epi1.style_mod.lin.weight")
        style = A.linear(code, wl, P(p + "epi1.style_mod.lin.bias"), wscale=wl.shape[1] ** -0.5)
        return A.instance_norm(a, st, style=style, res=res, out_scale=out_scale, res_up2=res_up2, post_gate=fuse_gate)     # res: the block's (skip + res) / sqrt2 merge

    def generator(self, sp, global_codes, noise=None, extract_features=False):
        """sp NHWC (B,h,w,256), codes 4 x (B,2048) -> rgb NCHW (B,3,8h,8w) [, feat NHWC, feat1 NHWC]."""
        net, P = self.G, self.G.p
        if not isinstance(noise, dict):
            # NoiseInjection draws N(0,1) per layer (stylegan2_layers.py:388-390): all fourteen planes of this pass from ONE
            # torch.randn (one generator launch instead of fourteen), handed out as views
            B, H0, W0 = sp.shape[0], sp.shape[1], sp.shape[2]
            sizes = [("HeadResnetBlock%d.%s" % (i, c), 0) for i in range(len(HEAD_CH)) for c in ("conv1", "conv2")]
            sizes += [("UpsamplingResBlock%d.%s" % (key, c), j + 1) for j, (key, _, _) in enumerate(UP) for c in ("conv1", "conv2")]
            flat = torch.randn(B * sum((H0 << e) * (W0 << e) for _, e in sizes), device=sp.device)
            noise, off = {}, 0
            for name, e in sizes:
                n_ = B * (H0 << e) * (W0 << e)
                noise[name] = flat[off:off + n_].view(B, 1, H0 << e, W0 << e)
                off += n_
        codes = [A.L2NormFn.apply(c, 1e-8, 0) for c in global_codes]
        sp = sp.float()                             # (E1 hands over its last activation: bfloat16 in precision mode 1; the modulation reads fp32)
        g = codes[-1]
        ws = P("SpatialCodeModulation.scale.weight")
        inv = 1.0 / math.sqrt(ws.shape[1])
        scale = A.linear(g, ws, P("SpatialCodeModulation.scale.bias"), wscale=inv)
        shift = A.linear(g, P("SpatialCodeModulation.bias.weight"), P("SpatialCodeModulation.bias.bias"), wscale=inv)
        x = A.SpatialModFn.apply(sp, scale, shift)
        for i, (ci, co) in enumerate(HEAD_CH):
            q = "HeadResnetBlock%d." % i
            sw = q + "skip.Conv.weight"
            skip = x if ci == co else A.conv(x, P(sw), net, sw, scale=1.0 / math.sqrt(ci))
            r = self._styled_conv(x, q + "conv1.", g, "HeadResnetBlock%d.conv1" % i, noise)
            if TRAIN_FUSE["merge"]:
                x = self._styled_conv(r, q + "conv2.", g, "HeadResnetBlock%d.conv2" % i, noise, res=skip, out_scale=INV_SQRT2)
            else:
                x = A.AddScaleFn.apply(skip, self._styled_conv(r, q + "conv2.", g, "HeadResnetBlock%d.conv2" % i, noise), INV_SQRT2)
        feas = []
        if extract_features:
            feas.append(self._feat_head(x.detach().float(), "layer32.", 3))
        for j, (key, ci, co) in enumerate(UP):
            q = "UpsamplingResBlock%d." % key
            g = codes[-2 - j]
            if ci == co:
                skip = x
            else:
                sw = q + "skip.Conv.weight"
                skip = A.conv(x, P(sw), net, sw, bias=P(q + "skip.Act.bias"), scale=1.0 / math.sqrt(ci), act=A.LRELU)
            # (the x2 bilinear upsample of the skip is sampled on the fly by the merge pass, as in the inference path)
            up2 = TRAIN_FUSE["res_up2"] and TRAIN_FUSE["merge"]
            if not up2:
                skip = A.BilinearFn.apply(skip, 2 * skip.shape[1], 2 * skip.shape[2])
            r = self._styled_conv(x, q + "conv1.", g, "UpsamplingResBlock%d.conv1" % key, noise, upsample=True)
            if TRAIN_FUSE["merge"]:
                x = self._styled_conv(r, q + "conv2.", g, "UpsamplingResBlock%d.conv2" % key, noise, res=skip, out_scale=INV_SQRT2,
                                      res_up2=up2)
            else:
                x = A.AddScaleFn.apply(skip, self._styled_conv(r, q + "conv2.", g, "UpsamplingResBlock%d.conv2" % key, noise), INV_SQRT2)
            if extract_features:
                feas.append(self._feat_head(x.detach().float(), "layer%d." % (2 ** (j + 6)), 3 if j < 2 else 1))
        wr = P("ToRGB.conv.weight")
        brgb = P("ToRGB.conv.bias") + P("ToRGB.bias").reshape(-1)
        y = A.ToRGBConvFn.apply(x, wr, brgb, 1.0 / math.sqrt(wr.shape[1]))
        wl = P("ToRGB.epi1.style_mod.lin.weight")
        style = A.linear(codes[0], wl, P("ToRGB.epi1.style_mod.lin.bias"), wscale=wl.shape[1] ** -0.5)
        rgb = A.ToNCHWFn.apply(A.instance_norm(y, None, style=style))
        if not extract_features:
            return rgb
        h = feas[0].shape[1]
        feat = torch.cat([feas[0]] + [A.AvgPoolFn.apply(f, f.shape[1] // h) for f in feas[1:]], dim=3)
        feat1 = torch.cat([f if f.shape[1] == 256 else A.BilinearFn.apply(f, 256, 256) for f in feas], dim=3)
        for i in range(3):
            feat = self._residual_block(feat, "layert.%d." % i)
        feat1 = self._residual_block(feat1, "layert1.0.")
        feat1 = A.conv(feat1, P("layert1.1.weight"), net, "layert1.1.weight", bias=P("layert1.1.bias"))
        return rgb, feat, feat1

    # correspondence feature heads (generator.py:174-238): IN on the ReplicationPad2d(1)-padded tensor, conv, IN, PReLU, ...
    def _feat_head(self, x, p, k):
        net, P = self.G, self.G.p
        if k == 3:
            # InstanceNorm runs on the padded tensor (quirk of the reference); the conv then needs no padding:
            # run the zero-padded kernel on the padded canvas and crop its border
            xp = A.PadFn.apply(x, 1, A.REPLICATE)
            n = A.instance_norm(xp)
            y = A.PadFn.apply(A.conv(n, P(p + "2.weight"), net, p + "2.weight", bias=P(p + "2.bias")), -1, A.Z)
        else:
            y = A.conv(A.instance_norm(x), P(p + "2.weight"), net, p + "2.weight", bias=P(p + "2.bias"))
        y = A.instance_norm(y, prelu=P(p + "4.weight"), act=A.PRELU)
        y = A.conv(y, P(p + "6.weight"), net, p + "6.weight", bias=P(p + "6.bias"), pad_mode=A.REPLICATE if k == 3 else A.Z)
        return A.instance_norm(y, prelu=P(p + "8.weight"), act=A.PRELU)

    def _residual_block(self, x, p):
        net, P = self.G, self.G.p
        a = P(p + "prelu.weight")
        y, st = A.conv(x, P(p + "conv1.weight"), net, p + "conv1.weight", bias=P(p + "conv1.bias"), pad_mode=A.REPLICATE, stats=True)
        y = A.instance_norm(y, st, prelu=a, act=A.PRELU)
        y, st = A.conv(y, P(p + "conv2.weight"), net, p + "conv2.weight", bias=P(p + "conv2.bias"), pad_mode=A.REPLICATE, stats=True)
        return A.PReluResFn.apply(A.instance_norm(y, st), x, a)

    # ------------------------------------------------------------ correspondence (ppst_model.py:330-387)
    def rselfcorr(self, fea1):
        return A.RSelfCorrFn.apply(fea1)

    def corrm(self, fea, fea0):
        """fea (keys) / fea0 (queries): NHWC (B,64,64,512) -> (B,4096,4096)."""
        mk = int(getattr(self.model.opt, "match_kernel", 1))
        if mk < 1 or mk % 2 == 0:
            raise ValueError("match_kernel must be odd (got %d)" % mk)
        return A.CorrMFn.apply(fea, fea0, mk)

    def warp_mask(self, mask, corr):
        """PPSTModel.warp of the one-hot mask (no gradient to the mask)."""
        s = int(((mask.shape[2] * mask.shape[3]) / corr.shape[1]) ** 0.5)
        patches = ops.unfold_patches(mask, s)
        b, c, h, w = mask.shape
        return A.FoldFn.apply(A.GemmConstBFn.apply(corr, patches), c, h, w, s)

    def warp_image(self, x, corr):
        """PPSTModel.warp (ppst_model.py:366-387) of an image-sized tensor: unfold into the s x s patches of the correspondence
        grid, corr @ patches, fold.  Differentiable in ``corr`` and -- when it carries a gradient (the second warp of the
        Cycwarp branch) -- in ``x``."""
        b, c, h, w = x.shape
        s = int(((h * w) / corr.shape[1]) ** 0.5)
        if x.requires_grad:
            patches = A.UnfoldFn.apply(x, s)
            out = A.WarpGemmFn.apply(corr, patches, patches.shape[2])
        else:
            out = A.GemmConstBFn.apply(corr, ops.unfold_patches(x, s))
        return A.FoldFn.apply(out, c, h, w, s)

    # ------------------------------------------------------------ losses (ppst_model.py:161-235)
    def gan_logits(self, img):
        return A.DiscriminatorLogitsFn.apply(img, self.d_trainer)

    def begin_backward_overlap(self):
        """Arm the per-network gradient all-reduces that fire from inside backward() (world > 1)."""
        if self.world > 1:
            self._install_overlap_hooks()
            self._pending = {}
            for k in self._done_count:
                self._done_count[k] = 0

    def compute_generator_losses(self, real, mask=None):
        """PPSTModel.compute_generator_losses (ppst_model.py:161-235; ``model(real, None, None, mask,
        command="compute_generator_losses")`` lands here): the loss / metric tensors carry the autograd graph."""
        opt, m = self.opt, self.model
        if torch.is_grad_enabled():
            self.begin_backward_overlap()
        lam = lambda k, d: float(getattr(opt, k, d))
        stage = int(getattr(opt, "training_stage", 2))
        B = real.shape[0]
        noise = m.noise if isinstance(m.noise, dict) else None
        losses, metrics = {}, {}
        sp = self.encoder_con(real)
        gl, _ = self.encoder_col(real)
        if stage == 2:
            _, feas, feas1 = self.generator(sp, gl, noise, extract_features=True)
            sps = torch.cat((feas, self.rselfcorr(feas1)), dim=3)
            corr = self.corrm(sps, glue.swap(sps))
            corr_self = self.corrm(sps, sps)
            _, gl = self.encoder_col(real, corrmatrix=corr_self)
            if lam("lambda_StyleCon", 1.0) > 0.0:
                _, pro_ms, gl_w, pro_mw = self.encoder_col(real, mask=mask, corrmatrix=corr)
            if lam("lambda_Cycwarp", 0.0) > 0.0:
                # ppst_model.py:175-179: warp the image to the partner and back, compare with a perceptual metric.  The
                # reference's metric is lpips.LPIPS(net='alex') (:61); its weights ship with neither the reference nor this
                # image, so the metric is INJECTED: model.perceptual_metric(image_rec, real) -> tensor, any differentiable
                # callable (parity of the LPIPS term itself is unpinned; the double warp and its backward are tested)
                metric = getattr(m, "perceptual_metric", None)
                if metric is None:
                    raise RuntimeError("lambda_Cycwarp > 0 needs model.perceptual_metric (the reference uses lpips.LPIPS(net='alex'), "
                                       "whose weights are not available here): set it to a differentiable callable or lambda_Cycwarp = 0")
                image_warp = self.warp_image(real, corr)
                image_rec = self.warp_image(image_warp, glue.swap(corr))
                losses["image_warp_reg"] = metric(image_rec, real) * lam("lambda_Cycwarp", 0.0)
            if lam("lambda_Maskwarp", 10.0) > 0.0:
                losses["Mask_warp"] = A.L1LossFn.apply(self.warp_mask(mask, corr), glue.swap(mask), lam("lambda_Maskwarp", 10.0))
        style_con = lam("lambda_StyleCon", 1.0) > 0.0
        both = None
        if style_con:
            # rec = G(sp, gl) and mix = G(swap(sp), gl_w) (ppst_model.py:182, :191) are independent: ONE generator pass over the
            # batch of 2B (the same arithmetic per image; fuller grids and half the launches at the reference's B = 2), and the two
            # masked E2 passes on them (:192-193) and the two D passes (:224-233) likewise
            nz2 = {k: torch.cat((v, v), 0) for k, v in noise.items()} if noise is not None else None
            both = self.generator(torch.cat((sp, glue.swap(sp)), 0), [torch.cat((a, b), 0) for a, b in zip(gl, gl_w)], nz2)
            rec, mix = both[:B], both[B:]
        else:
            rec = self.generator(sp, gl, noise)
        if lam("lambda_L1", 3.0) > 0.0:
            losses["G_L1"] = A.L1LossFn.apply(rec, real, lam("lambda_L1", 3.0))
        if style_con:
            _, pro_32m, _, _ = self.encoder_col(both, mask=torch.cat((mask, glue.swap(mask)), 0))
            pro_2m, pro_3m = [p_[:B] for p_ in pro_32m], [p_[B:] for p_ in pro_32m]
            sp_3 = self.encoder_con(mix)
            nz = {k: v[:B // 2] for k, v in noise.items()} if noise is not None else None
            cyc = self.generator(glue.swap(sp_3)[:B // 2], [g[:B // 2] for g in gl], nz)
            metrics["L1_dist"] = A.L1LossFn.apply(cyc, real[:B // 2].contiguous(), 1.0)
            losses["G_L1_cyc"] = A.L1LossFn.apply(cyc, real[:B // 2].contiguous(), 3.0)
            s1 = s2 = None
            pending = []
            for lid in range(0, 12, 3):
                li = lid // 3
                key0 = torch.cat(pro_ms[lid:lid + 3], 0).detach()
                keyw = torch.cat(pro_mw[lid:lid + 3], 0).detach()
                query, query_r = torch.cat(pro_3m[lid:lid + 3], 0), torch.cat(pro_2m[lid:lid + 3], 0)
                queue = getattr(m.criterionNCE, "queue_data_A%d" % li)
                a = A.RsclLossFn.apply(query, keyw, key0, queue, lam("nce_T", 0.07))
                b = A.RsclLossFn.apply(query_r, key0, keyw, queue, lam("nce_T", 0.07))
                s1 = a if s1 is None else s1 + a
                s2 = b if s2 is None else s2 + b
                pending.append((torch.cat((key0[0:3], keyw[0:3]), 0), li))
            m.criterionNCE.enqueue_all(pending)
            lsc = lam("lambda_StyleCon", 1.0)
            losses["G_styleContmix"] = s1 if lsc == 1.0 else s1 * lsc
            losses["G_styleContrec"] = s2 if lsc == 1.0 else s2 * lsc
        if lam("lambda_GAN", 1.0) > 0.0:
            if style_con:
                pred = self.gan_logits(both)
                losses["G_GAN_rec"] = A.LsganFn.apply(pred[:B], 1.0, 0.5 * lam("lambda_GAN", 1.0))
                losses["G_GAN_mix"] = A.LsganFn.apply(pred[B:], 1.0, lam("lambda_GAN", 1.0))
            else:
                losses["G_GAN_rec"] = A.LsganFn.apply(self.gan_logits(rec), 1.0, 0.5 * lam("lambda_GAN", 1.0))
        return losses, metrics

    # ------------------------------------------------------------ one generator iteration
    def zero_grad(self):
        for f in self.fp.values():
            f.zero_grad()

    def losses_and_grads(self, real, mask=None):
        self.zero_grad()
        with torch.enable_grad():
            losses, metrics = self.compute_generator_losses(real, mask)
            total = None
            for v in losses.values():
                total = v if total is None else total + v
            total.backward()
        out = {k: v.detach() for k, v in losses.items()}
        out.update({k: v.detach() for k, v in metrics.items()})
        return out

    # ---- data parallel: flat gradient all-reduce per network, overlapped with the rest of the backward pass.
    # A network's flat gradient is final once every contribution of this backward has landed.  Contributions are counted per
    # network: a post-accumulate hook fires once for each parameter whose gradient still travels through autograd's
    # AccumulateGrad, and every backward kernel that adds straight into the flat buffer reports itself (autograd._noted).  The
    # count of one full backward is LEARNED in the first armed step (whose all-reduces start after backward() returns); from
    # then on the all-reduce of a network is launched (async, RCCL) by the contribution that completes its count, while
    # autograd is still working on the networks behind it (backward order: D -> G -> E2 / E1).  The graph of the generator
    # iteration is static for fixed options; a step whose count differs from the learned one raises (call relearn_overlap()
    # after changing loss weights / training stage).
    def _install_overlap_hooks(self):
        if getattr(self, "_hooks", None) is not None:
            return
        self._hooks, self._pending, self._done_count = [], {}, {}
        self._expected = {}
        for key, f in self.fp.items():
            self._done_count[key] = 0

            def event(key=key, f=f):
                self._done_count[key] += 1
                if self.world > 1 and self._expected.get(key) == self._done_count[key] and key not in self._pending:
                    import torch.distributed as dist
                    self._pending[key] = dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, async_op=True)
            f.on_event = event
            for p in f.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, ev=event: ev()))

    def relearn_overlap(self):
        if getattr(self, "_expected", None) is not None:
            self._expected = {}

    def all_reduce(self):
        """Finish the gradient average: wait for the all-reduces launched during backward, launch + wait the rest."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        pending = getattr(self, "_pending", {})
        counts = getattr(self, "_done_count", None)
        if counts is not None and any(self._expected.get(k) is None for k in self.fp):
            # the learning step: every rank must have seen the same number of contributions per network -- a rank that learned
            # another count would launch its all-reduces at another point of backward, and the collectives of one communicator
            # have to be issued in the same order everywhere.  One small all_gather, once (and after relearn_overlap()).
            keys = list(self.fp)
            mine = torch.tensor([counts[k] for k in keys], dtype=torch.int64, device=self.fp[keys[0]].grad.device)
            # (the communicator's size, not self.world: the one-GPU rehearsal of bench.py tells the optimizer world = 2 on a 1-rank group)
            got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(got, mine)
            if any(not torch.equal(g_, got[0]) for g_ in got):
                raise RuntimeError("gradient contribution counts differ between ranks (%s per rank for %s): the overlapped "
                                   "all-reduces would be issued in different orders" % ([g_.tolist() for g_ in got], keys))
        for key, f in self.fp.items():
            if counts is not None:
                exp = self._expected.get(key)
                if exp is None:
                    self._expected[key] = counts[key]           # learned: the next step overlaps
                elif exp != counts[key]:
                    raise RuntimeError("gradient contributions of %s changed (%d, learned %d): the graph of the generator iteration "
                                       "is not the one the overlap was learned on -- call relearn_overlap()" % (key, counts[key], exp))
            h = pending.pop(key, None)
            if h is None:
                h = dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, async_op=True)
            h.wait()
            f.grad.div_(self.world)

    def adam(self):
        for k in ("G", "E2", "E1"):          # optimizer_G.step(); optimizer_E2.step(); optimizer_E1.step()
            self.fp[k].adam()

    def train_step(self, real, mask=None):
        losses = self.losses_and_grads(real, mask)
        self.all_reduce()
        self.adam()
        return losses


class FlatAdam:
    """torch.optim.Adam of one network for the restated PPSTOptimizer: ``zero_grad()`` / ``step()`` on the network's flat
    parameter / gradient / moment buffers (ONE launch).  ``step()`` first completes the data-parallel gradient average that
    DistributedDataParallel performs inside ``backward()`` for the reference (models/__init__.py:88)."""

    def __init__(self, flat, reduce=None):
        self.flat, self.reduce = flat, reduce

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        if self.reduce is not None:
            self.reduce()
        self.flat.adam()


class PPSTOptimizer:
    """optimizers/ppst_optimizer.py:PPSTOptimizer restated line by line on the model facade: the D / G alternation of
    ``train_one_step`` (:60-71 -- the first call is a discriminator iteration, the mode names are swapped in the reference),
    ``model(..., command=...)`` -> ``sum(v.mean())`` -> ``.backward()`` -> ``optimizer.step()`` for G, E2, E1 (:73-94) and for D
    with the lazy R1 penalty every ``R1_once_every`` discriminator iterations (:96-130; lazy-regularisation corrected lr /
    betas :46-49) and ``D_total`` (:127).  ``data_i`` = {"real_A": (B,3,H,W), "mask_A": (B,3,H,W) one-hot}.  Gradients are
    averaged over ranks with one flat all-reduce per network (RCCL): G / E2 / E1 overlapped with backward, D overlapped with the
    generator iteration that follows (DiscriminatorTrainer.step_deferred)."""

    def __init__(self, model, lr=1e-3, beta1=0.0, beta2=0.99, R1_once_every=16, world=1):
        self.model, self.opt = model, model.opt
        self.train_mode_counter = 0
        self.discriminator_iter_counter = 0
        self.R1_once_every = R1_once_every
        self.gen = GeneratorTrainer.for_model(model, lr=lr, beta1=beta1, beta2=beta2, world=world)
        self.gen.world = world
        for f in self.gen.fp.values():
            f.lr, f.b1, f.b2 = lr, beta1, beta2
        self.dis = self.gen.d_trainer
        # the generator-side all-reduces are launched from inside backward(); the first step() of an iteration completes them
        self.optimizer_G = FlatAdam(self.gen.fp["G"], reduce=self.gen.all_reduce)
        self.optimizer_E2 = FlatAdam(self.gen.fp["E2"])
        self.optimizer_E1 = FlatAdam(self.gen.fp["E1"])
        self.optimizer_D = None
        if self.dis is not None:
            c = R1_once_every / (1 + R1_once_every)
            self.dis.lr, self.dis.b1, self.dis.b2 = lr * c, beta1 ** c, beta2 ** c
            self.dis.R1_once_every, self.dis.world = R1_once_every, world
            self.optimizer_D = self.dis
        # DistributedDataParallel's constructor broadcasts rank 0's parameters and buffers (models/__init__.py:88): without it the
        # per-process torch.randn NCE queues (and any unseeded parameter) differ between the ranks
        import torch.distributed as dist
        if world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() == world:
            model.sync_from_rank0()

    def prepare_images(self, data_i):
        return data_i["real_A"], data_i["mask_A"]

    def toggle_training_mode(self):
        modes = ["discriminator", "generator"]
        self.train_mode_counter = (self.train_mode_counter + 1) % len(modes)
        return modes[self.train_mode_counter]

    def train_one_step(self, data_i, total_steps_so_far=0):
        images_minibatch, mask_minibatch = self.prepare_images(data_i)
        with ops.batch_aware():         # batch 2: the 64 x 64 layers would leave three quarters of the chip idle (ops.BATCH_AWARE)
            if self.toggle_training_mode() == "generator":
                losses = self.train_discriminator_one_step(images_minibatch, mask_minibatch)
            else:
                losses = self.train_generator_one_step(images_minibatch, mask_minibatch)
        return {k: float(v.float().mean()) for k, v in losses.items()}      # util.to_numpy

    def train_generator_one_step(self, images, mask):
        sp_ma, gl_ma = None, None
        self.optimizer_G.zero_grad()
        self.optimizer_E1.zero_grad()
        self.optimizer_E2.zero_grad()
        with torch.enable_grad():
            g_losses, g_metrics = self.model(images, sp_ma, gl_ma, mask, command="compute_generator_losses")
            g_loss = sum([v.mean() for v in g_losses.values()])
            g_loss.backward()
        self.optimizer_G.step()
        self.optimizer_E2.step()
        self.optimizer_E1.step()
        g_losses = {k: v.detach() for k, v in g_losses.items()}
        g_losses.update({k: v.detach() for k, v in g_metrics.items()})
        return {**g_losses}

    def train_discriminator_one_step(self, images, mask):
        if float(getattr(self.opt, "lambda_GAN", 1.0)) == 0.0 or self.dis is None:
            return {}
        self.discriminator_iter_counter += 1
        self.dis.iter_counter = self.discriminator_iter_counter
        self.optimizer_D.zero_grad()
        with torch.enable_grad():
            d_losses, d_metrics, sp, gl = self.model(images, mask, command="compute_discriminator_losses")
            self.previous_sp = sp.detach()
            self.previous_gl = [g.detach() for g in gl]
            d_loss = sum([v.mean() for v in d_losses.values()])
            d_loss.backward()
        lambda_R1 = float(getattr(self.opt, "lambda_R1", 10.0))
        needs_R1_at_current_iter = lambda_R1 > 0.0 and self.discriminator_iter_counter % self.R1_once_every == 0
        if needs_R1_at_current_iter:
            self.optimizer_D.all_reduce(); self.optimizer_D.adam()          # optimizer_D.step(): R1 needs the updated D at once
            d_losses.update(self._r1_backward(images))
        self.optimizer_D.step_deferred()                                     # optimizer_D.step(), all-reduce overlapped (world > 1)
        d_losses = {k: v.detach() for k, v in d_losses.items()}
        d_losses["D_total"] = sum([v.mean() for v in d_losses.values()])
        d_losses.update(d_metrics)
        return d_losses

    def _r1_backward(self, images):
        """ppst_optimizer.py:116-125 up to (not including) the optimizer step: zero_grad, R1 loss x R1_once_every, backward."""
        self.optimizer_D.zero_grad()
        with torch.enable_grad():
            r1_losses = self.model(images, command="compute_R1_loss")
            r1_loss = sum([v.mean() for v in r1_losses.values()])
            r1_loss = r1_loss * self.R1_once_every
            r1_loss.backward()
        return r1_losses

    def r1_iteration(self, images):
        """ONE lazy-R1 pass on its own, as it runs on every R1_once_every-th discriminator iteration (ppst_optimizer.py:116-126):
        zero_grad -> compute_R1_loss -> x R1_once_every -> backward -> optimizer_D.step().  bench.py times it to state the R1
        term of the train step when its timed region is shorter than 16 iterations."""
        self.dis.finish_pending()
        r1 = self._r1_backward(images)
        self.optimizer_D.all_reduce(); self.optimizer_D.adam()
        return {k: v.detach() for k, v in r1.items()}

    def save(self, total_steps_so_far):
        if self.dis is not None:
            self.dis.finish_pending()
        return self.model.save(total_steps_so_far)
