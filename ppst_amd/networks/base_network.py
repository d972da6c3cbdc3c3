"""Shared host-side machinery of the four HIP-backed networks.

``BaseNetwork`` plays the role of the reference's models/networks/base_network.py
(a torch Module built from ``opt``) but its parameter tree is generated from the
checkpoint key contract in ppst_amd/weights.py, so ``state_dict()`` has exactly
the reference's keys, shapes and order and
``load_state_dict`` accepts the authors' checkpoints.  Forward passes are
inference-only compositions of HIP kernels (ppst_amd/ops.py); tensors at the
class boundary are NCHW like the reference's, internally NHWC.
"""
import math

import torch
from torch import nn

from .. import ops, weights

SQRT2 = math.sqrt(2.0)
INV_SQRT2 = 1.0 / SQRT2


class _Node(nn.Module):
    """parameter container (no behaviour)."""


def to_nhwc(x):
    """NCHW tensor -> NHWC (B,H,W,C); zero-copy when x is a permuted NHWC tensor."""
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D NCHW tensor")
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        ops._chk(v, "input")
        return v
    return ops.nchw_to_nhwc(x)


def as_nchw(x_nhwc):
    """NHWC storage exposed with NCHW shape (a view: values identical to the reference's
    NCHW tensor, memory stays channels-last so the next network does not transpose)."""
    return x_nhwc.permute(0, 3, 1, 2)


class BaseNetwork(nn.Module):
    prefix = ""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt=None, size=512, seed=0):
        super().__init__()
        self.opt = opt
        self._flat = {}
        self._cache = {}
        specs = [s for s in weights.param_specs(size=size) if s[0].startswith(self.prefix)]
        for name, shape, kind, aux in specs:
            parts = name[len(self.prefix):].split(".")
            node = self
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            t = weights._draw(name, shape, kind, aux, seed, 0.0, 0.0)
            if kind in (weights.BLUR3, weights.BLUR4, weights.UP4):
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))

    # -- parameter access ----------------------------------------------------
    def p(self, name):
        t = self._flat.get(name)
        if t is None:
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                node = node._modules[part]
            t = node._parameters.get(parts[-1])
            if t is None:
                t = node._buffers[parts[-1]]
            self._flat[name] = t
        return t

    def _apply(self, fn, *a, **k):
        self._flat.clear()
        self._cache.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._flat.clear()
        self._cache.clear()
        return super().load_state_dict(*a, **k)

    def cached(self, key, params, build):
        """memoise derived data (packed weights, folded biases) per parameter version."""
        ver = tuple((id(t), t._version, t.data_ptr()) for t in params) + (ops.PRECISION["value"],)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        val = build()
        self._cache[key] = (ver, val)
        return val

    def plan(self, wname, kind="conv", scale=1.0):
        w = self.p(wname)
        return self.cached(("plan", wname, kind), [w], lambda: ops.ConvPlan(w, kind=kind, scale=scale))

    def refresh_plans(self):
        """After an in-place parameter update (the Adam kernel writes the flat buffer through a raw pointer: no version counter
        moves): re-pack every cached ConvPlan from its parameter storage with two batched launches (ops.repack_plans) instead of
        dropping the plans and rebuilding each with its own pack launch on next use (216 launches per train step); every other
        derived value (folded biases, style tables) is dropped as before."""
        plans, keep = [], {}
        for key, (ver, val) in self._cache.items():
            if isinstance(key, tuple) and key and key[0] == "plan" and isinstance(val, ops.ConvPlan) and val.precision != 2 \
                    and val.wparam.data_ptr() == self.p(key[1]).data_ptr():
                plans.append(val)
                keep[key] = (ver, val)
        self._cache.clear()
        if not plans:
            return
        sig = tuple((id(pl), tuple(sorted(map(str, pl._packs)))) for pl in plans)
        hit = self.__dict__.get("_repack")
        if hit is not None and hit[0] == sig:
            ops.run_repack(hit[1])
        else:
            self.__dict__["_repack"] = (sig, ops.repack_plans(plans), plans)     # (plans held: ids stay unique)
        self._cache.update(keep)

    def print_architecture(self, verbose=False):
        n = sum(p.numel() for p in self.parameters())
        print("[Network %s] Total number of parameters : %.3f M" % (type(self).__name__, n / 1e6))

    # -- ConvLayer / ResBlock of stylegan2_layers.py:497-579 on NHWC ---------
    def from_rgb(self, x, p, out_dtype=torch.float32):
        """ConvLayer(3, C, 1): 1x1 conv (no bias) + FusedLeakyReLU (HBM-bound kernel).  out_dtype: storage type the network's
        trunk runs in (half-precision activation storage, ops.HALF_STORE: E1 / E2 pass ops.act_dtype(); every kernel downstream
        keeps its input's type)."""
        w = self.p(p + "Conv.weight")
        return ops.conv1x1_small_cin(x, w, self.p(p + "Act.bias"), 1.0 / math.sqrt(w.shape[1]), ops.ACT_LRELU, out_dtype=out_dtype)

    def _norm_act(self, y, st, count, act_bias=None, act=ops.ACT_NONE, **kw):
        ss = ops.in_finalize(st, count, post_bias=act_bias)
        return ops.affine_act(y, ss, act=act, **kw), ss

    def res_block(self, x, p, blur_pad_mode, norm):
        """ResBlock(cin, cout, blur, downsample) -> (conv2(conv1(x)) + skip(x)) / sqrt2.
        blur_pad_mode: PAD_REFLECT for the encoders' main branch (reflection_pad=True),
        PAD_ZERO for the discriminator; the skip branch always zero-pads (:566-568)."""
        B, H, W, cin = x.shape
        k = self.p(p + "conv2.Blur.kernel")
        ks = k.shape[0]
        w1, w2, ws = self.p(p + "conv1.Conv.weight"), self.p(p + "conv2.Conv.weight"), self.p(p + "skip.Conv.weight")
        sc1 = 1.0 / math.sqrt(cin * 9)
        scs = 1.0 / math.sqrt(cin)
        conv_pad = ops.PAD_REFLECT if blur_pad_mode == ops.PAD_REFLECT else ops.PAD_ZERO
        pad_c = (ks - 2) + 2          # conv2: (len(k)-2)+(3-1)   (stylegan2_layers.py:515)
        pad_s = (ks - 2) + 0          # skip:  (len(k)-2)+(1-1)
        # skip: blur (zero pad) keeping every 2nd sample == blur then 1x1 stride-2 conv
        xs, _ = ops.blur_nhwc(x, self.p(p + "skip.Blur.kernel"), (pad_s + 1) // 2, pad_s // 2, ops.PAD_ZERO, down=2)
        if not norm:
            y1 = self.plan(p + "conv1.Conv.weight", scale=sc1)(x, bias=self.p(p + "conv1.Act.bias"), act=ops.ACT_LRELU,
                                                                pad_mode=conv_pad)
            skip = self.plan(p + "skip.Conv.weight", scale=scs)(xs)
            xb, bhw = ops.blur_nhwc(y1, k, (pad_c + 1) // 2, pad_c // 2, blur_pad_mode, s2d=True)
            ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
            return self.plan(p + "conv2.Conv.weight", "s2d", sc1)(
                xb, bias=self.p(p + "conv2.Act.bias"), act=ops.ACT_LRELU, residual=skip, res_after_act=True,
                out_scale=INV_SQRT2, out_hw=ohw)
        # norm == 'in' (E1): conv -> InstanceNorm -> FusedLeakyReLU (:542-549)
        y1, st1 = self.plan(p + "conv1.Conv.weight", scale=sc1)(x, stats=True, pad_mode=conv_pad)
        ss1 = ops.in_finalize(st1, H * W, post_bias=self.p(p + "conv1.Act.bias"))
        ys, sts = self.plan(p + "skip.Conv.weight", scale=scs)(xs, stats=True)
        # the norm + leaky-relu of conv1 is applied by the blur while it reads (no apply pass)
        xb, bhw = ops.blur_nhwc(y1, k, (pad_c + 1) // 2, pad_c // 2, blur_pad_mode, s2d=True, in_ss=ss1, in_act=ops.ACT_LRELU)
        ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
        y2, st2 = self.plan(p + "conv2.Conv.weight", "s2d", sc1)(xb, stats=True, out_hw=ohw)
        cnt = ohw[0] * ohw[1]
        ss_s = ops.in_finalize(sts, cnt)
        ss2 = ops.in_finalize(st2, cnt, post_bias=self.p(p + "conv2.Act.bias"))
        return ops.affine_act(y2, ss2, res=ys, res_scale_shift=ss_s, act=ops.ACT_LRELU, out_scale=INV_SQRT2)
