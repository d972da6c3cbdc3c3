"""Differentiable building blocks of the generator / encoder update.

Each class is a ``torch.autograd.Function`` whose forward AND backward are HIP kernels of
libppst_hip.so (ppst_amd/ops.py); torch's autograd engine is used as the tape only -- what the
reference gets from ``g_loss.backward()`` (optimizers/ppst_optimizer.py:88) through F.conv2d /
F.conv_transpose2d / InstanceNorm2d / StyleMod / upfirdn2d / F.interpolate / pooling.  Activations
are NHWC fp32 (B,H,W,C) like the inference path.  Views, cat and slicing between the blocks are
torch memory operations.
"""
import math

import torch
from torch.autograd import Function

from . import gates, ops

SQRT2 = math.sqrt(2.0)
Z, REFLECT, REPLICATE = ops.PAD_ZERO, ops.PAD_REFLECT, ops.PAD_REPLICATE
NONE, LRELU, PRELU = ops.ACT_NONE, ops.ACT_LRELU, ops.ACT_PRELU


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _gl(g, like):
    """an incoming gradient, contiguous and in the STORAGE TYPE of the tensor it belongs to (round 5: the training activations of
    precision mode 1 are bfloat16; a gradient that arrives from an fp32 island -- a loss, the 3-channel image, a pooled vector -- is
    rounded once here)"""
    g = g if g.dtype == like.dtype else g.to(like.dtype)
    return g if g.is_contiguous() else g.contiguous()


def lrelu_bwd(g, out, scale=1.0):
    """d/d(pre-activation) of y = lrelu(x)*sqrt2 given d/dy, gated by the sign of the saved output
    (fused_bias_act_kernel.cu:43)."""
    return ops.fused_bias_act_raw(g, None, gates.sign_gate(out, "lrelu"), 3, 1, 0.2, SQRT2 * scale)


def relu_gate(g, x):
    """g * [x > 0] (backward of the projectors' leading nn.ReLU)."""
    return ops.fused_bias_act_raw(g, None, gates.sign_gate(x, "relu"), 3, 1, 0.0, 1.0)


def prelu_bwd(g, y, prelu, scale_shift=None, res=None):
    """ops.prelu_bwd with the gate tape: (g * prelu'(z), dslope), z = a*y + s [+ res].  Record / replay need z's sign, which the
    kernel forms internally: those two modes recompute z with the apply kernel and gate with the recorded decision (the slope
    gradient sum g*z*[z<0] keeps the run's own z: a flipped element has z ~ 0)."""
    if gates.MODE["value"] is None:
        return ops.prelu_bwd(g, y, prelu, scale_shift=scale_shift, res=res)
    z = ops.affine_act(y, scale_shift, res=res, res_before_act=True) if (scale_shift is not None or res is not None) else y
    gate = gates.sign_gate(z, "prelu", ge=True)
    gpre, ds = ops.prelu_bwd(g, y, prelu, scale_shift=scale_shift, res=res)
    if gates.MODE["value"] == "replay":
        gpre = ops.fused_bias_act_raw(g, None, gate, 3, 1, float(prelu), 1.0)      # g * (gate > 0 ? 1 : slope)
    return gpre, ds


# ---- parameter gradients go straight into the trainer's flat gradient buffer -------------------------------------------
# FlatParams / DiscriminatorTrainer bind ``p.grad`` of every parameter to a view of ONE flat buffer per network and mark the
# parameter (``_ppst_direct``).  A block's backward then lets the producing kernel add into that view (the ``accumulate`` flag
# of the C ABI) and returns None for the parameter: no temporary, no zero fill and no AccumulateGrad ``add`` launch per
# (parameter, use) -- 866 + 320 launches per train step before.  ``_ppst_on_grad`` tells the trainer that one more
# contribution has landed (it counts them to launch a network's gradient all-reduce from inside backward()).
# Consequence: EVERY pass through a block's backward -- also torch.autograd.grad(..., inputs=[some activation]) -- adds that
# block's parameter gradients to the flat buffer, exactly as a second .backward() would.  The reference's pattern (zero_grad ->
# one backward -> step, ppst_optimizer.py:73-94) is what the trainers run; ``DIRECT["value"] = False`` restores returned
# gradients (autograd accumulates them) for code that walks a graph more than once.
DIRECT = {"value": True}
# round 5: LinearFn's backward in two or three launches instead of seven (ReLU gates, relu(x) and the bias column sums folded into the
# two gradient kernels); off while the gate tape records / replays (its gates go through gates.sign_gate)
FUSE_LINEAR = {"value": True}


def _direct(p):
    if not DIRECT["value"] or p is None or not getattr(p, "_ppst_direct", False) or not p.requires_grad:
        return None
    return p.grad


def _noted(p):
    cb = getattr(p, "_ppst_on_grad", None)
    if cb is not None:
        cb()


def _flipped(net, kname):
    """the blur taps of ``kname`` flipped (the adjoint FIR, upfirdn2d.py:116-121), kept per network: one flip + copy per train step
    and blur was 69 torch launches"""
    build = lambda: torch.flip(net.p(kname), [0, 1]).contiguous()
    cached = getattr(net, "cached", None)          # (BaseNetwork; a bare parameter holder of the tests has none)
    return cached(("flip", kname), [net.p(kname)], build) if cached is not None else build()


def _conv_grads(net, wname, kind, scale, pad_mode, x, gpre, need_x, need_w, out_hw=None, dw_out=None, want_bias=False, bias_out=None):
    """(dx, dw, db) of a stride-1 'conv' plan given the gradient at its raw output (``dw_out``: add dw into this view;
    ``want_bias``: the column sums of gpre come out of the weight-gradient kernel's own staging pass, into ``bias_out`` if given)."""
    plan = net.plan(wname, kind, scale)
    k = plan.k
    dx = dw = db = None
    acc = dw_out is not None
    bkw = dict(want_bias=True, bias_out=bias_out, bias_accumulate=bias_out is not None) if (want_bias and need_w) else {}

    def wg(xx, gg):
        r = ops.conv_wgrad(plan, xx, gg, out=dw_out, accumulate=acc, **bkw)
        return r if bkw else (r, None)
    if pad_mode == Z or k == 1:
        if need_x:
            dx = net.plan(wname, "dgrad", scale)(gpre)
        if need_w:
            dw, db = wg(x, gpre)
        return dx, dw, db
    # reflection / replication padding: y = conv_valid(pad(x)).  On the padded canvas the same zero-padded kernels
    # are exact: the gradient canvas is zero on the border, so border outputs / out-of-canvas taps contribute nothing.
    gp = ops.pad2d(gpre, 1, 1, 1, 1, Z)
    if need_x:
        dx = ops.pad2d_bwd(net.plan(wname, "dgrad", scale)(gp), 1, 1, 1, 1, pad_mode)
    if need_w:
        dw, db = wg(ops.pad2d(x, 1, 1, 1, 1, pad_mode), gp)      # (the zero border of gp adds nothing to the column sums)
    return dx, dw, db


class ConvFn(Function):
    """y = act(conv(x; w*scale) [+ noise_w*noise] [+ bias]) -> (y, tile statistics of y).
    kind 'conv': k in {1,3}, stride 1, padding k//2 in ``pad_mode``; 'convT': the fused 4x4 stride-2 transposed conv
    of EqualizedConv2d (stylegan2_layers.py:312-321)."""

    @staticmethod
    def forward(ctx, x, w, bias, noise_w, noise, net, wname, kind, scale, pad_mode, act, noise_w_host=None, bias_params=None,
                gate_downstream=False):
        """``bias_params``: leaf parameters whose sum is ``bias`` (StyledConv's three biases, summed by the caller outside the
        graph): their gradients are written directly, ``bias`` itself is then a constant.
        ``gate_downstream`` (act LRELU): the ONE consumer of y is an InstanceNormFn built with ``post_gate=True`` whose backward
        returns the gradient already multiplied by lrelu'(y) (ppst_in_bwd_apply's post gate: StyledConv's order is conv ->
        activation -> norm) -- this node then takes its upstream as the pre-activation gradient and runs no gate pass."""
        x = _c(x)
        plan = net.plan(wname, kind, scale)
        # the kernel takes the noise weight by value: the caller's cached host copy, or (a stream sync) the tensor itself
        nw = (noise_w_host if noise_w_host is not None else float(noise_w)) if noise_w is not None else 0.0
        y, st = plan(x, bias=bias, noise=(noise if noise_w is not None else None), noise_weight=nw, act=act, pad_mode=pad_mode, stats=True)
        ctx.save_for_backward(x, y if (act != NONE and not gate_downstream) else None, noise if noise_w is not None else None)
        ctx.cfg = (net, wname, kind, scale, pad_mode, act, bias is not None, noise_w is not None)
        ctx.gate_downstream = bool(gate_downstream and act == LRELU)
        ctx.refs = (w, bias, noise_w, bias_params)          # leaf parameters (for their flat-gradient views), not saved tensors
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)        # no zero tensor is built for the statistics output's (absent) gradient
        return y, st

    @staticmethod
    def backward(ctx, g, _gst):
        x, y, noise = ctx.saved_tensors
        net, wname, kind, scale, pad_mode, act, has_b, has_n = ctx.cfg
        w, bias, noise_w, bias_params = ctx.refs
        g = _gl(g, x)
        gpre = lrelu_bwd(g, y) if (act == LRELU and not ctx.gate_downstream) else g
        C = gpre.shape[3]
        db = dnw = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = bias_params is not None or (has_b and ctx.needs_input_grad[2])
        # the bias gradient = column sums of gpre: from the weight-gradient kernel's staging pass when that kernel runs on gpre
        # anyway (kind 'conv'; the transposed conv's weight gradient has gpre as its OTHER operand), else its own pass
        fuse_b = need_b and need_w and kind == "conv"
        bdst = _direct(bias) if (need_b and bias_params is None) else None
        if need_b and not fuse_b:
            v = ops.colsum(gpre.view(-1, C), out=bdst, accumulate=bdst is not None)
        if has_n and ctx.needs_input_grad[3]:
            dst = _direct(noise_w)
            dnw = ops.noise_wgrad(gpre, noise, out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(noise_w); dnw = None
        dstw = _direct(w) if need_w else None
        if kind == "conv":
            dx, dw, vb = _conv_grads(net, wname, kind, scale, pad_mode, x, gpre, need_x, need_w, dw_out=dstw, want_bias=fuse_b, bias_out=bdst)
            if fuse_b:
                v = vb
        else:  # convT: dX = stride-2 4x4 conv of dY (run over its space-to-depth copy); dW through the blurred 4x4 kernel
            gs = ops.space_to_depth(gpre)
            pl = net.plan(wname, "dgradT", scale)
            dx = pl(gs) if need_x else None
            dw = None
            if need_w:
                dw4 = ops.conv_wgrad(pl, gs, x)
                dw = ops.upscale_weight_bwd(dw4, pl.cin, pl.cout, pl.fwd_scale, out=dstw, accumulate=dstw is not None)
        if dstw is not None:
            _noted(w); dw = None
        if need_b:
            if bias_params is not None:       # one column sum, added to each of the summed biases' gradients in ONE launch
                torch._foreach_add_([_direct(q).view(-1) for q in bias_params], [v] * len(bias_params))
                for q in bias_params:
                    _noted(q)
            elif bdst is not None:
                _noted(bias)
            else:
                db = v
        return dx, dw, db, dnw, None, None, None, None, None, None, None, None, None, None


def conv(x, w, net, wname, bias=None, kind="conv", scale=1.0, pad_mode=Z, act=NONE, noise_w=None, noise=None, stats=False,
         noise_w_host=None, bias_params=None, gate_downstream=False):
    y, st = ConvFn.apply(x, w, bias, noise_w, noise, net, wname, kind, scale, pad_mode, act, noise_w_host, bias_params, gate_downstream)
    return (y, st) if stats else y


class BlurConvFn(Function):
    """ConvLayer(downsample=True) of stylegan2_layers.py:497-555 for the 3x3 conv: Blur (upfirdn2d, zero or reflection
    padding (p0, p1)) -> 3x3 stride-2 conv (no padding) [+ bias -> leaky-relu*sqrt2] -> (y, tile statistics)."""

    @staticmethod
    def forward(ctx, x, w, bias, net, wname, kname, scale, p0, p1, pad_mode, act):
        x = _c(x)
        k = net.p(kname)
        xb, bhw = ops.blur_nhwc(x, k, p0, p1, pad_mode, s2d=True)
        ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
        y, st = net.plan(wname, "s2d", scale)(xb, bias=bias, act=act, out_hw=ohw, stats=True)
        ctx.save_for_backward(xb, y if act != NONE else None)
        ctx.cfg = (net, wname, kname, scale, p0, p1, pad_mode, act, bhw, bias is not None)
        ctx.refs = (w, bias)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)        # no zero tensor is built for the statistics output's (absent) gradient
        return y, st

    @staticmethod
    def backward(ctx, g, _gst):
        xb, y = ctx.saved_tensors
        net, wname, kname, scale, p0, p1, pad_mode, act, bhw, has_b = ctx.cfg
        g = _gl(g, xb)
        gpre = lrelu_bwd(g, y) if act == LRELU else g
        C = gpre.shape[3]
        w, bias = ctx.refs
        db = dw = None
        need_b, need_w = has_b and ctx.needs_input_grad[2], ctx.needs_input_grad[1]
        bdst = _direct(bias) if need_b else None
        if need_w:
            dst = _direct(w)
            r = ops.conv_wgrad(net.plan(wname, "s2d", scale), xb, gpre, out=dst, accumulate=dst is not None,
                               want_bias=need_b, bias_out=bdst, bias_accumulate=bdst is not None)
            dw, db = r if need_b else (r, None)        # the bias gradient rides on the weight-gradient kernel's staging pass
            if dst is not None:
                _noted(w); dw = None
        elif need_b:
            db = ops.colsum(gpre.view(-1, C), out=bdst, accumulate=bdst is not None)
        if need_b and bdst is not None:
            _noted(bias); db = None
        dx = None
        if ctx.needs_input_grad[0]:
            d_xb = ops.dgrad_s2d(net, wname, scale, gpre, bhw)
            k = net.p(kname)
            ks = k.shape[0]
            kf = _flipped(net, kname)
            if pad_mode == Z:   # FIR with the flipped taps and g_pad = (ks-1-p0, ks-1-p1) (upfirdn2d.py:116-121)
                dx, _ = ops.blur_nhwc(d_xb, kf, ks - 1 - p0, ks - 1 - p1, Z)
            else:               # full correlation onto the padded extent, then the adjoint of the reflection padding
                dpad, _ = ops.blur_nhwc(d_xb, kf, ks - 1, ks - 1, Z)
                dx = ops.pad2d_bwd(dpad, p0, p1, p0, p1, pad_mode)
        return dx, dw, db, None, None, None, None, None, None, None, None


def blur_conv(x, w, net, wname, kname, bias=None, scale=1.0, p0=2, p1=2, pad_mode=Z, act=NONE, stats=False):
    y, st = BlurConvFn.apply(x, w, bias, net, wname, kname, scale, p0, p1, pad_mode, act)
    return (y, st) if stats else y


class BlurDownFn(Function):
    """Blur (zero padding (p0, p1)) keeping every second sample: the skip branch's Blur + stride-2 1x1 conv only reads
    those (stylegan2_layers.py:566-568)."""

    @staticmethod
    def forward(ctx, x, net, kname, p0, p1):
        x = _c(x)
        k = net.p(kname)
        y, _ = ops.blur_nhwc(x, k, p0, p1, Z, down=2)
        ctx.cfg = (net, kname, p0, p1, x.shape[1], x.shape[2])
        ctx.like = y
        return y

    @staticmethod
    def backward(ctx, g):
        net, kname, p0, p1, H, W = ctx.cfg
        k = net.p(kname)
        ks = k.shape[0]
        g = _gl(g, ctx.like)
        oh, ow = g.shape[1], g.shape[2]
        # UpFirDn2dBackward (upfirdn2d.py:24-60): zero-insert x2, FIR with the flipped taps, g_pad
        gp0 = ks - p0 - 1
        dx = ops.upfirdn2d_raw(g, _flipped(net, kname), 2, 2, 1, 1, gp0, W - 2 * ow + p0, gp0, H - 2 * oh + p0)
        return dx, None, None, None, None


class InstanceNormFn(Function):
    """out = act(IN(y) * (s0 + 1) + s1 + post_bias): nn.InstanceNorm2d (eps 1e-5, biased variance) [+ StyleMod
    (stylegan2_layers.py:361-374), style = its linear's (B, 2C) output] [+ the FusedLeakyReLU bias / activation that
    follows the norm in ConvLayer(norm='in'), :542-549] [or PReLU (generator.py:10-32)].
    ``st``: tile statistics of y from the producing kernel (None: computed here)."""

    @staticmethod
    def forward(ctx, y, st, style, post_bias, prelu, act, eps, res=None, out_scale=1.0, res_up2=False, post_gate=False):
        """``post_gate``: y is the OUTPUT of a leaky ReLU (x sqrt2) whose producer (ConvFn with gate_downstream) wants the
        gradient at its pre-activation: the apply pass of the backward multiplies by lrelu'(y) on its way out (no separate gate pass).
        ``res`` / ``out_scale`` (act NONE only): (IN(y) + res) * out_scale in the same pass -- the resnet merge
        (skip + res) / sqrt2 of generator.py:47-78 without materialising the normalised branch (bit-identical to
        instance_norm followed by AddScaleFn: the same fp32 operations in the same order).  ``res_up2``: ``res`` is the
        half-resolution skip tensor, sampled bilinearly (x2, align_corners=False) by the pass itself -- the upsampled skip of
        UpsamplingResnetBlock (generator.py:63-78) is never written."""
        y = _c(y)
        B, H, W, C = y.shape
        if st is None:
            st = ops.in_stats(y)
        ss, mr = ops.in_finalize_train(st, H * W, style=style, post_bias=post_bias, eps=eps)
        if res is not None:
            assert act == NONE
            out = ops.affine_act(y, ss, res=_c(res), out_scale=out_scale, res_up2=res_up2)
        else:
            assert out_scale == 1.0
            out = ops.affine_act(y, ss, act=act, prelu=prelu)
        ctx.save_for_backward(y, out if act == LRELU else None, mr, style, ss if act == PRELU else None, prelu)
        ctx.act = act
        ctx.post_gate = bool(post_gate)
        ctx.refs = (post_bias,)
        ctx.out_scale = float(out_scale) if res is not None else None
        ctx.res_low = (res.shape[1], res.shape[2]) if (res is not None and res_up2) else None
        return out

    @staticmethod
    def backward(ctx, g):
        y, out, mr, style, ss, prelu = ctx.saved_tensors
        g = _gl(g, y)
        B, H, W, C = y.shape
        dprelu = None
        gate = gates.sign_gate(out, "in-lrelu") if ctx.act == LRELU else None
        if ctx.act == PRELU:
            g, dprelu = prelu_bwd(g, y, prelu, scale_shift=ss)
        dres = None
        if ctx.out_scale is not None:
            # the merge's backward: g * out_scale is the gradient of the residual input AND the upstream of the norm
            g = dres = ops.affine_act(g, None, out_scale=ctx.out_scale)
            if ctx.res_low is not None:
                dres = ops.bilinear_bwd(g, *ctx.res_low) if ctx.needs_input_grad[7] else None
        part = ops.dual_stats(g, y, gate)
        want = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        coef, dstyle = ops.in_bwd_finalize(part, H * W, mr, style, want_dstyle=want)
        dy = None
        if ctx.needs_input_grad[0]:
            if ctx.post_gate and gates.MODE["value"] is not None:
                # gate tape recording / replaying (decided at BACKWARD time: the tape is switched on around backward()): the
                # producer's leaky-ReLU gate goes through the tape exactly where ConvFn's own gate pass would have put it
                dy = lrelu_bwd(ops.in_bwd_apply(g, y, coef, gate=gate), y)
            else:
                dy = ops.in_bwd_apply(g, y, coef, gate=gate, post_gate=ctx.post_gate)
        dpb = None
        if ctx.needs_input_grad[3]:
            (post_bias,) = ctx.refs
            dst = _direct(post_bias)
            dpb = ops.colsum(dstyle[:, C:], out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(post_bias); dpb = None
        return dy, None, (dstyle if ctx.needs_input_grad[2] else None), dpb, dprelu, None, None, dres, None, None, None


def instance_norm(y, st=None, style=None, post_bias=None, prelu=None, act=NONE, eps=1e-5, res=None, out_scale=1.0, res_up2=False,
                  post_gate=False):
    return InstanceNormFn.apply(y, st, style, post_bias, prelu, act, eps, res, out_scale, res_up2, post_gate)


class AddScaleFn(Function):
    """(a + b) * s  -- the resnet merge (skip + res) / sqrt2."""

    @staticmethod
    def forward(ctx, a, b, s):
        ctx.s = s
        ctx.like = a
        return ops.affine_act(_c(a), None, res=_c(b), out_scale=s)

    @staticmethod
    def backward(ctx, g):
        gs = ops.affine_act(_gl(g, ctx.like), None, out_scale=ctx.s)
        return gs, gs, None


class PReluResFn(Function):
    """prelu(a + res) with the shared slope of generator.py:ResidualBlock (:28-31)."""

    @staticmethod
    def forward(ctx, a, res, prelu):
        a, res = _c(a), _c(res)
        ctx.save_for_backward(a, res, prelu)
        return ops.affine_act(a, None, res=res, res_before_act=True, act=PRELU, prelu=prelu)

    @staticmethod
    def backward(ctx, g):
        a, res, prelu = ctx.saved_tensors
        gpre, ds = prelu_bwd(_c(g), a, prelu, res=res)
        return gpre, gpre, ds


class BilinearFn(Function):
    """F.interpolate(mode='bilinear', align_corners=False) to (OH, OW)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        x = _c(x)
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.bilinear(x, OH, OW)

    @staticmethod
    def backward(ctx, g):
        return ops.bilinear_bwd(_c(g), *ctx.hw), None, None


class AvgPoolFn(Function):
    """adaptive_avg_pool2d by an integer factor."""

    @staticmethod
    def forward(ctx, x, f):
        ctx.f = f
        return ops.avgpool(_c(x), f)

    @staticmethod
    def backward(ctx, g):
        return ops.avgpool_bwd(_c(g), ctx.f), None


class PadFn(Function):
    @staticmethod
    def forward(ctx, x, p, mode):
        ctx.cfg = (p, mode)
        return ops.pad2d(_c(x), p, p, p, p, mode)

    @staticmethod
    def backward(ctx, g):
        p, mode = ctx.cfg
        return ops.pad2d_bwd(_c(g), p, p, p, p, mode), None, None


class GapGmpFn(Function):
    """cat(AdaptiveAvgPool2d(1), AdaptiveMaxPool2d(1)) of x * mask (encoder_col.py:162-168, 217-245) -> (B, 2C)."""

    @staticmethod
    def forward(ctx, x, mask):
        x = _c(x)
        v = ops.gap_gmp(x, mask)
        ctx.save_for_backward(x, mask, v)
        return v

    @staticmethod
    def backward(ctx, g):
        x, mask, v = ctx.saved_tensors
        x, v = gates.values("gmp-argmax", x, v)          # (they only pick the arg-max pixel; the tape can replay another run's)
        return ops.gap_gmp_bwd(x, mask, v, _c(g).float()), None


class GapGmpMultiFn(Function):
    """GapGmpFn for the unmasked pooling and the three class-masked poolings of ONE feature map at once (encoder_col.py:162-168,
    217-245): -> (4 * B, 2C) head-major [plain | mask 0 | mask 1 | mask 2].  One read of x forward; backward one pass that writes the
    sum of the four heads' gradients (round 5: four dense gradients and three autograd adds per map before)."""

    @staticmethod
    def forward(ctx, x, masks):
        x = _c(x)
        v = ops.gap_gmp_multi(x, masks, True)
        ctx.save_for_backward(x, masks, v)
        return v

    @staticmethod
    def backward(ctx, g):
        x, masks, v = ctx.saved_tensors
        g = _c(g).float()
        if gates.MODE["value"] is not None:
            # gate tape (switched on around backward()): the arg-max of every head goes through the tape like GapGmpFn's, head by head
            B, dx = x.shape[0], None
            for h in range(4):
                xh, vh = gates.values("gmp-argmax", x, v[h * B:(h + 1) * B].contiguous())
                dx = ops.gap_gmp_bwd(xh, None if h == 0 else masks[..., h - 1].contiguous(), vh, g[h * B:(h + 1) * B].contiguous(), out=dx)
            return dx, None
        return ops.gap_gmp_multi_bwd(x, masks, v, g, True), None


class LinearFn(Function):
    """y = act(relu_in(x) @ (w*wscale)^T + b*bscale): EqualLinear / EqualizedLinear / nn.Linear behind nn.ReLU."""

    @staticmethod
    def forward(ctx, x, w, b, wscale, bscale, relu_in, act):
        x = _c(x)
        y = ops.linear(x, w.reshape(w.shape[0], -1), b, wscale=wscale, bscale=bscale, relu_in=relu_in, act=act)
        ctx.save_for_backward(x, w, y if act == LRELU else None)
        ctx.cfg = (wscale, bscale, relu_in, act, b is not None)
        ctx.refs = (w, b)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        wscale, bscale, relu_in, act, has_b = ctx.cfg
        g = _c(g)
        if act == LRELU:
            g = lrelu_bwd(g, y)
        w2 = w.reshape(w.shape[0], -1)
        dx = dw = db = None
        wp, bp = ctx.refs
        fused = FUSE_LINEAR["value"] and gates.MODE["value"] is None and w2.shape[1] % 4 == 0
        if ctx.needs_input_grad[0]:
            if relu_in and fused:          # the ReLU's backward rides on the slice reduction of the input gradient
                dx = ops.linear_dgrad_gate(g, w2, x, wscale)
            else:
                dx = ops.linear_dgrad(g, w2, wscale)
                if relu_in:
                    dx = relu_gate(dx, x)
        need_b = has_b and ctx.needs_input_grad[2]
        dst0 = _direct(wp) if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[1] and fused and (dst0 is None or dst0.data_ptr() % 16 == 0):
            # one launch: relu(x) read on the fly, dW added into the flat gradient, the bias gradient from the same rows of g
            dst, bdst = dst0, (_direct(bp) if need_b else None)
            dw, db = ops.linear_wgrad_fused(g, x, wscale, out=dst, accumulate=dst is not None, relu_in=relu_in, bias_out=bdst,
                                            bias_scale=bscale, bias_accumulate=bdst is not None, want_bias=need_b)
            if dst is not None:
                _noted(wp); dw = None
            else:
                dw = dw.view_as(w)
            if need_b and bdst is not None:
                _noted(bp); db = None
            return dx, dw, db, None, None, None, None
        if ctx.needs_input_grad[1]:
            xin = relu_gate(x, x) if relu_in else x      # relu(x) = x * [x > 0]
            dst = _direct(wp)
            dw = ops.linear_wgrad(g, xin, wscale, out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(wp); dw = None
            else:
                dw = dw.view_as(w)
        if need_b:
            dst = _direct(bp)
            db = ops.colsum(g, bscale, out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(bp); db = None
        return dx, dw, db, None, None, None, None


def linear(x, w, b=None, wscale=1.0, bscale=1.0, relu_in=False, act=NONE):
    return LinearFn.apply(x, w, b, wscale, bscale, relu_in, act)


class L2NormFn(Function):
    """mode 0: util.normalize (x * rsqrt(sum x^2 + 1e-8), util/util.py:18-22); mode 1: F.normalize (eps 1e-12)."""

    @staticmethod
    def forward(ctx, x, eps, mode):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.cfg = (eps, mode)
        return ops.l2norm_rows(x, eps, mode)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.l2norm_rows_bwd(_c(g), x, *ctx.cfg), None, None


class SpatialModFn(Function):
    """GeneratorModulation (generator.py:80-91): sp * scale[b, c] + shift[b, c]."""

    @staticmethod
    def forward(ctx, sp, scale, shift):
        sp, scale = _c(sp), _c(scale)
        ctx.save_for_backward(sp, scale)
        return ops.spatial_modulation(sp, scale, _c(shift), out_dtype=ops.train_dtype())

    @staticmethod
    def backward(ctx, g):
        sp, scale = ctx.saved_tensors
        g = _gl(g, sp)                      # (sp is an fp32 tensor: the 64 x 64 spatial code)
        B, H, W, C = sp.shape
        dsp = None
        if ctx.needs_input_grad[0]:
            ss = torch.stack((scale, torch.zeros_like(scale)), dim=2).contiguous()
            dsp = ops.affine_act(g, ss)
        _, d = ops.in_bwd_finalize(ops.dual_stats(g, sp), H * W)       # (sum g*sp, sum g) per (b, c)
        return dsp, d[:, :C], d[:, C:]


class FromRGBFn(Function):
    """ConvLayer(3, C, 1): 1x1 conv (no conv bias) + FusedLeakyReLU (stylegan2_layers.py:497-555)."""

    @staticmethod
    def forward(ctx, x, w, b, scale):
        x = _c(x)
        y = ops.conv1x1_small_cin(x, w, b, scale, LRELU, out_dtype=ops.train_dtype())
        ctx.save_for_backward(x, w, y)
        ctx.scale = scale
        ctx.refs = (w, b)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g0 = lrelu_bwd(_gl(g, y), y)
        C = g0.shape[3]
        wp, bp = ctx.refs
        db = dw = None
        if ctx.needs_input_grad[2]:
            dst = _direct(bp)
            db = ops.colsum(g0.view(-1, C), out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(bp); db = None
        if ctx.needs_input_grad[1]:
            dst = _direct(wp)
            dw = ops.wgrad_small_cin(x, g0, ctx.scale, out=dst, accumulate=dst is not None)
            if dst is not None:
                _noted(wp); dw = None
            else:
                dw = dw.view_as(w)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = w.reshape(w.shape[0], w.shape[1]).t().contiguous()
            dx = ops.conv1x1_small_cout(g0, wt, None, wscale=ctx.scale)
        return dx, dw, db, None


class ToRGBConvFn(Function):
    """EqualConv2d(C, 3, 1) + bias of ToRGB (stylegan2_layers.py:477-495)."""

    @staticmethod
    def forward(ctx, x, w, b, scale):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.scale = scale
        return ops.conv1x1_small_cout(x, w, b, scale)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _c(g).float()
        cout, cin = w.shape[0], w.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv1x1_small_cin(g, w.reshape(cout, cin).t().contiguous(), None, ctx.scale, NONE, out_dtype=x.dtype)
        if ctx.needs_input_grad[1]:
            dw = ops.wgrad_small_cin(g, x, ctx.scale).reshape(cin, cout).t().contiguous().view_as(w)   # (cin, cout) -> (cout, cin, 1, 1)
        if ctx.needs_input_grad[2]:
            db = ops.colsum(g.view(-1, cout))
        return dx, dw, db, None


class ToNHWCFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nchw_to_nhwc(x)

    @staticmethod
    def backward(ctx, g):
        return ops.nhwc_to_nchw(_c(g))


class ToNCHWFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nhwc_to_nchw(_c(x))

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(_c(g))


class L1LossFn(Function):
    """weight * torch.nn.L1Loss()(a, b) -> (1,); gradient flows to ``a`` only (b is a target)."""

    @staticmethod
    def forward(ctx, a, b, weight):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.w = weight
        return ops.l1_mean(a, b, weight)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        (a,) = gates.values("l1-sign", a)                # (sign(a - b) only)
        da = ops.l1_grad(a, b, ctx.w)
        return ops.scale_by(da, g), None, None


class LsganFn(Function):
    """weight * mean((pred - target)^2) (models/networks/loss.py:11-18)."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        loss, grad = ops.lsgan(_c(pred), target, weight)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return ops.scale_by(grad, g), None, None


class DiscriminatorLogitsFn(Function):
    """D(img) for the generator iteration: D's parameters are frozen (set_requires_grad(Dparams, False),
    ppst_optimizer.py:74), only d/d(img) is needed -- the taped forward / input-gradient backward of the
    discriminator trainer (ppst_amd/train.py)."""

    @staticmethod
    def forward(ctx, img, trainer):
        pred, tape = trainer.forward(img)
        ctx.tape, ctx.trainer = tape, trainer
        return pred

    @staticmethod
    def backward(ctx, g):
        keep = {}
        ctx.trainer.backward(ctx.tape, _c(g), param_grads=False, keep=keep)
        ctx.tape = None
        return ops.nhwc_to_nchw(keep["d_img"]), None


class DLossesFn(Function):
    """compute_image_discriminator_losses (ppst_model.py:68-103) as ONE autograd node over the discriminator trainer's taped
    forward / backward (ppst_amd/train.py): outputs D_real, D_rec[, D_mix]; backward writes d/d(theta_D) straight into the
    trainer's flat gradient (= the ``p.grad`` views of D's parameters) -- the images are constants here (rec / mix were made
    under no_grad), so no gradient leaves the node.  ``anchor`` is the trainer's requires-grad leaf that keeps the node alive."""

    @staticmethod
    def forward(ctx, anchor, trainer, real, rec, mix, lambda_GAN):
        losses, state = trainer.d_forward(real, rec, mix, lambda_GAN)
        ctx.trainer, ctx.state = trainer, state
        return tuple(losses.values())

    @staticmethod
    def backward(ctx, *gouts):
        ctx.trainer.d_backward(ctx.state, [None if g is None else _c(g) for g in gouts])
        ctx.state = None
        return None, None, None, None, None, None


class R1Fn(Function):
    """compute_R1_loss (ppst_model.py:140-159): per-sample penalty; backward = the second-order sweep of
    DiscriminatorTrainer.r1_backward with the upstream per-sample gradient."""

    @staticmethod
    def forward(ctx, anchor, trainer, real, lambda_R1):
        pen, state = trainer.r1_forward(real, lambda_R1)
        ctx.trainer, ctx.state = trainer, state
        return pen

    @staticmethod
    def backward(ctx, gout):
        ctx.trainer.r1_backward(ctx.state, _c(gout))
        ctx.state = None
        return None, None, None, None


# ---------------------------------------------------------------- correspondence (ppst_model.py:330-387) ----
class RSelfCorrFn(Function):
    """PPSTModel.Rselfcorr: (B,256,256,64) NHWC -> (B,64,64,256)."""

    @staticmethod
    def forward(ctx, fea1):
        fea1 = _c(fea1)
        ctx.save_for_backward(fea1)
        return ops.rselfcorr(fea1)

    @staticmethod
    def backward(ctx, g):
        (fea1,) = ctx.saved_tensors
        return ops.rselfcorr_bwd(fea1, _c(g))


class CorrMFn(Function):
    """PPSTModel.corrm: softmax(cos(q_i, k_j) / 0.01) over j; fea (keys) / fea0 (queries) NHWC (B,h,w,512).
    match_kernel k != 1: both maps as k x k neighbourhood rows first (ppst_model.py:345-347)."""

    @staticmethod
    def forward(ctx, fea, fea0, match_kernel=1):
        k, q = _c(fea), _c(fea0)
        B, h, w, C = k.shape
        if match_kernel == 1:
            k, q = k.reshape(B, h * w, C), q.reshape(B, h * w, C)
        else:
            k, q = ops.unfold_rows(k, match_kernel), ops.unfold_rows(q, match_kernel)
        kn, qn = ops.corr_prep(k, 256), ops.corr_prep(q, 256)
        corr = ops.softmax_rows_(ops.gemm_nt(qn, kn), 0.01)
        ctx.save_for_backward(k, q, kn, qn, corr)
        ctx.shape, ctx.mk = (B, h, w, C), match_kernel
        return corr

    @staticmethod
    def backward(ctx, g):
        k, q, kn, qn, corr = ctx.saved_tensors
        ds = ops.softmax_rows_bwd_(corr, _c(g).clone(), 0.01)          # d/d(cosine matrix)

        def rows_to_map(d):
            return d.view(ctx.shape) if ctx.mk == 1 else ops.unfold_rows_bwd(d, ctx.shape, ctx.mk)
        dk = dq = None
        if ctx.needs_input_grad[1]:
            dq = rows_to_map(ops.corr_prep_bwd(ops.gemm_nn(ds, kn, mode="x3"), q, 256))
        if ctx.needs_input_grad[0]:
            dk = rows_to_map(ops.corr_prep_bwd(ops.gemm_nn(ops.transpose_last2(ds), qn, mode="x3"), k, 256))
        return dk, dq, None


class WarpGemmFn(Function):
    """corr (B,P,P) @ V (B,P,C).  The correspondence matrix receives a gradient through the first ``nlive`` channels
    of V only (E2.warp uses corrmatrix.detach() for the deeper levels, encoder_col.py:197)."""

    @staticmethod
    def forward(ctx, corr, V, nlive):
        corr, V = _c(corr), _c(V)
        ctx.save_for_backward(corr, V)
        ctx.nlive = nlive
        return ops.gemm_nn(corr, V, mode="x3")

    @staticmethod
    def backward(ctx, g):
        corr, V = ctx.saved_tensors
        g = _c(g)
        dcorr = dV = None
        if ctx.needs_input_grad[0]:
            n = ctx.nlive
            dcorr = ops.gemm_nt(g[..., :n].contiguous(), V[..., :n].contiguous(), mode="x3")
        if ctx.needs_input_grad[1]:
            dV = ops.gemm_nn(ops.transpose_last2(corr), g, mode="x3")
        return dcorr, dV, None


class GemmConstBFn(Function):
    """corr @ M with a constant right operand (PPSTModel.warp of the one-hot mask)."""

    @staticmethod
    def forward(ctx, corr, M):
        ctx.save_for_backward(M)
        return ops.gemm_nn(_c(corr), M, mode="x3")

    @staticmethod
    def backward(ctx, g):
        (M,) = ctx.saved_tensors
        return ops.gemm_nt(_c(g), M, mode="x3"), None


class FoldFn(Function):
    """F.fold(x^T, (h, w), s, stride=s) of (B, P, C*s*s) patches -> NCHW (ppst_model.py:366-387)."""

    @staticmethod
    def forward(ctx, x, c, h, w, s):
        ctx.s = s
        return ops.fold_patches(_c(x), c, h, w, s)

    @staticmethod
    def backward(ctx, g):
        return ops.unfold_patches(_c(g), ctx.s), None, None, None, None


class UnfoldFn(Function):
    """F.unfold(x, s, stride=s)^T: NCHW -> (B, P, C*s*s) non-overlapping patches (PPSTModel.warp, ppst_model.py:366-387);
    its adjoint is the fold of the same patches."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.shape, ctx.s = tuple(x.shape), s
        return ops.unfold_patches(_c(x), s)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return ops.fold_patches(_c(g), C, H, W, ctx.s), None


class RsclLossFn(Function):
    """rsclLoss.forward (networks/rscl.py:42-64); gradient to the queries only (keys and queue are detached)."""

    @staticmethod
    def forward(ctx, q, k, k0, queue, T):
        q, k, k0, queue = _c(q), _c(k), _c(k0), _c(queue).clone()
        ctx.save_for_backward(q, k, k0, queue)
        ctx.T = T
        return ops.rscl_loss(q, k, k0, queue, T)

    @staticmethod
    def backward(ctx, g):
        q, k, k0, queue = ctx.saved_tensors
        return ops.rscl_loss_bwd(q, k, k0, queue, _c(g), ctx.T), None, None, None, None
