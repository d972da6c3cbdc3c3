"""Discriminator update of the GAN train step on HIP kernels.

Reference: PPSTOptimizer.train_discriminator_one_step (optimizers/ppst_optimizer.py:96-130)
-> PPSTModel.compute_discriminator_losses (models/ppst_model.py:105-138): the frozen E1/E2/G
produce ``rec`` (half batch) and ``mix`` under no_grad, D classifies real / rec / mix with the
LSGAN loss (loss.py:11-18), ``sum(v.mean())`` is back-propagated through D only and Adam
(lr*c, betas**c, c = R1_every/(1+R1_every)) updates D.  Data parallel: one process per GPU,
gradients averaged with one flat all-reduce (RCCL over xGMI; 29.0 M fp32 = 116 MB, SURVEY
section 2 "Collective call sites") before the optimiser step.

Everything numerical is a HIP kernel: the forward kernels of the inference path (unfused
where the backward needs the pre-merge activation), the conv input gradient = the same MFMA
conv kernel on a transposed / flipped pack, the conv weight gradient = fp32-MFMA reduction over
pixels (csrc/train.hip).  The lazy R1 penalty (double backward) is one more forward sweep with
the same kernels (r1_forward / r1_backward).  The generator / encoder update lives in ppst_amd/train_g.py.
"""
import math

import torch

from . import gates, ops, weights
from .networks.base_network import to_nhwc

SQRT2 = math.sqrt(2.0)
INV_SQRT2 = 1.0 / SQRT2
# round 5: elementwise passes of D's backward folded into the convs beside them (tests/train_ab.py flips them for same-process A/B)
TRAIN_FUSE = {"d_fanin": True, "d_skip_scale": True}


def _lrelu_bwd(g, out, scale=1.0):
    """grad wrt the pre-activation of y = lrelu(x)*sqrt2 given grad wrt y, gated by sign(y)
    (fused_bias_act_kernel.cu:43); ``scale`` folds an upstream constant factor in."""
    return ops.fused_bias_act_raw(g, None, gates.sign_gate(out, "lrelu"), 3, 1, 0.2, SQRT2 * scale)


class DiscriminatorTrainer:
    """Owns D's parameters (views into one flat buffer), their gradients and Adam state."""

    @classmethod
    def for_network(cls, D, **kw):
        """The trainer that owns ``D`` (created on first use and kept on the network).  The constructor rebinds
        every parameter of D into one flat buffer; a second trainer on the same D would leave the first with an
        orphaned flat / m / v whose Adam step updates memory D no longer reads."""
        tr = D.__dict__.get("_trainer")
        if tr is None or not tr.owns_parameters():
            tr = cls(D, **kw)
        return tr

    def owns_parameters(self):
        """True while D's parameters still alias this trainer's flat buffer (``.to()`` / ``.cuda()`` or another
        trainer's constructor break the aliasing)."""
        params = [p for _, p in self.D.named_parameters()]
        off = 0
        for p, n in zip(params, self.sizes):
            if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                return False
            off += n
        return True

    def __init__(self, D, lr=1e-3, beta1=0.0, beta2=0.99, R1_once_every=16, world=1):
        self.D = D
        D.__dict__["_trainer"] = self
        self.size = D.size
        c = R1_once_every / (1 + R1_once_every)
        self.lr, self.b1, self.b2, self.eps = lr * c, beta1 ** c, beta2 ** c, 1e-8
        self.step_count = 0
        self.iter_counter = 0            # discriminator_iter_counter (ppst_optimizer.py:103)
        self.R1_once_every = R1_once_every
        self.world = world
        # flatten parameters so the all-reduce and Adam are single launches
        self.names = [n for n, _ in D.named_parameters()]
        params = [p for _, p in D.named_parameters()]
        self.sizes = [p.numel() for p in params]
        flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        self.flat = flat
        off = 0
        for p, n in zip(params, self.sizes):
            p.data = flat[off:off + n].view_as(p)
            off += n
        D._flat.clear(); D._cache.clear()
        self.grad = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.offsets = {}
        off = 0
        for n_, p, sz in zip(self.names, params, self.sizes):
            self.offsets[n_] = (off, sz)
            p.grad = self.grad[off:off + sz].view_as(p)      # what optimizer_D reads in the reference: views of the flat gradient
            off += sz
        # the facade's loss tensors hang on this leaf (ppst_model.compute_discriminator_losses / compute_R1_loss): D's
        # parameters themselves are not autograd inputs, their gradients are written by the taped backward below
        self.anchor = torch.zeros(1, device=flat.device, requires_grad=True)
        self._pending = None                                  # async gradient all-reduce whose Adam step is still owed

    def g(self, name):
        off, sz = self.offsets[name]
        return self.grad[off:off + sz]

    # ---------------------------------------------------------------- forward
    def forward(self, img):
        """D forward keeping what the backward needs.  img NCHW.  Returns (pred (B,1), tape)."""
        self.finish_pending()
        D, p = self.D, "stylegan2_D."
        tape = {}
        x0 = to_nhwc(img)
        x0 = x0 if x0.is_contiguous() else x0.contiguous()
        tape["img"] = x0
        x = D.from_rgb(x0, p + "convs.0.", out_dtype=ops.train_dtype())      # (bfloat16 storage of the tape in precision mode 1)
        tape["x0"] = x
        blocks = []
        for name in weights.discriminator_block_names(self.size):
            q = p + "convs.%s." % name
            B, S, _, cin = x.shape
            k = D.p(q + "conv2.Blur.kernel")
            sc1, scs = 1.0 / math.sqrt(cin * 9), 1.0 / math.sqrt(cin)
            y1 = D.plan(q + "conv1.Conv.weight", scale=sc1)(x, bias=D.p(q + "conv1.Act.bias"), act=ops.ACT_LRELU)
            xs, _ = ops.blur_nhwc(x, D.p(q + "skip.Blur.kernel"), 1, 1, ops.PAD_ZERO, down=2)
            skip = D.plan(q + "skip.Conv.weight", scale=scs)(xs)
            xb, bhw = ops.blur_nhwc(y1, k, 2, 2, ops.PAD_ZERO, s2d=True)
            ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
            a2 = D.plan(q + "conv2.Conv.weight", "s2d", sc1)(xb, bias=D.p(q + "conv2.Act.bias"), act=ops.ACT_LRELU, out_hw=ohw)
            out = ops.affine_act(a2, None, res=skip, out_scale=INV_SQRT2)
            blocks.append(dict(q=q, x=x, y1=y1, xs=xs, xb=xb, bhw=bhw, a2=a2, cin=cin, S=S))
            x = out
        tape["blocks"] = blocks
        cin = x.shape[3]
        fc = D.plan(p + "final_conv.Conv.weight", scale=1.0 / math.sqrt(cin * 9))(x, bias=D.p(p + "final_conv.Act.bias"), act=ops.ACT_LRELU)
        tape["x_last"], tape["fc"] = x, fc
        f = ops.nhwc_to_nchw(fc.float()).reshape(fc.shape[0], -1)           # (the 4 x 4 head and the two linears are fp32)
        w0, w1 = D.p(p + "final_linear.0.weight"), D.p(p + "final_linear.1.weight")
        h = ops.linear(f, w0, D.p(p + "final_linear.0.bias"), wscale=1.0 / math.sqrt(w0.shape[1]), act=ops.ACT_LRELU)
        pred = ops.linear(h, w1, D.p(p + "final_linear.1.bias"), wscale=1.0 / math.sqrt(w1.shape[1]))
        tape["f"], tape["h"] = f, h
        return pred, tape

    # --------------------------------------------------------------- backward
    def _gdst(self, name):
        """flat-gradient view of ``stylegan2_D.<name>``: the destination the gradient kernels add into (accumulate flag of the
        C ABI) -- no temporary and no separate accumulation launch per parameter."""
        return self.g("stylegan2_D." + name)

    def backward(self, tape, dpred, param_grads=True, keep=None):
        """Back-propagate ``dpred`` (B,1) through the taped forward.  ``param_grads`` accumulates
        d/d(theta_D) into self.grad; ``keep`` (a dict) receives the gradient wrt every conv / linear
        pre-activation and wrt the image (what the R1 second-order pass contracts against)."""
        D, p = self.D, "stylegan2_D."
        G = self._gdst                     # parameter gradients are ADDED into the flat buffer by the kernels themselves
        pg = param_grads
        w0, w1 = D.p(p + "final_linear.0.weight"), D.p(p + "final_linear.1.weight")
        s0, s1 = 1.0 / math.sqrt(w0.shape[1]), 1.0 / math.sqrt(w1.shape[1])
        f, h = tape["f"], tape["h"]
        # final_linear.1 (EqualLinear 512 -> 1)
        if pg:
            ops.linear_wgrad(dpred, h, s1, out=G("final_linear.1.weight"), accumulate=True)
            ops.colsum(dpred, out=G("final_linear.1.bias"), accumulate=True)
        dh = ops.linear_dgrad(dpred, w1, s1)
        # final_linear.0 (EqualLinear 8192 -> 512, fused lrelu)
        gpre_h = _lrelu_bwd(dh, h)
        if pg:
            ops.linear_wgrad(gpre_h, f, s0, out=G("final_linear.0.weight"), accumulate=True)
            ops.colsum(gpre_h, out=G("final_linear.0.bias"), accumulate=True)
        df = ops.linear_dgrad(gpre_h, w0, s0)
        fc = tape["fc"]
        B, hh, ww, C = fc.shape
        dfc = ops.nchw_to_nhwc(df.view(B, C, hh, ww)).to(fc.dtype)
        # final_conv (3x3 + fused lrelu)
        gpre = _lrelu_bwd(dfc, fc)
        scf = 1.0 / math.sqrt(tape["x_last"].shape[3] * 9)
        if pg:
            # (bias gradients = column sums of the same gradient tensor: out of the weight-gradient kernel's staging pass)
            ops.conv_wgrad(D.plan(p + "final_conv.Conv.weight", scale=scf), tape["x_last"], gpre, out=G("final_conv.Conv.weight"), accumulate=True,
                           bias_out=G("final_conv.Act.bias"), bias_accumulate=True)
        dx = self._dgrad(p + "final_conv.Conv.weight", "dgrad", scf)(gpre)
        if keep is not None:
            keep["dpred"], keep["gpre_h"], keep["gpre_fc"], keep["blocks"] = dpred, gpre_h, gpre, []
        for blk in reversed(tape["blocks"]):
            q, cin = blk["q"], blk["cin"]
            name = q[len(p):]
            sc1, scs = 1.0 / math.sqrt(cin * 9), 1.0 / math.sqrt(cin)
            cout = blk["a2"].shape[3]
            # out = (a2 + skip)/sqrt2 ;  a2 = lrelu(conv2 + b)*sqrt2
            g2 = _lrelu_bwd(dx, blk["a2"], INV_SQRT2)
            if pg:
                ops.conv_wgrad(D.plan(q + "conv2.Conv.weight", "s2d", sc1), blk["xb"], g2, out=G(name + "conv2.Conv.weight"), accumulate=True,
                               bias_out=G(name + "conv2.Act.bias"), bias_accumulate=True)
            d_xb = ops.dgrad_s2d(self.D, q + "conv2.Conv.weight", sc1, g2, blk["bhw"])
            # blur backward: upfirdn2d with the flipped (symmetric) taps and g_pad = (1, 1)
            kf = D.cached(("flip", q + "conv2.Blur.kernel"), [D.p(q + "conv2.Blur.kernel")],
                          lambda: torch.flip(D.p(q + "conv2.Blur.kernel"), [0, 1]).contiguous())
            d_y1, _ = ops.blur_nhwc(d_xb, kf, 1, 1, ops.PAD_ZERO)
            g1 = _lrelu_bwd(d_y1, blk["y1"])
            if pg:
                ops.conv_wgrad(D.plan(q + "conv1.Conv.weight", scale=sc1), blk["x"], g1, out=G(name + "conv1.Conv.weight"), accumulate=True,
                               bias_out=G(name + "conv1.Act.bias"), bias_accumulate=True)
            # skip branch: 1x1 conv on the blurred + decimated input, no bias / activation.  Its upstream gradient is dx / sqrt2: the
            # factor rides on the two kernels that consume it (round 5: no pass that materialises gs) -- unless the R1 sweep wants
            # gs itself (``keep``)
            if keep is not None or not TRAIN_FUSE["d_skip_scale"]:
                gs = ops.affine_act(dx, None, out_scale=INV_SQRT2)
                if pg:
                    ops.conv_wgrad(D.plan(q + "skip.Conv.weight", scale=scs), blk["xs"], gs, out=G(name + "skip.Conv.weight"), accumulate=True)
                d_xs = self._dgrad(q + "skip.Conv.weight", "dgrad", scs)(gs)
            else:
                gs = None
                if pg:
                    ops.conv_wgrad(D.plan(q + "skip.Conv.weight", scale=scs), blk["xs"], dx, out=G(name + "skip.Conv.weight"), accumulate=True,
                                   dy_scale=INV_SQRT2)
                d_xs = self._dgrad(q + "skip.Conv.weight", "dgrad", scs)(dx, out_scale=INV_SQRT2)
            # blur(down=2, pad (1,1)) backward = zero-insert x2 then FIR with g_pad (upfirdn2d.py:116-121)
            S = blk["S"]
            ks = D.p(q + "skip.Blur.kernel")
            ksz = ks.shape[0]
            oh = d_xs.shape[1]
            gp0 = ksz - 1 - 1
            gp1 = S - oh * 2 + 1 - 1 + 1
            ksf = D.cached(("flip", q + "skip.Blur.kernel"), [ks], lambda: torch.flip(D.p(q + "skip.Blur.kernel"), [0, 1]).contiguous())
            d_xb2 = ops.upfirdn2d_raw(d_xs, ksf, 2, 2, 1, 1, gp0, gp1, gp0, gp1)
            # dx = d_xa + d_xb2: the fan-in add rides on the epilogue of conv1's input-gradient conv (its ``residual``)
            if TRAIN_FUSE["d_fanin"]:
                dx = self._dgrad(q + "conv1.Conv.weight", "dgrad", sc1)(g1, residual=d_xb2)
            else:
                d_xa = self._dgrad(q + "conv1.Conv.weight", "dgrad", sc1)(g1)
                dx = ops.affine_act(d_xa, None, res=d_xb2)
            if keep is not None:
                keep["blocks"].append(dict(g1=g1, g2=g2, gs=gs))
        # FromRGB: 1x1 conv (no bias) + fused lrelu
        x0 = tape["x0"]
        g0 = _lrelu_bwd(dx, x0)
        w = D.p(p + "convs.0.Conv.weight")
        sc0 = 1.0 / math.sqrt(w.shape[1])
        if pg:
            ops.colsum(g0.view(-1, g0.shape[3]), out=G("convs.0.Act.bias"), accumulate=True)
            ops.wgrad_small_cin(tape["img"], g0, sc0, out=G("convs.0.Conv.weight"), accumulate=True)
        if keep is not None:
            keep["blocks"].reverse()
            keep["g0"] = g0
            # d/d(image) = W0^T g0  (3 output channels)
            wt = w.reshape(w.shape[0], w.shape[1]).t().contiguous()
            keep["d_img"] = ops.conv1x1_small_cout(g0, wt, None, wscale=sc0)

    # -------------------------------------------------------------- R1 penalty
    def r1_penalty(self, real, lambda_R1=10.0):
        """compute_R1_loss value only: per-sample 0.5*lambda*||d sum(D(x))/dx||^2 (forward + backward to the image)."""
        return self.r1_forward(real, lambda_R1)[0]

    def r1_forward(self, real, lambda_R1=10.0):
        """compute_R1_loss (ppst_model.py:140-159): per-sample penalty 0.5*lambda*||d sum(D(x)) / dx||^2 -> (pen (B,), state)."""
        pred, tape = self.forward(real)
        B = pred.shape[0]
        keep = {}
        self.backward(tape, torch.ones_like(pred), param_grads=False, keep=keep)
        g_img = keep["d_img"]                                    # (B,S,S,3)
        part = ops.in_stats(g_img)                                # (B, n, 3, 2): per-channel (sum, sumsq) partials
        pen = torch.stack([ops.colsum(part[b].view(-1, 2), 0.5 * lambda_R1)[1] for b in range(B)])
        return pen, (tape, keep, lambda_R1)

    def r1_backward(self, state, gout):
        """d(sum_b gout[b] * pen[b]) / d(theta_D) accumulated into self.grad (ppst_optimizer.py:116-126 feeds
        gout = R1_once_every / B: ``r1_loss = mean(pen) * R1_once_every``).

        Leaky-ReLU gates are piecewise constant, so g(x) = J(theta)^T 1 is linear in each weight
        with the taped gates fixed and biases drop out.  Reverse mode over the *backward* pass
        turns into one more forward sweep: t = dL/dg enters at the image, runs through the same
        convs / blurs / gates (no biases), and every layer adds wgrad(input = t_in, dy = the
        first backward's gradient at that layer's pre-activation)."""
        D, p = self.D, "stylegan2_D."
        G = self._gdst
        tape, keep, lambda_R1 = state
        g_img = keep["d_img"]
        B = g_img.shape[0]
        # dL/dg_b = gout[b] * lambda * g_b: the per-image factor rides in as the (a, s) table of the apply kernel
        ss = torch.stack((gout.reshape(B, 1).expand(B, 3), torch.zeros((B, 3), device=g_img.device)), dim=2).contiguous()
        t = ops.affine_act(g_img, ss, out_scale=lambda_R1)
        w = D.p(p + "convs.0.Conv.weight")
        sc0 = 1.0 / math.sqrt(w.shape[1])
        ops.wgrad_small_cin(t, keep["g0"], sc0, out=G("convs.0.Conv.weight"), accumulate=True)
        t = ops.conv1x1_small_cin(t, w, None, sc0, ops.ACT_NONE, out_dtype=tape["x0"].dtype)
        t = _lrelu_bwd(t, tape["x0"])
        for blk, kb in zip(tape["blocks"], keep["blocks"]):
            q, cin = blk["q"], blk["cin"]
            name = q[len(p):]
            sc1, scs = 1.0 / math.sqrt(cin * 9), 1.0 / math.sqrt(cin)
            ops.conv_wgrad(D.plan(q + "conv1.Conv.weight", scale=sc1), t, kb["g1"], out=G(name + "conv1.Conv.weight"), accumulate=True)
            t1 = D.plan(q + "conv1.Conv.weight", scale=sc1)(t)
            t1 = _lrelu_bwd(t1, blk["y1"])
            ts, _ = ops.blur_nhwc(t, D.p(q + "skip.Blur.kernel"), 1, 1, ops.PAD_ZERO, down=2)
            ops.conv_wgrad(D.plan(q + "skip.Conv.weight", scale=scs), ts, kb["gs"], out=G(name + "skip.Conv.weight"), accumulate=True)
            tskip = D.plan(q + "skip.Conv.weight", scale=scs)(ts)
            tb, bhw = ops.blur_nhwc(t1, D.p(q + "conv2.Blur.kernel"), 2, 2, ops.PAD_ZERO, s2d=True)
            ops.conv_wgrad(D.plan(q + "conv2.Conv.weight", "s2d", sc1), tb, kb["g2"], out=G(name + "conv2.Conv.weight"), accumulate=True)
            ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
            t2 = D.plan(q + "conv2.Conv.weight", "s2d", sc1)(tb, out_hw=ohw)
            t2 = _lrelu_bwd(t2, blk["a2"])
            t = ops.affine_act(t2, None, res=tskip, out_scale=INV_SQRT2)
        scf = 1.0 / math.sqrt(t.shape[3] * 9)
        ops.conv_wgrad(D.plan(p + "final_conv.Conv.weight", scale=scf), t, keep["gpre_fc"], out=G("final_conv.Conv.weight"), accumulate=True)
        t = D.plan(p + "final_conv.Conv.weight", scale=scf)(t)
        t = _lrelu_bwd(t, tape["fc"])
        tf = ops.nhwc_to_nchw(t.float()).reshape(B, -1)
        w0, w1 = D.p(p + "final_linear.0.weight"), D.p(p + "final_linear.1.weight")
        s0, s1 = 1.0 / math.sqrt(w0.shape[1]), 1.0 / math.sqrt(w1.shape[1])
        ops.linear_wgrad(keep["gpre_h"], tf, s0, out=G("final_linear.0.weight"), accumulate=True)
        th = ops.linear(tf, w0, None, wscale=s0)
        th = _lrelu_bwd(th, tape["h"])
        ops.linear_wgrad(keep["dpred"], th, s1, out=G("final_linear.1.weight"), accumulate=True)

    def r1_losses_and_grads(self, real, lambda_R1=10.0, R1_once_every=16):
        """Lazy R1 (ppst_model.py:140-159, ppst_optimizer.py:116-126): zero_grad, penalty, and
        d(mean(penalty) * R1_once_every)/d(theta_D) into self.grad."""
        self.zero_grad()          # (settles an owed deferred step first: its all-reduce may still be reading self.grad)
        pen, state = self.r1_forward(real, lambda_R1)
        B = pen.shape[0]
        self.r1_backward(state, torch.full((B,), R1_once_every / B, device=pen.device))
        return {"D_R1": pen}

    def _dgrad(self, wname, kind, scale):
        return self.D.plan(wname, kind, scale)

    # ------------------------------------------------------------------- step
    def d_forward(self, real, rec, mix, lambda_GAN=1.0):
        """LSGAN losses on real / rec / mix (ppst_model.py:68-103) -> (losses dict, state).  The three image sets go through D
        as ONE batch (the gradient is the sum over the sets either way): one forward, one backward, a third of the launches,
        and the weight-gradient reduction runs over all images at once."""
        sets = [(n, img, t, w) for n, img, t, w in (("D_real", real, 1.0, lambda_GAN), ("D_rec", rec, 0.0, 0.5 * lambda_GAN),
                                                    ("D_mix", mix, 0.0, 0.5 * lambda_GAN)) if img is not None]
        imgs = torch.cat([s_[1] for s_ in sets], dim=0) if len(sets) > 1 else sets[0][1]
        pred, tape = self.forward(imgs)
        losses, dparts, o = {}, [], 0
        for name, img, target, wgt in sets:
            nb = img.shape[0]
            loss, dp = ops.lsgan(pred[o:o + nb].contiguous(), target, wgt)   # mean over this set's own batch
            losses[name] = loss
            dparts.append(dp)
            o += nb
        return losses, (tape, dparts)

    def d_backward(self, state, gouts=None):
        """d(sum_i gouts[i] * loss_i) / d(theta_D) accumulated into self.grad (gouts None: all ones)."""
        tape, dparts = state
        if gouts is not None:
            dparts = [dp if g is None else ops.scale_by(dp, g.reshape(1)) for dp, g in zip(dparts, gouts)]
        self.backward(tape, torch.cat(dparts, dim=0) if len(dparts) > 1 else dparts[0])

    def losses_and_grads(self, real, rec, mix, lambda_GAN=1.0):
        """zero_grad + LSGAN losses + d(sum of losses)/d(theta_D) into self.grad."""
        self.zero_grad()          # (settles an owed deferred step first: its all-reduce may still be reading self.grad)
        losses, state = self.d_forward(real, rec, mix, lambda_GAN)
        self.d_backward(state)
        return losses

    def zero_grad(self):
        self.finish_pending()
        self.grad.zero_()

    def all_reduce(self):
        """DDP gradient averaging: one flat all-reduce (RCCL when the tensors are on the GPU)."""
        self.finish_pending()
        ddp_average_(self.grad, self.world)

    def adam(self):
        self.finish_pending()     # (an owed step is applied first, in order; no-op when called from finish_pending itself)
        if not self.owns_parameters():
            raise RuntimeError("D's parameters no longer alias this trainer's flat buffer (a second trainer was built "
                               "on the same network, or .to()/.cuda() moved it): the update would be lost")
        self.step_count += 1
        ops.adam_step_(self.flat, self.grad, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self.step_count)
        self.D.refresh_plans()  # packed weights are stale: re-packed in place, two launches

    # Data parallel, overlapped: the discriminator iteration ends with its 116-MB gradient all-reduce; nothing else in that
    # iteration needs D any more, and the generator iteration that follows does not touch D before its GAN terms.  So the
    # all-reduce is launched asynchronously and the Adam step that consumes it is OWED: finish_pending() settles it (wait,
    # divide by the world size, Adam) in front of the next use of D -- forward(), zero_grad(), a checkpoint -- and the collective
    # runs under the E1 / E2 / G forward passes of the generator iteration.
    def step_deferred(self):
        """all_reduce + adam of the reference's ``optimizer_D.step()`` under DDP; the collective is asynchronous for world > 1."""
        if self.world <= 1:
            return self.adam()
        import torch.distributed as dist
        self._pending = dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, async_op=True)

    def finish_pending(self):
        h = self._pending
        if h is not None:
            self._pending = None
            h.wait()
            self.grad.div_(self.world)
            self.adam()

    def train_step(self, model, real, lambda_StyleCon=1.0, lambda_R1=10.0):
        """One discriminator iteration (ppst_optimizer.py:96-130): images from the frozen E1/E2/G,
        LSGAN losses, backward, all-reduce, Adam; every R1_once_every-th iteration a second
        zero_grad / R1 backward / all-reduce / Adam on the same real images."""
        self.iter_counter += 1
        rec, mix = d_step_images(model, real, lambda_StyleCon)
        losses = self.losses_and_grads(real, rec, mix)
        self.all_reduce()
        self.adam()
        if lambda_R1 > 0.0 and self.iter_counter % self.R1_once_every == 0:
            losses.update(self.r1_losses_and_grads(real, lambda_R1, self.R1_once_every))
            self.all_reduce()
            self.adam()
        return losses


def ddp_average_(flat_grad, world):
    """What DistributedDataParallel does for the reference (models/__init__.py:88): sum the
    gradients of all ranks and divide by the world size -- here as ONE flat collective instead of
    25-MB buckets (the whole D gradient is 116 MB; xGMI ring all-reduce ~ 2*(N-1)/N * S / 153 GB/s)."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.div_(world)
    return flat_grad


def d_step_images(model, real, lambda_StyleCon=1.0, want_codes=False):
    """rec (B/2) and mix (B) of compute_discriminator_losses (ppst_model.py:106-131), inference path."""
    from . import glue
    B = real.shape[0]
    assert B % 2 == 0, "Batch size must be even on each GPU."
    sp = model.E1(real)
    gl, _ = model.E2(real)
    nzh = {k: v[:B // 2] for k, v in model.noise.items()} if isinstance(model.noise, dict) else model.noise
    if int(getattr(model.opt, "training_stage", 2)) == 1:      # ppst_model.py:109-112, 128-131
        rec = model.G(sp[:B // 2], [g[:B // 2] for g in gl], noise=nzh)
        return (rec, None, sp, gl) if want_codes else (rec, None)
    _, feas, feas1 = model.G(sp, gl, extract_features=True, noise=model.noise, want_rgb=not getattr(model, 'skip_unused_rgb', False))
    sps = torch.cat((feas, model.Rselfcorr(feas1)), dim=1)
    corrms = model.corrm(sps, glue.swap(sps))
    corr_self = model.corrm(sps, sps)
    mix = None
    if lambda_StyleCon > 0.0:
        _, gl_w = model.E2(real, corrmatrix=corrms)
        mix = model.G(glue.swap(sp), gl_w, noise=model.noise)
    _, gl2 = model.E2(real, corrmatrix=corr_self)
    nz = model.noise
    if isinstance(nz, dict):
        nz = {k: v[:B // 2] for k, v in nz.items()}
    rec = model.G(sp[:B // 2], [g[:B // 2] for g in gl2], noise=nz)
    if want_codes:
        return rec, mix, sp, gl2
    return rec, mix
