"""GPU diagnostic: run every HIP kernel against the CPU oracle and print error tables.
Usage (on the GPU box):  python tests/gpu_diag.py [ops|nets|all]  > gpurun_out/diag.log
Not a pytest file; the pytest -m gpu tests assert the same comparisons with tolerances."""
import math
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ppst_oracle as O  # noqa: E402
from ppst_amd import ops, weights as W  # noqa: E402
from ppst_amd.networks.base_network import to_nhwc  # noqa: E402

dev = "cuda"
RES = []
_ORACLE_SWAPS = {}


def oracle_swap512():
    """The 512x512 oracle swap of t_networks / t_precision (same weights, images, noise): computed once per process --
    CPU oracle time dominates the GPU suite."""
    if "s" not in _ORACLE_SWAPS:
        sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
        imgs = W.synthetic_images(5, 2)
        with torch.no_grad():
            _ORACLE_SWAPS["s"] = O.PPSTOracle(sd, noise=W.make_noise(3, 1)).simple_swap(imgs[0:1], imgs[1:2], alpha=1.0)
    return _ORACLE_SWAPS["s"]


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if a.shape != b.shape:
        return float("nan"), "SHAPE %s vs %s" % (tuple(a.shape), tuple(b.shape))
    d = (a - b).abs().max().item()
    return d / (b.abs().max().item() + 1e-30), "maxabs %.3e refmax %.3e" % (d, b.abs().max().item())


def report(name, a, b, tol):
    r, info = rel(a, b)
    ok = r <= tol
    RES.append((name, ok))
    print("%-46s %s rel %.3e (tol %.0e) %s" % (name, "ok  " if ok else "FAIL", r, tol, info), flush=True)


def run(fn):
    try:
        fn()
    except Exception:
        RES.append((fn.__name__, False))
        print("EXC in %s\n%s" % (fn.__name__, traceback.format_exc()), flush=True)


def g(t):
    return t.to(dev)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ ops ----
def t_upfirdn2d():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))
    from ppst_amd.stylegan2_op import upfirdn2d
    for i in range(int(gold["upfirdn2d.n"])):
        x = torch.from_numpy(gold["upfirdn2d.%d.x" % i])
        k = torch.from_numpy(gold["upfirdn2d.%d.k" % i])
        u, d, p0, p1 = [int(v) for v in gold["upfirdn2d.%d.cfg" % i]]
        y = upfirdn2d(g(x), g(k), up=u, down=d, pad=(p0, p1))
        report("upfirdn2d golden %d (k%d u%d d%d p%d,%d)" % (i, k.shape[0], u, d, p0, p1), y, torch.from_numpy(gold["upfirdn2d.%d.y" % i]), 2e-6)
    # bigger planes incl. tile edges, NCHW
    torch.manual_seed(0)
    for (ks, p0, p1, H, Wd) in [(3, 0, 0, 131, 131), (3, 1, 0, 128, 128), (4, 2, 2, 64, 64), (4, 1, 1, 100, 70)]:
        x = torch.randn(2, 5, H, Wd)
        k = torch.randn(ks, ks)
        report("upfirdn2d planes k%d p%d,%d %dx%d" % (ks, p0, p1, H, Wd), upfirdn2d(g(x), g(k), pad=(p0, p1)), O.upfirdn2d(x, k, pad=(p0, p1)), 2e-6)
    # NHWC raw op (minor = C)
    x = torch.randn(2, 37, 41, 8)
    k = torch.randn(4, 4)
    y = ops.upfirdn2d_raw(g(x), g(k), 1, 1, 1, 1, 2, 1, 2, 1)
    report("upfirdn2d NHWC minor=8 k4", nchw(y.cpu()), O.upfirdn2d(nchw(x), k, pad=(2, 1)), 2e-6)
    y = ops.upfirdn2d_raw(g(x), g(k), 1, 1, 2, 2, 1, 1, 1, 1)
    report("upfirdn2d NHWC minor=8 k4 down2", nchw(y.cpu()), O.upfirdn2d(nchw(x), k, down=2, pad=(1, 1)), 2e-6)
    # fused-path blur: reflect pad, s2d, down2
    x = torch.randn(2, 32, 40, 40)
    k3 = O.make_kernel([1, 2, 1])
    yb, hw = ops.blur_nhwc(g(nhwc(x)), g(k3), 2, 1, ops.PAD_REFLECT, s2d=True)
    ref = O.upfirdn2d(F.pad(x, (2, 1, 2, 1), mode="reflect"), k3)
    oh, ow = ref.shape[2:]
    s2d = torch.zeros(2, (oh + 1) // 2, (ow + 1) // 2, 4, 32)
    for py in range(2):
        for px in range(2):
            sub = ref[:, :, py::2, px::2]
            s2d[:, :sub.shape[2], :sub.shape[3], py * 2 + px] = sub.permute(0, 2, 3, 1)
    report("blur_nhwc reflect s2d (%d,%d)" % hw, yb.cpu().view(2, (oh + 1) // 2, (ow + 1) // 2, 4, 32), s2d, 2e-6)
    yd, hw = ops.blur_nhwc(g(nhwc(x)), g(k3), 1, 0, ops.PAD_ZERO, down=2)
    report("blur_nhwc zero down2 (%d,%d)" % hw, nchw(yd.cpu()), O.upfirdn2d(x, k3, down=2, pad=(1, 0)), 2e-6)
    k4 = O.make_kernel([1, 3, 3, 1])
    yd, hw = ops.blur_nhwc(g(nhwc(x)), g(k4), 1, 1, ops.PAD_ZERO, down=2)
    report("blur_nhwc k4 zero down2", nchw(yd.cpu()), O.upfirdn2d(x, k4, down=2, pad=(1, 1)), 2e-6)


def t_fused_act():
    from ppst_amd.stylegan2_op import fused_leaky_relu, FusedLeakyReLU
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))
    x = g(torch.from_numpy(gold["flrelu.x"])).requires_grad_()
    b = g(torch.from_numpy(gold["flrelu.b"])).requires_grad_()
    y = fused_leaky_relu(x, b)
    report("fused_leaky_relu fwd golden", y, torch.from_numpy(gold["flrelu.y"]), 1e-6)
    gy = g(torch.from_numpy(gold["flrelu.g"]))
    gx, gb = torch.autograd.grad(y, [x, b], gy, create_graph=True)
    report("fused_leaky_relu grad_x golden", gx, torch.from_numpy(gold["flrelu.gx"]), 1e-6)
    report("fused_leaky_relu grad_b golden", gb, torch.from_numpy(gold["flrelu.gb"]), 1e-5)
    # double backward: d/d(gy) of sum(gx * v) == gate(v)
    v = torch.randn_like(gx)
    gy2 = gy.clone().requires_grad_()
    gx2, gb2 = torch.autograd.grad(fused_leaky_relu(x, b), [x, b], gy2, create_graph=True)
    gg, = torch.autograd.grad((gx2 * v).sum(), gy2)
    ref_gate = torch.where(y.detach() > 0, v, v * 0.2) * math.sqrt(2)
    report("fused_leaky_relu double backward", gg, ref_gate, 1e-6)
    x2, b2 = torch.from_numpy(gold["flrelu2.x"]), torch.from_numpy(gold["flrelu2.b"])
    report("fused_leaky_relu 2-D slope .1 scale 1.5", fused_leaky_relu(g(x2), g(b2), 0.1, 1.5), torch.from_numpy(gold["flrelu2.y"]), 1e-6)
    m = FusedLeakyReLU(5).to(dev)
    report("FusedLeakyReLU module", m(x.detach()), O.fused_leaky_relu(x.detach().cpu(), torch.zeros(5)), 1e-6)
    # upfirdn2d autograd (first + second order) against the oracle's autograd
    from ppst_amd.stylegan2_op import upfirdn2d
    xc = torch.randn(2, 3, 20, 22, dtype=torch.float64, requires_grad=True)
    k = torch.randn(4, 4, dtype=torch.float64)
    yc = O.upfirdn2d(xc, k, pad=(2, 1))
    gyc = torch.randn_like(yc, requires_grad=True)
    gxc, = torch.autograd.grad(yc, xc, gyc, create_graph=True)
    vc = torch.randn_like(gxc)
    ggc, = torch.autograd.grad((gxc * vc).sum(), gyc)
    xg = g(xc.detach().float()).requires_grad_()
    yg = upfirdn2d(xg, g(k.float()), pad=(2, 1))
    gyg = g(gyc.detach().float()).requires_grad_()
    gxg, = torch.autograd.grad(yg, xg, gyg, create_graph=True)
    ggg, = torch.autograd.grad((gxg * g(vc.float())).sum(), gyg)
    report("upfirdn2d backward", gxg, gxc, 2e-6)
    report("upfirdn2d double backward", ggg, ggc, 2e-6)


def t_ops_half():
    """The two native ops on half tensors (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF: upfirdn2d_kernel.cu:225,
    fused_bias_act_kernel.cu:79; round-3 verdict, missing #3): float16 and bfloat16 inputs through the public op functions,
    against the fp32 op on the widened input rounded once (fp32 accumulation, one rounding: bit-exact), forward and gradient."""
    from ppst_amd.stylegan2_op import upfirdn2d, fused_leaky_relu
    torch.manual_seed(31)
    for dt in (torch.float16, torch.bfloat16):
        tag = str(dt).split(".")[-1]
        k4 = torch.tensor([1., 3., 3., 1.]); k4 = (k4[:, None] * k4[None, :] / 64)
        for name, kw in (("blur pad (2,1)", dict(pad=(2, 1))), ("up 2 pad (2,1)", dict(up=2, pad=(2, 1))), ("down 2 pad (1,1)", dict(down=2, pad=(1, 1)))):
            x = g(torch.randn(2, 5, 23, 19)).to(dt).requires_grad_(True)
            xf = x.detach().float().requires_grad_(True)
            y = upfirdn2d(x, g(k4).to(dt), **kw)
            yf = upfirdn2d(xf, g(k4).to(dt).float(), **kw)
            ok = y.dtype == dt and bool(torch.equal(y.detach(), yf.detach().to(dt)))
            RES.append(("upfirdn2d %s %s forward == fp32 op rounded" % (tag, name), ok))
            print("upfirdn2d %-9s %-18s %s max diff %.3e" % (tag, name, "ok  " if ok else "FAIL", (y.float() - yf).abs().max().item()), flush=True)
            gy = g(torch.randn(y.shape)).to(dt)
            y.backward(gy); yf.backward(gy.float())
            report("upfirdn2d %s %s gradient" % (tag, name), x.grad.float(), xf.grad, 2e-2 if dt == torch.bfloat16 else 3e-3)
        x = g(torch.randn(3, 8, 11, 13)).to(dt).requires_grad_(True)
        b = g(torch.randn(8)).to(dt).requires_grad_(True)
        xf, bf = x.detach().float().requires_grad_(True), b.detach().float().requires_grad_(True)
        y, yf = fused_leaky_relu(x, b), fused_leaky_relu(xf, bf)
        ok = y.dtype == dt and bool(torch.equal(y.detach(), yf.detach().to(dt)))
        RES.append(("fused_leaky_relu %s forward == fp32 op rounded" % tag, ok))
        print("fused_leaky_relu %-9s %s max diff %.3e" % (tag, "ok  " if ok else "FAIL", (y.float() - yf).abs().max().item()), flush=True)
        gy = g(torch.randn(y.shape)).to(dt)
        y.backward(gy); yf.backward(gy.float())
        # (the gate of the gradient is the sign of the saved OUTPUT: identical in both runs unless rounding moved an output to 0)
        report("fused_leaky_relu %s input gradient" % tag, x.grad.float(), xf.grad, 2e-2 if dt == torch.bfloat16 else 3e-3)
        report("fused_leaky_relu %s bias gradient" % tag, b.grad.float(), bf.grad, 5e-2 if dt == torch.bfloat16 else 1e-2)


def t_ops_f64():
    """The two native ops in double (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF: upfirdn2d_kernel.cu:225,
    fused_bias_act_kernel.cu:79; round-4 verdict, missing #4) against tests/golden/ops_f64.npz -- the reference's own fallbacks on
    float64 tensors (oracle/gen_golden.py:gen_ops_f64): forward 1e-13, fused_leaky_relu gradients 1e-13; and
    torch.autograd.gradcheck of both public ops, which is what needs double."""
    from ppst_amd.stylegan2_op import upfirdn2d, fused_leaky_relu
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ops_f64.npz"))
    for i in range(int(gold["upfirdn2d.n"])):
        x, k = torch.from_numpy(gold["upfirdn2d.%d.x" % i]), torch.from_numpy(gold["upfirdn2d.%d.k" % i])
        u, d, p0, p1 = (int(v) for v in gold["upfirdn2d.%d.cfg" % i])
        y = upfirdn2d(g(x), g(k), up=u, down=d, pad=(p0, p1))
        RES.append(("upfirdn2d f64 case %d dtype" % i, y.dtype == torch.float64))
        report("upfirdn2d f64 case %d (up %d down %d pad %d,%d)" % (i, u, d, p0, p1), y, torch.from_numpy(gold["upfirdn2d.%d.y" % i]), 1e-13)
    # slope and scale cross the op boundary as C floats (fused_bias_act.cpp:4-20: ``float alpha, float scale``, cast to scalar_t in
    # the kernel, fused_bias_act_kernel.cu:19-49) -- here as in the reference's CUDA op; the golden comes from the reference's
    # Python fallback, which multiplies by the Python doubles 0.2 and 2 ** 0.5: the two differ by float(2 ** 0.5) / 2 ** 0.5 - 1 =
    # 1.7e-8.  Bar against the golden: 3e-8; against the same formula with the float-rounded constants: 1e-15 (double arithmetic).
    x = g(torch.from_numpy(gold["flrelu.x"])).requires_grad_(True)
    b = g(torch.from_numpy(gold["flrelu.b"])).requires_grad_(True)
    y = fused_leaky_relu(x, b)
    RES.append(("fused_leaky_relu f64 dtype", y.dtype == torch.float64))
    report("fused_leaky_relu f64 forward vs the reference fallback", y, torch.from_numpy(gold["flrelu.y"]), 3e-8)
    sl, sc = float(np.float32(0.2)), float(np.float32(2 ** 0.5))
    t = torch.from_numpy(gold["flrelu.x"]) + torch.from_numpy(gold["flrelu.b"]).view(1, -1, 1, 1)
    report("fused_leaky_relu f64 forward, float-rounded slope / scale", y, torch.where(t > 0, t, t * sl) * sc, 1e-15)
    gy = torch.from_numpy(gold["flrelu.g"])
    gx, gb = torch.autograd.grad(y, [x, b], g(gy))
    report("fused_leaky_relu f64 grad_x vs the reference fallback", gx, torch.from_numpy(gold["flrelu.gx"]), 3e-8)
    report("fused_leaky_relu f64 grad_b vs the reference fallback", gb, torch.from_numpy(gold["flrelu.gb"]), 3e-8)
    report("fused_leaky_relu f64 grad_x, float-rounded slope / scale", gx, torch.where(t > 0, gy, gy * sl) * sc, 1e-15)
    torch.manual_seed(5)
    xs = g(torch.randn(1, 2, 6, 5, dtype=torch.float64)).requires_grad_(True)
    ks = g(torch.randn(3, 3, dtype=torch.float64))
    ok = torch.autograd.gradcheck(lambda t: upfirdn2d(t, ks, up=2, down=1, pad=(2, 1)), (xs,), eps=1e-6, atol=1e-7)
    RES.append(("upfirdn2d f64 gradcheck", bool(ok)))
    xs = g(torch.randn(2, 3, 4, 5, dtype=torch.float64))
    xs = (xs + 0.2 * xs.sign()).requires_grad_(True)           # keep |x + b| away from the kink of the leaky ReLU
    bs = g(torch.zeros(3, dtype=torch.float64)).requires_grad_(True)
    ok = torch.autograd.gradcheck(fused_leaky_relu, (xs, bs), eps=1e-6, atol=1e-7)
    RES.append(("fused_leaky_relu f64 gradcheck", bool(ok)))
    print("gradcheck upfirdn2d / fused_leaky_relu in double:", RES[-2][1], RES[-1][1], flush=True)


def t_fuse_tail():
    """Round 5: apply passes folded into their only consumer, a 1x1 conv (ops.FUSE_TAIL).  (1) ConvPlan(in_res): the streaming 1x1
    kernel reading prelu(a*x + s + res) against ppst_affine_act(res_before_act, PReLU) followed by the plain conv of the same plan;
    (2) ops.torgb_apply against ppst_affine_act(res_up2) + ppst_conv1x1_small_cout, fp32 and half storage; (3) the generator's
    image pass and feature pass with the switch on against off.  Bars: 2e-6 (same fp32 operations; only FMA contraction may differ)."""
    torch.manual_seed(17)
    for (B, H, Wd, ci, co) in [(2, 64, 64, 256, 64), (1, 37, 29, 64, 64), (3, 16, 48, 128, 192)]:
        w = g(torch.randn(co, ci, 1, 1) / math.sqrt(ci))
        x, res = g(torch.randn(B, H, Wd, ci)), g(torch.randn(B, H, Wd, ci))
        ss = g(torch.rand(B, ci, 2) + 0.5)
        slope, bias = g(torch.tensor([0.25])), g(torch.randn(co))
        plan = ops.ConvPlan(w)
        RES.append(("in_res: %d->%d 1x1 plan takes it" % (ci, co), bool(plan.takes_in_res(H, Wd))))
        ref = plan(ops.affine_act(x, ss, res=res, res_before_act=True, act=ops.ACT_PRELU, prelu=slope), bias=bias)
        got = plan(x, bias=bias, in_ss=ss, in_act=ops.ACT_PRELU, in_prelu=slope, in_res=res)
        report("conv1x1 in_res %s %d->%d vs apply pass + conv" % ((B, H, Wd), ci, co), got, ref, 2e-6)
        xs = g(torch.randn(B, H, Wd, ci + 32))[..., 16:16 + ci]         # channel-slice views (pixel stride != channels)
        rs = g(torch.randn(B, H, Wd, ci + 64))[..., 32:32 + ci]
        ref = plan(ops.affine_act(xs, ss, res=rs, res_before_act=True, act=ops.ACT_LRELU), bias=bias)
        got = plan(xs, bias=bias, in_ss=ss, in_act=ops.ACT_LRELU, in_res=rs)
        report("conv1x1 in_res (slice views, lrelu) %d->%d" % (ci, co), got, ref, 2e-6)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        for (B, H, Wd, C) in [(2, 64, 48, 128), (1, 18, 22, 64)]:
            y = g(torch.randn(B, H, Wd, C)).to(dt)
            skip = g(torch.randn(B, H // 2, Wd // 2, C)).to(dt)
            ss = g(torch.rand(B, C, 2) + 0.5)
            w, bias = g(torch.randn(3, C, 1, 1)), g(torch.randn(3))
            ref = ops.conv1x1_small_cout(ops.affine_act(y, ss, res=skip, out_scale=0.7071, res_up2=True, out_dtype=torch.float32), w, bias, 0.1)
            got = ops.torgb_apply(y, ss, skip, 0.7071, w, bias, 0.1)
            report("torgb_apply %s %s vs apply pass + ToRGB conv" % (str(dt).split(".")[-1], (B, H, Wd, C)), got, ref, 3e-6)
            got = ops.torgb_apply(y, ss, None, 1.0, w, bias, 0.1)
            report("torgb_apply %s no residual" % str(dt).split(".")[-1], got, ops.conv1x1_small_cout(ops.affine_act(y, ss, out_dtype=torch.float32), w, bias, 0.1), 3e-6)
    from ppst_amd.ppst_model import create_model
    sd = W.make_state_dict(2, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    m = create_model(state_dict=sd)
    noise = {k: v.to(dev) for k, v in W.make_noise(4, 2).items()}
    sp = g(torch.randn(2, 256, 64, 64))
    gl = [g(torch.randn(2, 2048)) for _ in range(4)]
    prev = dict(ops.FUSE_TAIL)
    try:
        outs = {}
        for on in (False, True):
            ops.FUSE_TAIL.update(value=on, torgb=on)
            with torch.no_grad():
                outs[on] = (m.G(sp, gl, noise=noise), m.G(sp, gl, extract_features=True, noise=noise))
        report("generator image pass, tail fused vs passes", outs[True][0], outs[False][0], 1e-5)
        for name, a, b in zip(("rgb", "feat", "feat1"), outs[True][1], outs[False][1]):
            report("generator feature pass %s, tail fused vs passes" % name, a, b, 1e-5)
    finally:
        ops.FUSE_TAIL.update(prev)


def t_conv_ksplit():
    """ppst_conv_args.ksplit (round 5): the across-block K split of small grids against float64 torch at the bar of the unsplit
    kernels, beside the unsplit launch (same values up to fp32 summation order), with every epilogue option, normalise-on-load,
    the stride-2 input-gradient scatter, half storage; the give-up marker of the flag wait stays clear."""
    torch.manual_seed(11)
    lrelu = lambda t: F.leaky_relu(t, 0.2) * math.sqrt(2.0)
    prevk, prevw, prevp, prevb = dict(ops.KSPLIT), dict(ops.WINO), ops.PRECISION["value"], ops.BATCH_AWARE["value"]
    ops.BATCH_AWARE["value"] = True          # (the split is a batch-aware choice: the train step's)
    ops.WINO["ksplit_fill"] = 0              # (every shape below that the Winograd kernel takes stays on it)
    try:
        for variants, wino, tag in (((0,), False, "tile"), (ops.KSPLIT["variants"], True, "prod")):
            ops.WINO["value"] = wino
            for name, B, ci, co, H, Wd in [("256->256 64x64 B2", 2, 256, 256, 64, 64), ("512->512 16x16 B4", 4, 512, 512, 16, 16),
                                           ("512->512 8x8 B4", 4, 512, 512, 8, 8), ("512->512 4x4 B5", 5, 512, 512, 4, 4),
                                           ("384->256 33x20 B1 ragged", 1, 384, 256, 33, 20), ("256->64 32x32 B2 (bn 64)", 2, 256, 64, 32, 32)]:
                x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9)
                bias = torch.randn(co); noise = torch.randn(B, 1, H, Wd); res = torch.randn(B, co, H, Wd)
                conv = F.conv2d(x.double(), w.double(), padding=1)
                ref = lrelu(conv + 0.3 * noise.double() + bias.double().view(1, -1, 1, 1))
                plan = ops.ConvPlan(g(w))
                outs = {}
                for on in (False, True):
                    ops.KSPLIT.update(value=on, variants=variants)
                    outs[on] = plan(g(nhwc(x)), bias=g(bias), noise=g(noise), noise_weight=0.3, act=ops.ACT_LRELU, stats=True)
                y, st = outs[True]
                report("ksplit %s %s vs float64" % (tag, name), nchw(y.cpu()), ref, 3e-5)
                report("ksplit %s %s vs unsplit" % (tag, name), y.cpu(), outs[False][0].cpu(), 3e-6)
                report("ksplit %s %s stats" % (tag, name), st.cpu().double().sum(1)[..., 1], (ref ** 2).sum((2, 3)), 1e-5)
        ops.KSPLIT.update(value=True, variants=prevk["variants"])
        ops.WINO.update(prevw)
        ops.WINO["ksplit_fill"] = 0
        # a launch that really splits (the choice is the host's: say so)
        a_ = lambda B, H, Wd, co, bn, nst, ch: ops._ksplit_choice(B * ((H + 15) // 16) * ((Wd + 15) // 16) * -(-co // bn), list(range(0, nst + 1, ch)), 256, 16)[0]
        assert a_(2, 64, 64, 256, 128, 72, 9) == 4 and a_(4, 4, 4, 512, 128, 144, 9) == 8 and a_(8, 512, 512, 128, 128, 36, 9) == 0
        # chunks of unequal length (the stride-2 conv on the space-to-depth tensor: 4 / 2 / 2 / 2 taps per phase): cuts at chunk starts
        s2d_starts = [4 * i for i in range(16)] + [64 + 2 * i for i in range(48)] + [160]
        S_, cuts_ = ops._ksplit_choice(4, s2d_starts, 256, 16)
        assert S_ == 8 and cuts_ == [0, 20, 40, 60, 80, 100, 120, 140, 160] and all(c in s2d_starts for c in cuts_)
        ws2 = torch.randn(512, 512, 3, 3) / 68.0
        xs2 = torch.randn(4, 512, 16, 16)
        kb = torch.tensor([1., 3., 3., 1.]); kb2 = kb[:, None] * kb[None, :]; kb2 = kb2 / kb2.sum()
        xs2b, bhw2 = ops.blur_nhwc(g(nhwc(xs2)), g(kb2), 2, 2, ops.PAD_ZERO, s2d=True)        # as ConvLayer(downsample=True): 17 x 17 blurred
        ps2 = ops.ConvPlan(g(ws2), "s2d")
        outs2 = {}
        for on in (False, True):
            ops.KSPLIT["value"] = on
            outs2[on] = ps2(xs2b, out_hw=((bhw2[0] - 3) // 2 + 1, (bhw2[1] - 3) // 2 + 1)).cpu()
        report("ksplit s2d 512->512 (unequal chunks) vs float64", nchw(outs2[True]), F.conv2d(O.upfirdn2d(xs2, kb2, pad=(2, 2)).double(), ws2.double(), stride=2), 3e-5)
        report("ksplit s2d 512->512 vs unsplit", outs2[True], outs2[False], 3e-6)
        B, ci, co, H, Wd = 2, 256, 256, 32, 32
        x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / 48.0
        bias = torch.randn(co); res = torch.randn(B, co, H, Wd); a1 = torch.tensor([0.25])
        conv = F.conv2d(x.double(), w.double(), padding=1)
        plan = ops.ConvPlan(g(w))
        y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_PRELU, prelu=g(a1), residual=g(nhwc(res)))
        t = conv + bias.double().view(1, -1, 1, 1) + res.double()
        report("ksplit residual before prelu", nchw(y.cpu()), torch.where(t >= 0, t, 0.25 * t), 3e-5)
        y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_LRELU, residual=g(nhwc(res)), res_after_act=True, out_scale=0.5)
        report("ksplit residual after act * scale", nchw(y.cpu()), (lrelu(conv + bias.double().view(1, -1, 1, 1)) + res.double()) * 0.5, 3e-5)
        ss = torch.stack([torch.rand(B, ci) + 0.5, torch.randn(B, ci)], -1).contiguous()
        xn = x.double() * ss[..., 0].double().view(B, ci, 1, 1) + ss[..., 1].double().view(B, ci, 1, 1)
        for pm in (0, 1, 2):
            y = plan(g(nhwc(x)), pad_mode=pm, in_ss=g(ss), in_act=ops.ACT_LRELU)
            xp = F.pad(lrelu(xn), (1, 1, 1, 1), mode={0: "constant", 1: "reflect", 2: "replicate"}[pm])
            report("ksplit normalise-on-load pad %d" % pm, nchw(y.cpu()), F.conv2d(xp, w.double()), 3e-5)
        pd = ops.ConvPlan(g(w), kind="dgrad")
        dy = torch.randn(B, co, H, Wd)
        report("ksplit dgrad", nchw(pd(g(nhwc(dy))).cpu()), F.conv_transpose2d(dy.double(), w.double(), padding=1), 3e-5)
        ws = torch.randn(256, 256, 3, 3) / 48.0          # stride-2 conv on a 33x33 tensor: its scattered input gradient
        pg = ops.ConvPlan(g(ws), kind="dgrad_s2d")
        dys = torch.randn(B, 256, 16, 16)
        report("ksplit dgrad_s2d", nchw(pg(g(nhwc(dys)), out_hw=(33, 33)).cpu()), F.conv_transpose2d(dys.double(), ws.double(), stride=2), 3e-5)
        # single-pass bf16 on bf16-stored activations (the train step's mode): against the unsplit launch, one bf16 ulp
        ops.set_precision(1)
        wb = torch.randn(256, 256, 3, 3) / 48.0
        xb = g(nhwc(torch.randn(2, 256, 64, 64))).to(torch.bfloat16)
        pb = ops.ConvPlan(g(wb))
        outs = {}
        for on in (False, True):
            ops.KSPLIT["value"] = on
            outs[on] = pb(xb, bias=g(bias), act=ops.ACT_LRELU).float().cpu()
        report("ksplit bf16 storage vs unsplit", outs[True], outs[False], 8e-3)
        report("ksplit bf16 storage vs float64", nchw(outs[True]), lrelu(F.conv2d(nchw(xb.float().cpu()).double(), wb.double(), padding=1) + bias.double().view(1, -1, 1, 1)), 2e-2)
        ops.set_precision(prevp)
        rc = ops.lib.ppst_conv_ksplit_check(ops._stream())
        RES.append(("ksplit wait marker clear", rc == 0))
        print("ksplit give-up marker: %d (0 = every wait ended on its flag)" % rc, flush=True)
    finally:
        ops.KSPLIT.update(prevk); ops.WINO.update(prevw); ops.set_precision(prevp); ops.BATCH_AWARE["value"] = prevb


def t_gmp_multi():
    """Round 5: the multi-head GAP || GMP (ppst_gap_gmp_multi / _bwd: the plain and the three class-masked poolings of one feature map in
    one read) against the single-head kernels: forward bit-equal per head, backward = the sum of the four single-head gradients
    (1e-6: another summation order over the heads), with and without the unmasked head; the fused linear backward (ReLU gate on the
    input-gradient reduction, relu(x) and the bias column sums inside the weight-gradient launch) against the separate launches."""
    torch.manual_seed(23)
    for (B, H, Wd, C) in [(2, 64, 64, 32), (1, 16, 48, 256), (3, 32, 32, 64)]:
        x = g(torch.randn(B, H, Wd, C))
        lab = torch.randint(0, 3, (B, H, Wd))
        masks = g(F.one_hot(lab, 3).float())
        for plain in (True, False):
            v = ops.gap_gmp_multi(x, masks, plain)
            heads = ([None] if plain else []) + [masks[..., i].contiguous() for i in range(3)]
            ref = torch.cat([ops.gap_gmp(x, m) for m in heads], 0)
            ok = bool(torch.equal(v, ref))
            RES.append(("gap_gmp_multi %s plain=%s forward bit-equal to the single-head launches" % ((B, H, Wd, C), plain), ok))
            print("gap_gmp_multi fwd %s plain=%s %s" % ((B, H, Wd, C), plain, "ok" if ok else "FAIL %.3e" % (v - ref).abs().max().item()), flush=True)
            gg = g(torch.randn(v.shape))
            dx = ops.gap_gmp_multi_bwd(x, masks, v, gg, plain)
            dref = None
            for h, m in enumerate(heads):
                dref = ops.gap_gmp_bwd(x, m, ref[h * B:(h + 1) * B], gg[h * B:(h + 1) * B].contiguous(), out=dref)
            report("gap_gmp_multi %s plain=%s backward vs summed single-head gradients" % ((B, H, Wd, C), plain), dx, dref, 1e-6)
    for (B, N, K, relu) in [(16, 1024, 32, True), (8, 2048, 1024, True), (4, 512, 2048, False), (32, 2048, 2048, True)]:
        dy, x, w = g(torch.randn(B, N)), g(torch.randn(B, K)), g(torch.randn(N, K))
        gate = lambda t: ops.fused_bias_act_raw(t, None, x, 3, 1, 0.0, 1.0)
        xin = gate(x) if relu else x
        dw_ref, db_ref = ops.linear_wgrad(dy, xin, 0.37), ops.colsum(dy, 1.5)
        dw, db = ops.linear_wgrad_fused(dy, x, 0.37, relu_in=relu, want_bias=True, bias_scale=1.5)
        report("linear_wgrad_fused (%d,%d,%d) relu=%s dW" % (B, N, K, relu), dw, dw_ref, 1e-6)
        report("linear_wgrad_fused (%d,%d,%d) db" % (B, N, K), db, db_ref, 1e-5)
        acc_w, acc_b = g(torch.randn(N, K)), g(torch.randn(N))
        w0, b0 = acc_w.clone(), acc_b.clone()
        ops.linear_wgrad_fused(dy, x, 0.37, out=acc_w, accumulate=True, relu_in=relu, bias_out=acc_b, bias_scale=1.5, bias_accumulate=True)
        report("linear_wgrad_fused accumulate dW", acc_w, w0 + dw_ref, 1e-6)
        report("linear_wgrad_fused accumulate db", acc_b, b0 + db_ref, 1e-5)
        dx_ref = gate(ops.linear_dgrad(dy, w, 0.5))
        report("linear_dgrad_gate (%d,%d,%d)" % (B, N, K), ops.linear_dgrad_gate(dy, w, x, 0.5), dx_ref, 1e-6)


def t_train_half():
    """Round 5: the `_st` twins of the BACKWARD kernels (half-precision storage of the training activations and their gradients in
    precision mode 1) against their fp32 forms on the widened inputs: statistics / parameter-gradient outputs (fp32 either way)
    bit-equal, activation-shaped outputs bit-equal to the fp32 result rounded once; the bf16-input weight gradient
    (conv_wgrad_tr2b_kernel) against the fp32-tile single-pass kernel on the same (bf16-representable) values: equal partial sums."""
    torch.manual_seed(29)
    bf = torch.bfloat16
    prev = ops.PRECISION["value"]
    ops.set_precision(1)
    try:
        B, H, Wd, C = 2, 34, 38, 64
        gq, yq, tq = (g(torch.randn(B, H, Wd, C)).to(bf) for _ in range(3))
        gf, yf, tf = gq.float(), yq.float(), tq.float()
        def same(name, a, b):
            ok = bool(torch.equal(a, b))
            RES.append((name, ok))
            print("%-70s %s" % (name, "ok" if ok else "FAIL max diff %.3e" % (a.float() - b.float()).abs().max().item()), flush=True)
        same("dual_stats bf16 == fp32 on the widened inputs", ops.dual_stats(gq, yq, tq), ops.dual_stats(gf, yf, tf))
        coef = g(torch.randn(B, C, 4))
        for pg in (False, True):
            same("in_bwd_apply bf16 (post_gate=%s) == fp32 result rounded" % pg, ops.in_bwd_apply(gq, yq, coef, gate=tq, post_gate=pg),
                 ops.in_bwd_apply(gf, yf, coef, gate=tf, post_gate=pg).to(bf))
        for mode in (ops.PAD_ZERO, ops.PAD_REFLECT, ops.PAD_REPLICATE):
            same("pad2d bf16 mode %d" % mode, ops.pad2d(gq, 1, 1, 1, 1, mode), ops.pad2d(gf, 1, 1, 1, 1, mode).to(bf))
            same("pad2d_bwd bf16 mode %d" % mode, ops.pad2d_bwd(gq, 1, 1, 1, 1, mode), ops.pad2d_bwd(gf, 1, 1, 1, 1, mode).to(bf))
        same("pad2d_bwd bf16 asymmetric reflect (2, 1)", ops.pad2d_bwd(gq, 2, 1, 2, 1, ops.PAD_REFLECT), ops.pad2d_bwd(gf, 2, 1, 2, 1, ops.PAD_REFLECT).to(bf))
        same("space_to_depth bf16", ops.space_to_depth(gq), ops.space_to_depth(gf).to(bf))
        same("bilinear_bwd bf16 (x2)", ops.bilinear_bwd(gq, H // 2, Wd // 2), ops.bilinear_bwd(gf, H // 2, Wd // 2).to(bf))
        same("colsum bf16", ops.colsum(gq.view(-1, C), 0.5), ops.colsum(gf.view(-1, C), 0.5))
        nz = g(torch.randn(B, 1, H, Wd))
        same("noise_wgrad bf16", ops.noise_wgrad(gq, nz), ops.noise_wgrad(gf, nz))
        img = g(torch.randn(B, H, Wd, 3))
        same("wgrad_small_cin bf16 dy", ops.wgrad_small_cin(img, gq, 0.3), ops.wgrad_small_cin(img, gf, 0.3))
        x64 = g(torch.randn(2, 64, 64, C)).to(bf)
        m1 = g((torch.rand(2, 64, 64) > 0.5).float())
        v = ops.gap_gmp(x64, m1)
        same("gap_gmp bf16 forward == fp32 on the widened input", v, ops.gap_gmp(x64.float(), m1))
        gv = g(torch.randn(2, 2 * C))
        same("gap_gmp_bwd bf16", ops.gap_gmp_bwd(x64, m1, v, gv), ops.gap_gmp_bwd(x64.float(), m1, v, gv).to(bf))
        masks = g(F.one_hot(torch.randint(0, 3, (2, 64, 64)), 3).float())
        vm = ops.gap_gmp_multi(x64, masks, True)
        same("gap_gmp_multi bf16 forward", vm, ops.gap_gmp_multi(x64.float(), masks, True))
        gm = g(torch.randn(vm.shape))
        same("gap_gmp_multi_bwd bf16", ops.gap_gmp_multi_bwd(x64, masks, vm, gm, True), ops.gap_gmp_multi_bwd(x64.float(), masks, vm, gm, True).to(bf))
        k4 = torch.tensor([1., 3., 3., 1.]); k4 = g((k4[:, None] * k4[None, :] / 64))
        same("upfirdn2d up 2 (NHWC, 64 channels) bf16", ops.upfirdn2d_raw(gq, k4, 2, 2, 1, 1, 2, 1, 2, 1), ops.upfirdn2d_raw(gf, k4, 2, 2, 1, 1, 2, 1, 2, 1).to(bf))
        # the weight gradient: every table family (3x3, stride-2 s2d, fused upscale's dgradT, 1x1 with 4 / 2 chunks per block, thin cout)
        for name, kind, ci, co, k, Hh, Ww in [("3x3 64->128", "conv", 64, 128, 3, 36, 40), ("3x3 32->32 (thin)", "conv", 32, 32, 3, 33, 70),
                                              ("1x1 128->64", "conv", 128, 64, 1, 32, 48), ("1x1 64->256", "conv", 64, 256, 1, 20, 36),
                                              ("s2d 64->128", "s2d", 64, 128, 3, 33, 37), ("3x3 96->160 (ragged)", "conv", 96, 160, 3, 30, 34)]:
            w = g(torch.randn(co, ci, k, k))
            plan = ops.ConvPlan(w, kind=kind, scale=0.7, precision=1)
            if kind == "s2d":
                xx = g(torch.randn(2, (Hh + 1) // 2, (Ww + 1) // 2, 4 * ci)).to(bf)
                oh, ow = (Hh - 3) // 2 + 1, (Ww - 3) // 2 + 1
            else:
                xx = g(torch.randn(2, Hh, Ww, ci)).to(bf)
                oh, ow = Hh, Ww
            dy = g(torch.randn(2, oh, ow, co)).to(bf)
            dw_h, db_h = ops.conv_wgrad(plan, xx, dy, want_bias=True)
            dw_f, db_f = ops.conv_wgrad(plan, xx.float(), dy.float(), want_bias=True)
            report("conv_wgrad bf16-stored operands %s vs fp32 tiles (single pass)" % name, dw_h, dw_f, 1e-6)
            report("conv_wgrad bf16-stored operands %s bias column sums" % name, db_h, db_f, 1e-6)
    finally:
        ops.set_precision(prev)


def t_layout_misc():
    torch.manual_seed(1)
    for (B, C, H, Wd) in [(2, 3, 17, 19), (1, 32, 64, 64), (2, 70, 9, 33)]:
        x = torch.randn(B, C, H, Wd)
        report("nchw_to_nhwc %s" % ((B, C, H, Wd),), ops.nchw_to_nhwc(g(x)), nhwc(x), 0)
        report("nhwc_to_nchw %s" % ((B, C, H, Wd),), ops.nhwc_to_nchw(g(nhwc(x))), x, 0)
    a, b = torch.randn(3, 2048), torch.randn(3, 2048)
    report("lerp", ops.lerp(g(a), g(b), 0.3), O.lerp(a, b, 0.3), 0)
    report("l2norm mode0 (util.normalize)", ops.l2norm_rows(g(a), 1e-8, 0), O.normalize(a), 1e-6)
    report("l2norm mode1 (F.normalize)", ops.l2norm_rows(g(a), 1e-12, 1), F.normalize(a), 1e-6)
    x = torch.rand(2, 3, 33, 31) * 2.6 - 1.3
    y = ops.tensor2im_u8(g(x)).cpu().numpy()
    RES.append(("tensor2im exact", bool(np.array_equal(y, O.tensor2im(x)))))
    print("tensor2im exact:", np.array_equal(y, O.tensor2im(x)), flush=True)
    for (B, K, N, relu, act) in [(1, 2048, 512, False, 0), (8, 2048, 1024, False, 0), (3, 64, 32, False, 0), (5, 32, 1024, True, 0),
                                 (2, 8192, 512, False, 1), (17, 100, 7, True, 1)]:
        x = torch.randn(B, K); w = torch.randn(N, K); bb = torch.randn(N)
        xr = F.relu(x) if relu else x
        ref = F.linear(xr, w * 0.1, bb * 0.5)
        if act:
            ref = O.fused_leaky_relu(ref, None)
        report("linear B%d K%d N%d relu%d act%d" % (B, K, N, relu, act), ops.linear(g(x), g(w), g(bb), 0.1, 0.5, relu, act), ref, 3e-6)
    x = torch.randn(2, 16, 16, 256); sc = torch.randn(2, 256); bi = torch.randn(2, 256)
    report("spatial_modulation", ops.spatial_modulation(g(x), g(sc), g(bi)), x * sc[:, None, None] + bi[:, None, None], 1e-6)


def conv_ref(x, w, kind, pad_mode):
    mode = {0: "constant", 1: "reflect", 2: "replicate"}[pad_mode]
    if kind == "conv":
        k = w.shape[2]
        if k == 3:
            x = F.pad(x, (1, 1, 1, 1), mode=mode)
        return F.conv2d(x, w)
    if kind == "convT":
        return F.conv_transpose2d(x, O.upscale_weight(w), stride=2, padding=1)
    raise ValueError


def t_conv():
    torch.manual_seed(2)
    cases = [
        # name, B, Cin, Cout, H, W, k, kind, pad_mode
        ("3x3 zero 64->128 32x32", 2, 64, 128, 32, 32, 3, "conv", 0),
        ("3x3 zero 32->32 40x24 (ragged)", 1, 32, 32, 40, 24, 3, "conv", 0),
        ("3x3 reflect 32->64 33x47", 2, 32, 64, 33, 47, 3, "conv", 1),
        ("3x3 replicate 128->64 16x16", 1, 128, 64, 16, 16, 3, "conv", 2),
        ("3x3 zero 256->384 16x16", 1, 256, 384, 16, 16, 3, "conv", 0),
        ("1x1 64->128 20x20", 2, 64, 128, 20, 20, 1, "conv", 0),
        ("1x1 256->64 8x8", 1, 256, 64, 8, 8, 1, "conv", 0),
        ("convT 64->128 16x16", 2, 64, 128, 16, 16, 3, "convT", 0),
        ("convT 128->64 20x12", 1, 128, 64, 20, 12, 3, "convT", 0),
        ("3x3 zero 512->512 8x8", 1, 512, 512, 8, 8, 3, "conv", 0),
    ]
    for name, B, ci, co, H, Wd, k, kind, pm in cases:
        x = torch.randn(B, ci, H, Wd)
        w = torch.randn(co, ci, k, k) / math.sqrt(ci * k * k)
        ref = conv_ref(x.double(), w.double(), kind, pm)
        plan = ops.ConvPlan(g(w), kind=kind)
        y = plan(g(nhwc(x)), pad_mode=pm)
        report("conv bf16x3 " + name, nchw(y.cpu()), ref, 3e-5)
        plan1 = ops.ConvPlan(g(w), kind=kind, precision=1)
        y1 = plan1(g(nhwc(x)), pad_mode=pm)
        report("conv bf16x1 " + name, nchw(y1.cpu()), ref, 2e-2)
        plan3 = ops.ConvPlan(g(w), kind=kind, precision=3)      # single-pass fp16 (11 significant bits per operand)
        report("conv fp16x1 " + name, nchw(plan3(g(nhwc(x)), pad_mode=pm).cpu()), ref, 3e-3)
        if ops.EXPERIMENTS:                                      # (experiment kernels: PPST_EXPERIMENTS=1 builds only)
            plan4 = ops.ConvPlan(g(w), kind=kind, precision=4)      # two-pass fp16: activation hi + lo, weight rounded to fp16
            report("conv fp16x2 " + name, nchw(plan4(g(nhwc(x)), pad_mode=pm).cpu()), ref, 3e-4)
        plan2 = ops.ConvPlan(g(w), kind=kind, precision=2)      # exact-fp32 verification kernel (conv_f32.hip)
        st_ = plan2(g(nhwc(x)), pad_mode=pm, stats=True)
        report("conv fp32  " + name, nchw(st_[0].cpu()), ref, 5e-6)     # fp32 fmaf chain over up to 4608 terms
        report("conv fp32 stats " + name, st_[1].sum(1)[..., 0].cpu(), ref.sum((2, 3)), 2e-5)
    # scale folded into the pack
    x = torch.randn(1, 64, 16, 16); w = torch.randn(64, 64, 3, 3)
    y = ops.ConvPlan(g(w), scale=0.05)(g(nhwc(x)))
    report("conv weight scale", nchw(y.cpu()), F.conv2d(x.double(), w.double() * 0.05, padding=1), 3e-5)
    # stride-2 3x3 over blur output (s2d) == blur + conv stride 2
    x = torch.randn(2, 32, 40, 40); w = torch.randn(64, 32, 3, 3) / 17.0
    k3 = O.make_kernel([1, 2, 1])
    xb = O.upfirdn2d(F.pad(x, (2, 1, 2, 1), mode="reflect"), k3)
    ref = F.conv2d(xb.double(), w.double(), stride=2)
    yb, bhw = ops.blur_nhwc(g(nhwc(x)), g(k3), 2, 1, ops.PAD_REFLECT, s2d=True)
    ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
    y = ops.ConvPlan(g(w), kind="s2d")(yb, out_hw=ohw)
    report("conv s2d stride-2 32->64 (blur 41->20)", nchw(y.cpu()), ref, 3e-5)
    x = torch.randn(1, 64, 34, 34); w = torch.randn(128, 64, 3, 3) / 24.0
    k4 = O.make_kernel([1, 3, 3, 1])
    xb = O.upfirdn2d(x, k4, pad=(2, 2))
    ref = F.conv2d(xb.double(), w.double(), stride=2)
    yb, bhw = ops.blur_nhwc(g(nhwc(x)), g(k4), 2, 2, ops.PAD_ZERO, s2d=True)
    ohw = ((bhw[0] - 3) // 2 + 1, (bhw[1] - 3) // 2 + 1)
    y = ops.ConvPlan(g(w), kind="s2d")(yb, out_hw=ohw)
    report("conv s2d stride-2 64->128 k4 (35->17)", nchw(y.cpu()), ref, 3e-5)
    # epilogue: bias + noise + lrelu + stats; residual before/after; prelu; out slice
    B, ci, co, H, Wd = 2, 64, 128, 24, 40
    x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / 24.0
    bias = torch.randn(co); noise = torch.randn(B, 1, H, Wd); res = torch.randn(B, co, H, Wd)
    conv = F.conv2d(x.double(), w.double(), padding=1)
    ref = O.fused_leaky_relu(conv + 0.3 * noise.double() + bias.double().view(1, -1, 1, 1), None)
    plan = ops.ConvPlan(g(w))
    y, st = plan(g(nhwc(x)), bias=g(bias), noise=g(noise), noise_weight=0.3, act=ops.ACT_LRELU, stats=True)
    report("conv epilogue bias+noise+lrelu", nchw(y.cpu()), ref, 3e-5)
    s = st.cpu().double().sum(1)
    report("conv tile stats sum", s[..., 0], ref.sum((2, 3)), 1e-5)
    report("conv tile stats sumsq", s[..., 1], (ref ** 2).sum((2, 3)), 1e-5)
    ss = ops.in_finalize(st, H * Wd)
    report("in_finalize+affine == instance_norm", nchw(ops.affine_act(y, ss).cpu()), O.instance_norm(ref), 3e-5)
    y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_LRELU, residual=g(nhwc(res)), res_after_act=True, out_scale=0.5)
    report("conv residual after act * scale", nchw(y.cpu()), (O.fused_leaky_relu(conv + bias.double().view(1, -1, 1, 1), None) + res.double()) * 0.5, 3e-5)
    a = torch.tensor([0.25])
    y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_PRELU, prelu=g(a), residual=g(nhwc(res)))
    report("conv residual before prelu", nchw(y.cpu()), O.prelu(conv + bias.double().view(1, -1, 1, 1) + res.double(), a.double()), 3e-5)
    big = torch.zeros(B, H, Wd, 200, device=dev)
    plan(g(nhwc(x)), out=big[..., 40:168])
    report("conv out slice", nchw(big[..., 40:168].contiguous().cpu()), conv, 3e-5)
    report("conv out slice untouched", big[..., :40], torch.zeros(B, H, Wd, 40), 0)
    # normalise-on-load: conv(act(a*x+s)) with zero / replicate padding of the normalised tensor
    B, ci, co, H, Wd = 2, 64, 128, 24, 40
    x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / 24.0
    ssin = torch.randn(B, ci, 2)
    xa = x.double() * ssin[:, :, 0, None, None].double() + ssin[:, :, 1, None, None].double()
    y = ops.ConvPlan(g(w))(g(nhwc(x)), in_ss=g(ssin))
    report("conv normalise-on-load (zero pad)", nchw(y.cpu()), F.conv2d(xa, w.double(), padding=1), 3e-5)
    a = torch.tensor([0.25])
    y = ops.ConvPlan(g(w))(g(nhwc(x)), in_ss=g(ssin), in_act=ops.ACT_PRELU, in_prelu=g(a), pad_mode=ops.PAD_REPLICATE)
    report("conv normalise-on-load prelu (replicate)", nchw(y.cpu()), F.conv2d(F.pad(O.prelu(xa, a.double()), (1, 1, 1, 1), mode="replicate"), w.double()), 3e-5)
    y = ops.ConvPlan(g(w))(g(nhwc(x)), in_ss=g(ssin), in_act=ops.ACT_LRELU)
    report("conv normalise-on-load lrelu", nchw(y.cpu()), F.conv2d(O.fused_leaky_relu(xa, None), w.double(), padding=1), 3e-5)
    # cold-cache determinism (regression: an LDS-DMA weight copy once raced the barrier when
    # the first launch found cold caches): flush L2/MALL, run, compare with a warm rerun
    x = torch.randn(2, 128, 128, 128); w = torch.randn(128, 128, 3, 3) / 34.0
    xg = g(nhwc(x)); plan = ops.ConvPlan(g(w)); ssg = g(torch.randn(2, 128, 2))
    for trial in range(3):
        junk = torch.empty(160 * 1024 * 1024, device=dev).normal_()  # 640 MB > L2 + Infinity Cache
        torch.cuda.synchronize()
        cold = plan(xg, in_ss=ssg, stats=True)[0].clone()
        warm = plan(xg, in_ss=ssg, stats=True)[0]
        report("conv cold-cache == warm (trial %d)" % trial, cold, warm, 0)
        del junk
    # convT stats cover the 4 phases
    x = torch.randn(1, 64, 16, 16); w = torch.randn(64, 64, 3, 3) / 24.0
    y, st = ops.ConvPlan(g(w), kind="convT")(g(nhwc(x)), stats=True)
    ref = conv_ref(x.double(), w.double(), "convT", 0)
    report("convT stats sum", st.cpu().double().sum(1)[..., 0], ref.sum((2, 3)), 1e-5)
    # small-channel convs
    x = torch.randn(2, 3, 20, 20); w = torch.randn(32, 3, 1, 1); b = torch.randn(32)
    y = ops.conv1x1_small_cin(g(nhwc(x)), g(w), g(b), 1 / math.sqrt(3), ops.ACT_LRELU)
    report("conv1x1 small cin (FromRGB)", nchw(y.cpu()), O.fused_leaky_relu(O.equal_conv2d(x, w), b), 2e-6)
    x = torch.randn(2, 128, 20, 20); w = torch.randn(3, 128, 1, 1); b = torch.randn(3)
    y = ops.conv1x1_small_cout(g(nhwc(x)), g(w), g(b), 1 / math.sqrt(128))
    report("conv1x1 small cout (ToRGB)", nchw(y.cpu()), O.equal_conv2d(x, w, b), 3e-6)


CONV_DEFAULTS = (ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS)


def direct_kernels_only(fn):
    """Run ``fn`` with the Winograd kernel (variant 10), the phase-pair / nine-product upscales and the 64-channel steps switched
    off: the bit-identity checks below compare the DIRECT kernel families with each other (same MFMA sequence per output element);
    those forms are other sums of the same products (each has its own test)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        prev = ops.WINO["value"], ops.DUAL_CONVT["value"], ops.UP9["value"], ops.K64["value"]
        ops.WINO["value"] = ops.DUAL_CONVT["value"] = ops.UP9["value"] = ops.K64["value"] = False
        try:
            return fn(*a, **k)
        finally:
            ops.WINO["value"], ops.DUAL_CONVT["value"], ops.UP9["value"], ops.K64["value"] = prev
    return wrapped


@direct_kernels_only
def t_conv_variants():
    """The fat-wave kernel (conv_mfma2.hip, N tile 128 and 256) against the 8-wave kernel on the same plans: same MFMA
    sequence per output element, so outputs must be bit-identical; tile statistics to rounding (other summation tree)."""
    torch.manual_seed(5)
    nz_ = torch.randn
    cases = [
        # name, B, Cin, Cout, H, W, kind, pad_mode, features
        ("3x3 zero 64->128 48x40 bias+noise+lrelu+stats", 2, 64, 128, 48, 40, "conv", 0, "full"),
        ("3x3 reflect 32->256 33x47", 2, 32, 256, 33, 47, "conv", 1, "plain"),
        ("3x3 replicate 128->512 16x16 in_ss+prelu", 1, 128, 512, 16, 16, "conv", 2, "inss"),
        ("3x3 zero 256->384 24x24 residual", 1, 256, 384, 24, 24, "conv", 0, "res"),
        ("convT 64->256 20x12", 2, 64, 256, 20, 12, "convT", 0, "full"),
        ("convT 128->128 16x16", 1, 128, 128, 16, 16, "convT", 0, "plain"),
        ("dgrad 3x3 (128<-64) 32x32", 2, 64, 128, 32, 32, "dgrad", 0, "plain"),
        ("dgradT (256->64 fwd) 16x16", 1, 64, 256, 16, 16, "dgradT", 0, "plain"),
    ]
    for name, B, ci, co, H, Wd, kind, pm, feat in cases:
        w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
        outs = {}
        # 8-wave; fat N=128; fat N=256; 8-wave 128x64 tiles N=256; two 4-wave blocks per CU, 128x64 tiles, N=128
        tall = dict(ops.TALL_TILE_128)
        ksp = dict(ops.KSPLIT_128)
        t24 = dict(ops.TILE24_128)
        # the production library carries the tile kernel (0), its N-256 form (2) and its 8-row two-block form ("8row": under-filled
        # grids of the train step); variants 1 / 3 / 7 / 8 / 9 are measured-and-off experiments, compiled only with PPST_EXPERIMENTS=1
        todo = ((0, 384), (1, 1 << 30), (1, 0), (2, 0), (3, 0), (7, 0), (8, 0), (9, 0), ("8row", 0)) if ops.EXPERIMENTS else ((0, 384), (2, 0), ("8row", 0))
        row8 = dict(ops.TWO_BLOCK_8ROW)
        for variant, minb in todo:
            if variant == "8row":
                ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = 2, 1 << 30
                ops.TALL_TILE_128["value"] = ops.KSPLIT_128["value"] = ops.TILE24_128["value"] = False
                ops.TWO_BLOCK_8ROW.update(value=True, min_blocks=0)
                plan = ops.ConvPlan(w, kind=kind)
                cin_eff = plan.max_chan + 32
                torch.manual_seed(11)
                x = g(nz_(B, H, Wd, cin_eff))
                kw = {}
                oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
                if feat == "full":
                    kw = dict(bias=g(nz_(plan.cout)), noise=g(nz_(B, 1, oh, ow)), noise_weight=0.3, act=ops.ACT_LRELU)
                elif feat == "inss":
                    kw = dict(in_ss=g(torch.rand(B, cin_eff, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])),
                              act=ops.ACT_PRELU, prelu=g(torch.tensor([0.1])))
                elif feat == "res":
                    kw = dict(residual=g(nz_(B, oh, ow, plan.cout)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
                y, st = plan(x, pad_mode=pm, stats=True, **kw)
                outs[(variant, minb)] = (y.cpu(), st.sum(1).cpu())
                ops.TWO_BLOCK_8ROW.update(row8)
                continue
            ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = variant, minb
            ops.TALL_TILE_128.update(value=variant == 7, min_blocks=0)
            ops.KSPLIT_128.update(value=variant == 8, min_blocks=0)
            ops.TILE24_128.update(value=variant == 9, min_blocks=0, max_waste=10.0)   # (9: on every 128-wide plan, any tile height)
            if variant in (7, 8, 9):  # the 32 x 16-pixel-tile / the K-split kernel on every 128-wide plan (the N-256 tile switched off)
                ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = 2, 1 << 30
            plan = ops.ConvPlan(w, kind=kind)
            cin_eff = plan.max_chan + 32
            torch.manual_seed(11)
            x = g(nz_(B, H, Wd, cin_eff))
            kw = {}
            oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
            if feat == "full":
                kw = dict(bias=g(nz_(plan.cout)), noise=g(nz_(B, 1, oh, ow)), noise_weight=0.3, act=ops.ACT_LRELU)
            elif feat == "inss":
                kw = dict(in_ss=g(torch.rand(B, cin_eff, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])),
                          act=ops.ACT_PRELU, prelu=g(torch.tensor([0.1])))
            elif feat == "res":
                kw = dict(residual=g(nz_(B, oh, ow, plan.cout)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
            y, st = plan(x, pad_mode=pm, stats=True, **kw)
            outs[(variant, minb)] = (y.cpu(), st.sum(1).cpu())
        ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = CONV_DEFAULTS
        ops.TALL_TILE_128.update(tall)
        ops.KSPLIT_128.update(ksp)
        ops.TILE24_128.update(t24)
        # the K-split kernel adds an even-step and an odd-step partial sum: not bit-identical, 2e-6 of the largest output
        if (8, 0) in outs:
            y8, s8 = outs[(8, 0)]
            report("K-split conv (variant 8) %s vs tile kernel" % name, y8, outs[(0, 384)][0], 2e-6)
            report("K-split conv (variant 8) %s stats" % name, s8, outs[(0, 384)][1], 1e-5)
        y0, s0 = outs[(0, 384)]
        for key, tag in (((1, 1 << 30), "N=128"), ((1, 0), "N=256|128"), ((2, 0), "8w N=256"), ((3, 0), "2blk N=128"), ((7, 0), "32x16 N=128"),
                         ((9, 0), "24x16 N=128"), (("8row", 0), "8x16 2blk")):
            if key not in outs:
                continue
            y1, s1 = outs[key]
            RES.append(("fat conv %s %s bit-identical" % (tag, name), bool(torch.equal(y0, y1))))
            print("fat conv %-9s %-52s %s max diff %.3e" % (tag, name, "ok  " if torch.equal(y0, y1) else "FAIL", (y0 - y1).abs().max().item()), flush=True)
            report("fat conv %s %s stats" % (tag, name), s1, s0, 1e-5)


def t_conv_wino():
    """conv_wino.hip (variant 10: Winograd F(2,3) along x, 1.5x fewer MFMAs) against float64 torch for every padding mode, ragged
    and sliced tensors, every epilogue option, normalise-on-load and the input-gradient plan -- the bar of the direct kernels
    (3e-5 of the largest output); NOT bit-identical to them (another sum), so the direct result is printed beside it."""
    torch.manual_seed(5)
    prev = dict(ops.WINO)
    ops.WINO.update(value=True, min_blocks=1)          # the small shapes here on the Winograd kernel too
    try:
        pad_ref = lambda x, pm: F.pad(x, (1, 1, 1, 1), mode={0: "constant", 1: "reflect", 2: "replicate"}[pm])
        lrelu = lambda t: F.leaky_relu(t, 0.2) * math.sqrt(2.0)
        for name, B, ci, co, H, Wd, pm in [("64->128 32x32 zero", 2, 64, 128, 32, 32, 0), ("32->128 40x24 ragged zero", 1, 32, 128, 40, 24, 0),
                                          ("64->256 33x47 reflect", 2, 64, 256, 33, 47, 1), ("128->128 16x16 replicate", 1, 128, 128, 16, 16, 2),
                                          ("256->384 16x16 zero", 1, 256, 384, 16, 16, 0), ("512->512 8x8 zero", 1, 512, 512, 8, 8, 0),
                                          ("128->128 64x64 zero", 3, 128, 128, 64, 64, 0), ("96->160 18x50 zero (cout % 128 != 0)", 1, 96, 160, 18, 50, 0),
                                          ("32->128 3x5 zero (smaller than a tile)", 1, 32, 128, 3, 5, 0)]:
            x = torch.randn(B, ci, H, Wd)
            w = torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9)
            ref = F.conv2d(pad_ref(x.double(), pm), w.double())
            plan = ops.ConvPlan(g(w))
            assert plan.choose_kernel(H, Wd, H, Wd, H, Wd, 1)[0] == 10
            report("wino " + name, nchw(plan(g(nhwc(x)), pad_mode=pm).cpu()), ref, 3e-5)
        B, ci, co, H, Wd = 2, 64, 128, 24, 40
        x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / 24.0
        bias = torch.randn(co); noise = torch.randn(B, 1, H, Wd); res = torch.randn(B, co, H, Wd)
        conv = F.conv2d(x.double(), w.double(), padding=1)
        plan = ops.ConvPlan(g(w))
        y, st = plan(g(nhwc(x)), bias=g(bias), noise=g(noise), noise_weight=0.3, act=ops.ACT_LRELU, stats=True)
        ref = lrelu(conv + 0.3 * noise.double() + bias.double().view(1, -1, 1, 1))
        report("wino epilogue bias+noise+lrelu", nchw(y.cpu()), ref, 3e-5)
        s_ = st.cpu().double().sum(1)
        report("wino tile stats sum", s_[..., 0], ref.sum((2, 3)), 1e-5)
        report("wino tile stats sumsq", s_[..., 1], (ref ** 2).sum((2, 3)), 1e-5)
        report("wino in_finalize+affine == instance_norm", nchw(ops.affine_act(y, ops.in_finalize(st, H * Wd)).cpu()), O.instance_norm(ref), 3e-5)
        y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_LRELU, residual=g(nhwc(res)), res_after_act=True, out_scale=0.5)
        report("wino residual after act * scale", nchw(y.cpu()), (lrelu(conv + bias.double().view(1, -1, 1, 1)) + res.double()) * 0.5, 3e-5)
        a_ = torch.tensor([0.25])
        y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_PRELU, prelu=g(a_), residual=g(nhwc(res)))
        t = conv + bias.double().view(1, -1, 1, 1) + res.double()
        report("wino residual before prelu", nchw(y.cpu()), torch.where(t >= 0, t, 0.25 * t), 3e-5)
        big = torch.zeros(B, H, Wd, 200, device=dev)
        plan(g(nhwc(x)), out=big[..., 40:168])
        report("wino out slice", nchw(big[..., 40:168].contiguous().cpu()), conv, 3e-5)
        report("wino out slice untouched", big[..., :40].cpu(), torch.zeros(B, H, Wd, 40), 0)
        ss = torch.stack([torch.rand(B, ci) + 0.5, torch.randn(B, ci)], -1).contiguous()
        xn = x.double() * ss[..., 0].double().view(B, ci, 1, 1) + ss[..., 1].double().view(B, ci, 1, 1)
        for pm in (0, 1, 2):
            for in_act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_PRELU, lambda t: torch.where(t >= 0, t, 0.25 * t)), (ops.ACT_LRELU, lrelu)):
                y = plan(g(nhwc(x)), pad_mode=pm, in_ss=g(ss), in_act=in_act, in_prelu=g(a_))
                report("wino normalise-on-load pad %d act %d" % (pm, in_act), nchw(y.cpu()), F.conv2d(pad_ref(fn(xn), pm), w.double()), 3e-5)
        xb = torch.randn(B, H, Wd, 160)
        report("wino input slice", nchw(plan(g(xb)[..., 32:96]).cpu()), F.conv2d(xb[..., 32:96].permute(0, 3, 1, 2).double(), w.double(), padding=1), 3e-5)
        wd = torch.randn(128, 256, 3, 3) / 30.0      # forward (Cout 128, Cin 256): its input gradient maps 128 -> 256 channels
        dy = torch.randn(B, 128, H, Wd)
        pd = ops.ConvPlan(g(wd), kind="dgrad")
        assert pd.choose_kernel(H, Wd, H, Wd, H, Wd, 1)[0] == 10
        report("wino dgrad 128->256", nchw(pd(g(nhwc(dy))).cpu()), F.conv_transpose2d(dy.double(), wd.double(), padding=1), 3e-5)
        y = ops.ConvPlan(g(w), scale=0.05)(g(nhwc(x)))
        report("wino weight scale", nchw(y.cpu()), F.conv2d(x.double(), w.double() * 0.05, padding=1), 3e-5)
        # cold caches == warm (the weight fragments come straight from memory), and two launches agree bit for bit
        xg = g(nhwc(torch.randn(4, 128, 64, 64))); wg = g(torch.randn(256, 128, 3, 3) / 34.0)
        pw = ops.ConvPlan(wg)
        y1 = pw(xg).clone()
        _ = torch.empty(256 << 20, device=dev).fill_(1.0)     # 1 GB through the caches
        y2 = pw(xg)
        RES.append(("wino cold == warm", bool(torch.equal(y1, y2))))
        print("wino cold == warm                                           %s" % ("ok" if torch.equal(y1, y2) else "FAIL"), flush=True)
        # a shard of a batch reproduces the batch bit for bit (the kernel choice and the sums do not depend on B)
        ya = pw(xg[1:3])
        RES.append(("wino batch shard bit-identical", bool(torch.equal(ya, y2[1:3]))))
        print("wino batch shard bit-identical                              %s" % ("ok" if torch.equal(ya, y2[1:3]) else "FAIL"), flush=True)
    finally:
        ops.WINO.update(prev)


def t_conv_dual():
    """The fused 4x4 stride-2 upscale with Cout % 128 == 0 as two phase pairs on the N-256 kernel (ppst_conv_args.dual_b) against
    the four-group tile-kernel form of the same plan: the per-element MFMA sequence is the same, so outputs are bit-identical;
    tile statistics to rounding; and against float64 torch.  The same identity in the single-pass modes (precision 1 / 3)."""
    torch.manual_seed(9)
    prev = dict(ops.DUAL_CONVT)
    up9_prev = ops.UP9["value"]
    ops.UP9["value"] = False             # (variant 11 is a different sum: t_conv_up9)
    try:
      for prec in (0, 1, 3):
        for name, B, ci, co, H, Wd, feat in [("convT 64->128 40x24 full", 2, 64, 128, 40, 24, "full"), ("convT 256->128 64x64 plain", 1, 256, 128, 64, 64, "plain"),
                                               ("convT 128->384 17x33 inss (ragged)", 2, 128, 384, 17, 33, "inss"), ("convT 32->128 16x16 res", 1, 32, 128, 16, 16, "res")]:
              w = g(torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9))
              x = g(torch.randn(B, H, Wd, ci))
              kw = {}
              if feat == "full":
                  kw = dict(bias=g(torch.randn(co)), noise=g(torch.randn(B, 1, 2 * H, 2 * Wd)), noise_weight=0.3, act=ops.ACT_LRELU)
              elif feat == "inss":
                  kw = dict(in_ss=g(torch.rand(B, ci, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])))
              elif feat == "res":
                  kw = dict(residual=g(torch.randn(B, 2 * H, 2 * Wd, co)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
              outs = []
              for on in (False, True):
                  ops.DUAL_CONVT.update(value=on, min_blocks=0)
                  plan = ops.ConvPlan(w, kind="convT", precision=prec)
                  assert (plan.choose_kernel(H, Wd, 2 * H, 2 * Wd, H, Wd, 2)[0] == "dual") == on
                  y, st = plan(x, stats=True, **kw)
                  outs.append((y.cpu(), st.sum(1).cpu(), st.shape))
              ok = bool(torch.equal(outs[0][0], outs[1][0])) and outs[0][2] == outs[1][2]
              RES.append(("dual-phase upscale %s precision %d bit-identical to the four-group form" % (name, prec), ok))
              print("dual convT prec %d %-44s %s max diff %.3e" % (prec, name, "ok  " if ok else "FAIL", (outs[0][0] - outs[1][0]).abs().max().item()), flush=True)
              report("dual convT prec %d %s stats" % (prec, name), outs[1][1], outs[0][1], 1e-5)
              if feat == "plain" and prec == 0:
                  ref = conv_ref(nchw(x.cpu()).double(), w.cpu().double(), "convT", 0)
                  report("dual convT %s vs float64" % name, nchw(outs[1][0]), ref, 3e-5)
    finally:
        ops.DUAL_CONVT.update(prev)
        ops.UP9["value"] = up9_prev


def t_conv_up9():
    """The fused 4x4 stride-2 upscale as the un-blurred 3x3 transposed conv + 2x2 box sum in the epilogue (ppst_conv_args.variant 11:
    nine products per input pixel instead of sixteen) against float64 torch (the reference's own formula, stylegan2_layers.py:312-321)
    at the fp32-class bar of the other conv kernels, and against the four-phase kernels of the same plan; every epilogue option it
    takes (bias, noise, leaky ReLU, out_scale, statistics), ragged extents (not multiples of 15 / 16), Cout 128 / 256 / 512."""
    torch.manual_seed(11)
    prev = dict(ops.UP9)
    try:
        for name, B, ci, co, H, Wd, feat in [("convT 64->128 40x24 full", 2, 64, 128, 40, 24, "full"), ("convT 256->128 64x64 plain", 1, 256, 128, 64, 64, "plain"),
                                             ("convT 128->256 17x33 full (ragged)", 2, 128, 256, 17, 33, "full"), ("convT 32->512 31x30 scale", 1, 32, 512, 31, 30, "scale"),
                                             ("convT 512->256 128x128 full", 2, 512, 256, 128, 128, "full")]:
            w = g(torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9))
            x = g(torch.randn(B, H, Wd, ci))
            kw = {}
            if feat == "full":
                kw = dict(bias=g(torch.randn(co)), noise=g(torch.randn(B, 1, 2 * H, 2 * Wd)), noise_weight=0.3, act=ops.ACT_LRELU)
            elif feat == "scale":
                kw = dict(bias=g(torch.randn(co)), out_scale=0.7)
            outs = []
            for on in (False, True):
                ops.UP9.update(value=on, min_blocks=0, min_fill=0.0)
                plan = ops.ConvPlan(w, kind="convT", scale=0.5)
                assert (plan.choose_kernel(H, Wd, 2 * H, 2 * Wd, H, Wd, 2)[0] == "up9") == on
                y, st = plan(x, stats=True, **kw)
                outs.append((y.cpu(), st.sum(1).cpu()))
            report("nine-product upscale %s vs the four-phase form" % name, outs[1][0], outs[0][0], 1e-5)
            report("nine-product upscale %s stats" % name, outs[1][1], outs[0][1], 2e-5)
            if feat != "full":
                ref = conv_ref(nchw(x.cpu()).double(), (w.cpu().double() * 0.5), "convT", 0)
                if feat == "scale":
                    ref = (ref + kw["bias"].cpu().double().view(1, -1, 1, 1)) * 0.7
                report("nine-product upscale %s vs float64" % name, nchw(outs[1][0]), ref, 3e-5)
        # a call that carries an option the nine-product kernel does not take (normalise-on-load, residual, PReLU, non-zero padding)
        # FALLS BACK to the phase-pair / four-phase forms with UP9 left on (round-4 ADVICE: it used to raise) -- bit-identical to the
        # same call with UP9 off
        ops.UP9.update(value=True, min_blocks=0, min_fill=0.0)
        B, ci, co, H, Wd = 2, 64, 128, 30, 30
        w = g(torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9))
        x = g(torch.randn(B, H, Wd, ci))
        plan = ops.ConvPlan(w, kind="convT")
        assert plan.choose_kernel(H, Wd, 2 * H, 2 * Wd, H, Wd, 2)[0] == "up9"
        for tag, kw in (("residual", dict(residual=g(torch.randn(B, 2 * H, 2 * Wd, co)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)),
                        ("in_ss", dict(in_ss=g(torch.rand(B, ci, 2) + 0.5), in_act=ops.ACT_LRELU)),
                        ("prelu", dict(act=ops.ACT_PRELU, prelu=g(torch.tensor([0.25])))),
                        ("reflect pad", dict(pad_mode=ops.PAD_REFLECT))):
            y_on = plan(x, **kw)
            ops.UP9["value"] = False
            y_off = ops.ConvPlan(w, kind="convT")(x, **kw)
            ops.UP9["value"] = True
            ok = bool(torch.equal(y_on, y_off))
            RES.append(("upscale with %s and UP9 on falls back to the four-phase forms (bit-identical)" % tag, ok))
            print("UP9 fallback %-12s %s" % (tag, "ok" if ok else "FAIL"), flush=True)
    finally:
        ops.UP9.update(prev)


@direct_kernels_only
def t_conv_variants_single_pass():
    """The N-256 kernel in the single-pass modes (bf16, fp16) against the tile kernel in the same mode: bit-identical."""
    torch.manual_seed(8)
    nz_ = torch.randn
    cd = (ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS)
    for prec in (1, 3):
        for name, B, ci, co, H, Wd, kind, pm in (("3x3 reflect 64->256 33x47", 2, 64, 256, 33, 47, "conv", 1),
                                                  ("convT 64->512 20x12", 2, 64, 512, 20, 12, "convT", 0),
                                                  ("3x3 zero 256->256 256x256", 1, 256, 256, 256, 256, "conv", 0)):
            w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
            x = g(nz_(B, H, Wd, ci))
            outs = []
            for variant in (0, 2):
                ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = variant, 0
                plan = ops.ConvPlan(w, kind=kind, precision=prec)
                y, st = plan(x, pad_mode=pm, stats=True, bias=g(torch.arange(co, dtype=torch.float32) * 0.01), act=ops.ACT_LRELU)
                outs.append((y.cpu(), st.sum(1).cpu()))
            ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = cd
            ok = bool(torch.equal(outs[0][0], outs[1][0]))
            RES.append(("single-pass precision %d N=256 %s bit-identical" % (prec, name), ok))
            print("single-pass prec %d %-30s %s max diff %.3e" % (prec, name, "ok  " if ok else "FAIL", (outs[0][0] - outs[1][0]).abs().max().item()), flush=True)
            report("single-pass prec %d %s stats" % (prec, name), outs[1][1], outs[0][1], 1e-5)
        # the 32 x 16 px x 128 ch tile of the Cout = 128-class layers (variant 7, production form in the single-pass modes), fp32 and
        # half storage, against the 16-row tile kernel
        tt = dict(ops.TALL_TILE_SINGLE)
        for name, B, ci, co, H, Wd, kind, pm, feat in (("3x3 reflect 64->128 33x47", 2, 64, 128, 33, 47, "conv", 1, "plain"),
                                                        ("3x3 zero 128->128 70x40 in_ss + res", 1, 128, 128, 70, 40, "conv", 0, "inss"),
                                                        ("s2d 64->128 -> 37x20", 2, 64, 128, 37, 20, "s2d", 0, "plain")):
            w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
            for dt in (torch.float32, torch.float16 if prec == 3 else torch.bfloat16):
                outs = []
                for tall in (False, True):
                    ops.TALL_TILE_SINGLE.update(value=tall, min_blocks=0)
                    plan = ops.ConvPlan(w, kind=kind, precision=prec)
                    torch.manual_seed(17)
                    cx = plan.max_chan + 32
                    x = (g(nz_(B, H + 1, Wd + 1, cx)) if kind == "s2d" else g(nz_(B, H, Wd, cx))).to(dt)
                    kw = dict(out_hw=(H, Wd)) if kind == "s2d" else {}
                    if feat == "inss":
                        kw.update(in_ss=g(torch.rand(B, cx, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])),
                                  residual=g(nz_(B, H, Wd, co)).to(dt), out_scale=0.7)
                    assert (plan.choose_kernel(H, Wd, H, Wd, x.shape[1], x.shape[2], 1)[0] == 7) == tall
                    y, st = plan(x, pad_mode=pm, stats=True, bias=g(torch.arange(co, dtype=torch.float32) * 0.01), act=ops.ACT_LRELU, **kw)
                    outs.append((y.cpu(), st.sum(1).cpu()))
                ops.TALL_TILE_SINGLE.update(tt)
                ok = bool(torch.equal(outs[0][0], outs[1][0]))
                nm = "single-pass precision %d tall tile %s %s" % (prec, name, str(dt).split(".")[-1])
                RES.append((nm + " bit-identical", ok))
                print("%-80s %s max diff %.3e" % (nm, "ok  " if ok else "FAIL", (outs[0][0].float() - outs[1][0].float()).abs().max().item()), flush=True)
                report(nm + " stats", outs[1][1], outs[0][1], 1e-5)
        # the streaming kernels (1x1, thin stride-2, thin 3x3) in the same mode
        dmax, st1 = dict(ops.DIRECT_MAX), ops.STREAM_1X1["value"]
        for name, B, ci, co, H, Wd, kind, k, pm in (("1x1 128->64 40x48", 2, 128, 64, 40, 48, "conv", 1, 0),
                                                     ("3x3 reflect 32->32 33x47", 2, 32, 32, 33, 47, "conv", 3, 1),
                                                     ("3x3 zero 64->64 256x256", 1, 64, 64, 256, 256, "conv", 3, 0),
                                                     ("s2d 32->64 -> 20x23", 2, 32, 64, 20, 23, "s2d", 3, 0)):
            w = g(nz_(co, ci, k, k) / math.sqrt(ci * k * k))
            outs = []
            for stream in (False, True):
                ops.STREAM_1X1["value"] = stream
                ops.DIRECT_MAX["cout"] = dmax["cout"] if stream else 0
                plan = ops.ConvPlan(w, kind=kind, precision=prec)
                torch.manual_seed(13)
                x = g(nz_(B, H + 1, Wd + 1, plan.max_chan + 32)) if kind == "s2d" else g(nz_(B, H, Wd, plan.max_chan + 32))
                kw = dict(out_hw=(H, Wd)) if kind == "s2d" else {}
                y, st = plan(x, pad_mode=pm, stats=True, bias=g(torch.arange(co, dtype=torch.float32) * 0.01), act=ops.ACT_LRELU, **kw)
                outs.append((y.cpu(), st.sum(1).cpu()))
            ops.STREAM_1X1["value"] = st1
            ops.DIRECT_MAX.update(dmax)
            ok = bool(torch.equal(outs[0][0], outs[1][0]))
            RES.append(("single-pass precision %d stream/direct %s bit-identical" % (prec, name), ok))
            print("single-pass prec %d %-30s %s max diff %.3e" % (prec, name, "ok  " if ok else "FAIL", (outs[0][0] - outs[1][0]).abs().max().item()), flush=True)
            report("single-pass prec %d %s stats" % (prec, name), outs[1][1], outs[0][1], 1e-5)


def t_conv_k64():
    """64 input channels per step in the single-pass modes on half-stored activations (ppst_conv_args.k64: the chunk's second 32 channels
    where the fp32-class kernel keeps its lo planes, two MFMAs per product) against the 32-channel-step form of the same plan: the same
    products in another fp32 summation order, so outputs agree to a rounding of the stored type and statistics to 1e-5; every kernel
    geometry that takes it (N-256 plain / four phases / phase pairs, the 24-row tile of the Cout = 128 layers), with normalise-on-load,
    residual and noise; and the plain case against float64."""
    torch.manual_seed(23)
    nz_ = torch.randn
    prev = ops.K64["value"], ops.FAT_MIN_BLOCKS, dict(ops.TALL_TILE_SINGLE), dict(ops.DUAL_CONVT)
    ops.FAT_MIN_BLOCKS = 0
    ops.TALL_TILE_SINGLE.update(min_blocks=0)
    ops.DUAL_CONVT.update(min_blocks=0)
    pre = g(torch.tensor([0.25]))
    try:
        for prec, dt in ((3, torch.float16), (1, torch.bfloat16)):
            tag, tol = ("fp16", 2e-3) if prec == 3 else ("bf16", 1.6e-2)
            for name, B, ci, co, H, Wd, kind, pm, want in (("N-256 3x3 zero 128->256 64x64", 2, 128, 256, 64, 64, "conv", 0, 2),
                                                            ("N-256 3x3 reflect 256->256 33x47", 1, 256, 256, 33, 47, "conv", 1, 2),
                                                            ("N-256 convT 128->512 20x12", 2, 128, 512, 20, 12, "convT", 0, 2),
                                                            ("dual convT 64->128 64x48", 2, 64, 128, 64, 48, "convT", 0, "dual"),
                                                            ("24-row tile 3x3 reflect 64->128 50x47", 2, 64, 128, 50, 47, "conv", 1, 7),
                                                            ("24-row tile s2d 64->128 -> 37x20", 2, 64, 128, 37, 20, "s2d", 0, 7)):
                w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
                plan = ops.ConvPlan(w, kind=kind, precision=prec)
                cx = plan.max_chan + 32
                x = (g(nz_(B, H + 1, Wd + 1, cx)) if kind == "s2d" else g(nz_(B, H, Wd, cx))).to(dt)
                oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
                kw0 = dict(out_hw=(H, Wd)) if kind == "s2d" else {}
                assert plan.choose_kernel(H, Wd, oh, ow, x.shape[1], x.shape[2], 2 if kind == "convT" else 1)[0] == want, name
                bias = g(torch.arange(co, dtype=torch.float32) * 0.01)
                res = g(nz_(B, oh, ow, co)).to(dt); nzp = g(nz_(B, 1, oh, ow)); iss = g(torch.rand(B, cx, 2) + 0.5)
                for vname, kw in (("", dict(bias=bias, act=ops.ACT_LRELU)),
                                  ("in_ss prelu + res", dict(bias=bias, in_ss=iss, in_act=ops.ACT_PRELU, in_prelu=pre, residual=res, out_scale=0.7)),
                                  ("noise + res after act", dict(bias=bias, noise=nzp, noise_weight=0.3, act=ops.ACT_LRELU, residual=res, res_after_act=True))):
                    outs = []
                    for on in (False, True):
                        ops.K64["value"] = on
                        y, st = plan(x, pad_mode=pm, stats=True, **kw, **kw0)
                        outs.append((y.float().cpu(), st.sum(1).cpu()))
                    report("k64 %s %s %s" % (tag, name, vname), outs[1][0], outs[0][0], tol)
                    report("k64 %s %s %s stats" % (tag, name, vname), outs[1][1], outs[0][1], 1e-5)
                if kind == "conv" and prec == 3:
                    ops.K64["value"] = True
                    ref = conv_ref(nchw(x.float().cpu()).double(), w.cpu().double().half().double(), "conv", pm)      # operands as the kernel sees them
                    report("k64 %s %s vs float64 (fp16 operands)" % (tag, name), nchw(plan(x, pad_mode=pm).float()), ref, 2e-3)
    finally:
        ops.K64["value"], ops.FAT_MIN_BLOCKS = prev[0], prev[1]
        ops.TALL_TILE_SINGLE.update(prev[2]); ops.DUAL_CONVT.update(prev[3])


def _k64_off(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        prev = ops.K64["value"]
        ops.K64["value"] = False        # (bit-exactness against the fp32-storage form needs the same 32-channel step order)
        try:
            return fn(*a, **k)
        finally:
            ops.K64["value"] = prev
    return wrapped


@_k64_off
def t_half_storage():
    """Half-precision activation storage (ppst_conv_args.io_st and the `_st` entry points; include/ppst_hip.h): a kernel given
    IEEE-half (mode 3) / bfloat16 (mode 1) tensors computes in fp32 exactly as its fp32 form and rounds once, to nearest even, at
    its store -- so every launch must equal, BIT FOR BIT, the fp32-storage launch on the widened inputs, rounded by torch.
    Covers each kernel of the inference networks' paths: the four conv families in both single-pass modes (plain, normalise-on-load,
    residual before / after the activation, noise, the scattered output of the fused upscale), blur (plain / decimating / space-to-
    depth, normalise-on-load), affine_act (residual, residual affine, on-the-fly bilinear skip, fp32 output), FromRGB / ToRGB,
    GAP+GMP, SpatialCodeModulation, nearest upsample."""
    torch.manual_seed(21)
    nz_ = torch.randn

    def same(name, a, b):
        ok = a.dtype == b.dtype and bool(torch.equal(a, b))
        RES.append((name, ok))
        d = (a.float() - b.float()).abs().max().item() if a.shape == b.shape else float("nan")
        print("half storage %-58s %s max diff %.3e" % (name, "ok  " if ok else "FAIL", d), flush=True)

    for prec, dt in ((3, torch.float16), (1, torch.bfloat16)):
        tag = "fp16" if prec == 3 else "bf16"
        h = lambda t: g(t).to(dt)                       # a stored activation
        # ---- elementwise / blur / small convs
        x = h(nz_(2, 24, 20, 64)); r = h(nz_(2, 24, 20, 64)); rh = h(nz_(2, 12, 10, 64))
        ss = g(torch.rand(2, 64, 2) + 0.5); rss = g(torch.rand(2, 64, 2) + 0.5)
        pre = g(torch.tensor([0.25]))
        for name, kw in (("plain", {}), ("res", dict(res=r, out_scale=0.7)), ("res affine lrelu", dict(res=r, res_scale_shift=rss, act=ops.ACT_LRELU)),
                         ("res before prelu", dict(res=r, res_before_act=True, act=ops.ACT_PRELU, prelu=pre)),
                         ("res_up2", dict(res=rh, res_up2=True, out_scale=0.7))):
            kwf = {k: (v.float() if isinstance(v, torch.Tensor) and v.dtype == dt else v) for k, v in kw.items()}
            same("%s affine_act %s" % (tag, name), ops.affine_act(x, ss, **kw), ops.affine_act(x.float(), ss, **kwf).to(dt))
        x12 = h(nz_(2, 9, 7, 12)); r12 = h(nz_(2, 9, 7, 12)); ss12 = g(torch.rand(2, 12, 2) + 0.5)     # C % 8 != 0: the 4-channel form
        same("%s affine_act 12 channels" % tag, ops.affine_act(x12, ss12, res=r12, act=ops.ACT_LRELU),
             ops.affine_act(x12.float(), ss12, res=r12.float(), act=ops.ACT_LRELU).to(dt))
        same("%s affine_act -> fp32" % tag, ops.affine_act(x, ss, res=r, out_dtype=torch.float32), ops.affine_act(x.float(), ss, res=r.float()))
        same("%s affine_act fp32 -> half" % tag, ops.affine_act(x.float(), ss, out_dtype=dt), ops.affine_act(x.float(), ss).to(dt))
        k3 = g(torch.tensor([1., 2., 1.])); k3 = (k3[:, None] * k3[None, :] / 16).contiguous()
        k4 = g(torch.tensor([1., 3., 3., 1.])); k4 = (k4[:, None] * k4[None, :] / 64).contiguous()
        xb = h(nz_(2, 37, 41, 32)); bss = g(torch.rand(2, 32, 2) + 0.5)
        for name, kk, a_ in (("3x3 reflect s2d", k3, dict(pad0=1, pad1=1, pad_mode=ops.PAD_REFLECT, s2d=True)),
                             ("3x3 reflect s2d in_ss lrelu", k3, dict(pad0=1, pad1=1, pad_mode=ops.PAD_REFLECT, s2d=True, in_ss=bss, in_act=ops.ACT_LRELU)),
                             ("3x3 zero down 2", k3, dict(pad0=1, pad1=0, pad_mode=ops.PAD_ZERO, down=2)),
                             ("4x4 zero", k4, dict(pad0=2, pad1=1, pad_mode=ops.PAD_ZERO)),
                             ("4x4 zero s2d", k4, dict(pad0=2, pad1=2, pad_mode=ops.PAD_ZERO, s2d=True))):
            p0, p1 = a_.pop("pad0"), a_.pop("pad1")
            ya, hwa = ops.blur_nhwc(xb, kk, p0, p1, **a_)
            yb, hwb = ops.blur_nhwc(xb.float(), kk, p0, p1, **a_)
            same("%s blur %s" % (tag, name), ya, yb.to(dt))
        xb12 = h(nz_(2, 21, 18, 12))
        same("%s blur 3x3 reflect s2d 12 channels" % tag, ops.blur_nhwc(xb12, k3, 1, 1, pad_mode=ops.PAD_REFLECT, s2d=True)[0],
             ops.blur_nhwc(xb12.float(), k3, 1, 1, pad_mode=ops.PAD_REFLECT, s2d=True)[0].to(dt))
        img = g(nz_(2, 19, 23, 3)); w_in = g(nz_(32, 3, 1, 1)); b_in = g(nz_(32) * 0.1)
        same("%s FromRGB (small cin) -> half" % tag, ops.conv1x1_small_cin(img, w_in, b_in, 0.5, ops.ACT_LRELU, out_dtype=dt),
             ops.conv1x1_small_cin(img, w_in, b_in, 0.5, ops.ACT_LRELU).to(dt))
        xr = h(nz_(2, 19, 23, 64)); w_out = g(nz_(3, 64, 1, 1)); b_out = g(nz_(3) * 0.1)
        # (a 64-term dot product per output: the compiler contracts the two instantiations' multiply-adds differently -- last-bit
        #  differences of an fp32 output, not a storage rounding)
        report("half storage %s ToRGB (cout 3) from half" % tag, ops.conv1x1_small_cout(xr, w_out, b_out, 0.125),
               ops.conv1x1_small_cout(xr.float(), w_out, b_out, 0.125), 1e-6)
        same("%s gap_gmp" % tag, ops.gap_gmp(xr), ops.gap_gmp(xr.float()))
        sp = g(nz_(2, 9, 7, 256)); sc = g(nz_(2, 256)); sh = g(nz_(2, 256))
        same("%s spatial_modulation -> half" % tag, ops.spatial_modulation(sp, sc, sh, out_dtype=dt), ops.spatial_modulation(sp, sc, sh).to(dt))
        same("%s upsample_nearest2" % tag, ops.upsample_nearest2(xr), ops.upsample_nearest2(xr.float()).to(dt))
        # ---- the conv families of the mode
        cases = (("tile 3x3 reflect 64->128 33x47", 2, 64, 128, 33, 47, "conv", 3, 1),
                 ("tile convT 64->128 20x12", 2, 64, 128, 20, 12, "convT", 3, 0),
                 ("dual convT 64->128 64x48", 2, 64, 128, 64, 48, "convT", 3, 0),
                 ("N-256 3x3 zero 128->256 64x64", 2, 128, 256, 64, 64, "conv", 3, 0),
                 ("N-256 convT 64->512 20x12", 2, 64, 512, 20, 12, "convT", 3, 0),
                 ("stream 1x1 128->64 40x48", 2, 128, 64, 40, 48, "conv", 1, 0),
                 ("stream 1x1 64->256 17x19", 2, 64, 256, 17, 19, "conv", 1, 0),
                 ("direct 3x3 reflect 32->32 33x47", 2, 32, 32, 33, 47, "conv", 3, 1),
                 ("direct 3x3 zero 64->64 40x40", 1, 64, 64, 40, 40, "conv", 3, 0),
                 ("direct s2d 32->64 -> 20x23", 2, 32, 64, 20, 23, "s2d", 3, 0))
        fm = ops.FAT_MIN_BLOCKS
        ops.FAT_MIN_BLOCKS = 0
        try:
            for name, B, ci, co, H, Wd, kind, k, pm in cases:
                w = g(nz_(co, ci, k, k) / math.sqrt(ci * k * k))
                plan = ops.ConvPlan(w, kind=kind, precision=prec)
                cx = plan.max_chan + 32
                x = h(nz_(B, H + 1, Wd + 1, cx)) if kind == "s2d" else h(nz_(B, H, Wd, cx))
                oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
                kw0 = dict(out_hw=(H, Wd)) if kind == "s2d" else {}
                bias = g(torch.arange(co, dtype=torch.float32) * 0.01)
                res = h(nz_(B, oh, ow, co)); nzp = g(nz_(B, 1, oh, ow)); iss = g(torch.rand(B, cx, 2) + 0.5)
                for vname, kw in (("", dict(bias=bias, act=ops.ACT_LRELU)),
                                  ("in_ss prelu + res", dict(bias=bias, in_ss=iss, in_act=ops.ACT_PRELU, in_prelu=pre, residual=res, out_scale=0.7)),
                                  ("noise + res after act", dict(bias=bias, noise=nzp, noise_weight=0.3, act=ops.ACT_LRELU, residual=res, res_after_act=True))):
                    kwf = dict(kw)
                    if "residual" in kwf:
                        kwf["residual"] = kwf["residual"].float()
                    ya, sa = plan(x, pad_mode=pm, stats=True, **kw, **kw0)
                    yb, sb = plan(x.float(), pad_mode=pm, stats=True, **kwf, **kw0)
                    same("%s conv %s %s" % (tag, name, vname), ya, yb.to(dt))
                    same("%s conv %s %s stats" % (tag, name, vname), sa, sb)
        finally:
            ops.FAT_MIN_BLOCKS = fm


@direct_kernels_only
def t_conv1x1_stream():
    """conv1x1.hip (streaming 1x1 kernel and its direct form for thin 3x3 / stride-2 layers; operands swapped) against
    conv_mfma.hip on the same plans: bit-identical outputs, statistics to rounding; every epilogue / normalise-on-load
    option, the three padding modes, ragged sizes, Cout not a multiple of 64 (and <= 32: the 2-tile build), > 8 steps."""
    torch.manual_seed(6)
    nz_ = torch.randn
    cases = [
        # name, B, Cin, Cout, H, W, kind, k, pad_mode, features
        ("1x1 64->64 48x40 bias+noise+lrelu", 2, 64, 64, 48, 40, "conv", 1, 0, "full"),
        ("1x1 32->384 33x47 plain", 2, 32, 384, 33, 47, "conv", 1, 0, "plain"),
        ("1x1 512->256 16x16 in_ss+prelu (16 steps)", 1, 512, 256, 16, 16, "conv", 1, 0, "inss"),
        ("1x1 128->64 24x24 residual", 3, 128, 64, 24, 24, "conv", 1, 0, "res"),
        ("1x1 256->36 17x9 bias", 1, 256, 36, 17, 9, "conv", 1, 0, "bias"),
        ("1x1 dgrad (64<-128) 32x32", 2, 64, 128, 32, 32, "dgrad", 1, 0, "plain"),
        ("3x3 zero 32->32 48x40 bias+noise+lrelu", 2, 32, 32, 48, 40, "conv", 3, 0, "full"),
        ("3x3 reflect 64->64 33x47 plain", 2, 64, 64, 33, 47, "conv", 3, 1, "plain"),
        ("3x3 replicate 32->48 19x21 in_ss+prelu", 1, 32, 48, 19, 21, "conv", 3, 2, "inss"),
        ("3x3 zero 128->64 24x24 residual (36 steps)", 2, 128, 64, 24, 24, "conv", 3, 0, "res"),
        ("3x3 zero in_ss 64->20 17x9", 1, 64, 20, 17, 9, "conv", 3, 0, "inss"),
        ("s2d 32->64 -> 20x23 residual", 2, 32, 64, 20, 23, "s2d", 3, 0, "res"),
        ("dgrad 3x3 (32<-64) 32x32", 2, 32, 64, 32, 32, "dgrad", 3, 0, "plain"),
        # 65..128 output channels: the 32 px x 128 ch form of the 3x3 kernel (blobs packed for bn = 128)
        ("3x3 zero 128->128 48x40 bias+noise+lrelu", 2, 128, 128, 48, 40, "conv", 3, 0, "full"),
        ("3x3 reflect 64->96 33x47 in_ss+prelu", 2, 64, 96, 33, 47, "conv", 3, 1, "inss"),
        ("3x3 replicate 256->128 19x21 residual (72 steps)", 1, 256, 128, 19, 21, "conv", 3, 2, "res"),
        ("dgrad 3x3 (128<-128) 32x32", 2, 128, 128, 32, 32, "dgrad", 3, 0, "plain"),
        ("3x3 zero 128->128 512x512 bias+noise+lrelu", 2, 128, 128, 512, 512, "conv", 3, 0, "full"),
        # full-size launches: > 512 tiles, so the single-chunk 3x3 kernel runs its persistent loop with resident weights
        ("3x3 reflect 32->32 512x512 bias+noise+lrelu (persistent)", 2, 32, 32, 512, 512, "conv", 3, 1, "full"),
        ("3x3 zero 64->64 400x512 in_ss (2 chunks)", 2, 64, 64, 400, 512, "conv", 3, 0, "inss"),
        ("1x1 128->64 512x512 residual", 2, 128, 64, 512, 512, "conv", 1, 0, "res"),
        ("s2d 32->64 -> 256x256", 2, 32, 64, 256, 256, "s2d", 3, 0, "plain"),
    ]
    dmax = dict(ops.DIRECT_MAX)
    ksp = dict(ops.KSPLIT_128)
    ops.KSPLIT_128["value"] = False      # the reference side of these bit-identity checks is the tile kernel (K-split is not bit-identical)
    for name, B, ci, co, H, Wd, kind, k, pm, feat in cases:
        w = g(nz_(co, ci, k, k) / math.sqrt(ci * k * k))
        outs = {}
        for stream in (False, True):
            ops.STREAM_1X1["value"] = stream
            ops.DIRECT_MAX["cout"] = dmax["cout"] if stream else 0
            ops.DIRECT_MAX["cout3x3"] = 128 if stream else 0      # (off by default: measured slower; kept bit-identical)
            plan = ops.ConvPlan(w, kind=kind)
            cin_eff = plan.max_chan + 32
            torch.manual_seed(12)
            x = g(nz_(B, H + 1, Wd + 1, cin_eff)) if kind == "s2d" else g(nz_(B, H, Wd, cin_eff))
            kw = dict(out_hw=(H, Wd)) if kind == "s2d" else {}
            if feat == "full":
                kw.update(bias=g(nz_(plan.cout)), noise=g(nz_(B, 1, H, Wd)), noise_weight=0.3, act=ops.ACT_LRELU)
            elif feat == "bias":
                kw.update(bias=g(nz_(plan.cout)))
            elif feat == "inss":
                kw.update(in_ss=g(torch.rand(B, cin_eff, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])),
                          act=ops.ACT_PRELU, prelu=g(torch.tensor([0.1])))
            elif feat == "res":
                kw.update(residual=g(nz_(B, H, Wd, plan.cout)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
            y, st = plan(x, stats=True, pad_mode=pm, **kw)
            outs[stream] = (y.cpu(), st.sum(1).cpu())
        ops.STREAM_1X1["value"] = STREAM_DEFAULT
        ops.DIRECT_MAX.update(dmax)
        (y0, s0), (y1, s1) = outs[False], outs[True]
        RES.append(("stream/direct %s bit-identical" % name, bool(torch.equal(y0, y1))))
        print("stream/direct %-46s %s max diff %.3e" % (name, "ok  " if torch.equal(y0, y1) else "FAIL", (y0 - y1).abs().max().item()), flush=True)
        report("stream/direct %s stats" % name, s1, s0, 1e-5)
    ops.KSPLIT_128.update(ksp)


STREAM_DEFAULT = ops.STREAM_1X1["value"]


def t_norm_pool():
    torch.manual_seed(3)
    for (B, C, H, Wd) in [(2, 64, 40, 40), (1, 3, 64, 64), (2, 512, 16, 16), (1, 128, 70, 30)]:
        x = torch.randn(B, C, H, Wd) * 2 + 0.5
        st = ops.in_stats(g(nhwc(x)))
        ss = ops.in_finalize(st, H * Wd)
        report("in_stats+affine C%d %dx%d" % (C, H, Wd), nchw(ops.affine_act(g(nhwc(x)), ss).cpu()), O.instance_norm(x.double()), 1e-5)
    x = torch.randn(2, 64, 20, 24)
    st = ops.in_stats(g(nhwc(x)), rep_pad=True)
    ss = ops.in_finalize(st, 22 * 26)
    ref = O.instance_norm(F.pad(x.double(), (1, 1, 1, 1), mode="replicate"))[:, :, 1:-1, 1:-1]
    report("in_stats rep_pad (head quirk)", nchw(ops.affine_act(g(nhwc(x)), ss).cpu()), ref, 1e-5)
    style = torch.randn(2, 128); pb = torch.randn(64)
    ss = ops.in_finalize(ops.in_stats(g(nhwc(x))), 20 * 24, style=g(style), post_bias=g(pb))
    s = style.view(2, 2, 64, 1, 1)
    ref = O.instance_norm(x.double()) * (s[:, 0] + 1) + s[:, 1] + pb.view(1, -1, 1, 1)
    report("in_finalize style+post_bias", nchw(ops.affine_act(g(nhwc(x)), ss).cpu()), ref, 1e-5)
    res = torch.randn(2, 64, 20, 24)
    rss = torch.randn(2, 64, 2)
    y = ops.affine_act(g(nhwc(x)), ss, res=g(nhwc(res)), res_scale_shift=g(rss), act=ops.ACT_LRELU, out_scale=0.7)
    ref2 = (O.fused_leaky_relu(ref, None) + (res * rss[:, :, 0, None, None] + rss[:, :, 1, None, None])) * 0.7
    report("affine_act lrelu + affine residual", nchw(y.cpu()), ref2, 1e-5)
    y2, part = ops.affine_act_stats(g(nhwc(x)), ss, res=g(nhwc(res)), res_scale_shift=g(rss), act=ops.ACT_LRELU, out_scale=0.7, rep_pad=True)
    report("affine_act_stats output", nchw(y2.cpu()), ref2, 1e-5)
    refn = O.instance_norm(F.pad(ref2, (1, 1, 1, 1), mode="replicate"))[:, :, 1:-1, 1:-1]
    report("affine_act_stats partials (rep_pad IN)", nchw(ops.affine_act(y2, ops.in_finalize(part, 22 * 26)).cpu()), refn, 1e-5)
    lo = torch.randn(2, 64, 10, 12)
    up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False)
    y3 = ops.affine_act(g(nhwc(x)), ss, res=g(nhwc(lo)), out_scale=0.7, res_up2=True)
    report("affine_act + on-the-fly x2 bilinear residual", nchw(y3.cpu()), (ref + up) * 0.7, 1e-5)
    y4, _ = ops.affine_act_stats(g(nhwc(x)), ss, res=g(nhwc(lo)), out_scale=0.7, res_up2=True)
    report("affine_act_stats + x2 bilinear residual", nchw(y4.cpu()), (ref + up) * 0.7, 1e-5)
    gg = ops.gap_gmp(g(nhwc(x)))
    report("gap_gmp", gg, torch.cat([x.mean((2, 3)), x.amax((2, 3))], 1), 2e-6)
    mask = (torch.rand(2, 20, 24) > 0.5).float()
    xm = x * mask[:, None]
    report("gap_gmp masked", ops.gap_gmp(g(nhwc(x)), g(mask)), torch.cat([xm.mean((2, 3)), xm.amax((2, 3))], 1), 2e-6)
    x = torch.randn(2, 32, 64, 64)
    for f in (1, 2, 4, 8):
        report("avgpool f%d" % f, nchw(ops.avgpool(g(nhwc(x)), f).cpu()), F.adaptive_avg_pool2d(x, 64 // f), 2e-6)
    for (oh, ow) in [(128, 128), (256, 256), (64, 64), (32, 32), (512, 512)]:
        report("bilinear 64->%d" % oh, nchw(ops.bilinear(g(nhwc(x)), oh, ow).cpu()), F.interpolate(x, (oh, ow), mode="bilinear"), 2e-6)
    report("bilinear scale_factor 8 (E2.warp)", nchw(ops.bilinear(g(nhwc(x[:, :, :16, :16])), 128, 128).cpu()), F.interpolate(x[:, :, :16, :16], scale_factor=8, mode="bilinear"), 2e-6)
    dst = torch.zeros(2, 32, 32, 96, device=dev)
    ops.avgpool(g(nhwc(x)), 2, out=dst[..., 32:64])
    report("avgpool into slice", nchw(dst[..., 32:64].contiguous().cpu()), F.adaptive_avg_pool2d(x, 32), 2e-6)
    m = (torch.rand(2, 3, 32, 32) > 0.6).float()
    report("maxpool2 mask", nchw(ops.maxpool2(g(nhwc(m))).cpu()), F.max_pool2d(m, 2, 2), 0)
    report("upsample_nearest2", nchw(ops.upsample_nearest2(g(nhwc(x))).cpu()), x.repeat_interleave(2, 2).repeat_interleave(2, 3), 0)


def t_corr():
    torch.manual_seed(4)
    fea = torch.randn(2, 64, 64, 48)
    report("rselfcorr", nchw(ops.rselfcorr(g(nhwc(fea))).cpu()), O.rselfcorr(fea.double()), 1e-5)
    f = torch.randn(2, 512, 16, 16)
    fr = f.reshape(2, 512, -1)
    h1 = fr[:, :256] - fr[:, :256].mean(1, keepdim=True)
    ref = torch.cat((h1, fr[:, 256:]), 1)
    ref = (ref / (ref.norm(2, 1, keepdim=True) + O.EPS64)).permute(0, 2, 1)
    report("corr_prep", ops.corr_prep(g(nhwc(f)).reshape(2, 256, 512), 256), ref, 2e-6)
    for (b, M, N, K) in [(2, 256, 256, 512), (1, 4096, 4096, 512), (1, 130, 70, 64)]:
        A = torch.randn(b, M, K); Bm = torch.randn(b, N, K)
        ref = 0.5 * torch.matmul(A.double(), Bm.double().transpose(1, 2))
        for mode, tol in (("f32", 2e-6), ("x6", 2e-6), ("x3", 3e-5)):
            report("gemm_nt %s %dx%dx%d" % (mode, M, N, K), ops.gemm_nt(g(A), g(Bm), 0.5, mode=mode), ref, tol)
    for (b, M, N, K) in [(2, 256, 480, 256), (1, 4096, 480, 4096), (1, 200, 192, 64), (1, 128, 32, 128)]:
        A = torch.randn(b, M, K); Bm = torch.randn(b, K, N)
        ref = torch.matmul(A.double(), Bm.double())
        for mode, tol in (("f32", 2e-6 if K <= 256 else 1e-5), ("x6", 2e-6 if K <= 256 else 1e-5), ("x3", 3e-5)):
            report("gemm_nn %s %dx%dx%d" % (mode, M, N, K), ops.gemm_nn(g(A), g(Bm), mode=mode), ref, tol)
    x = torch.randn(300, 4096) * 0.3
    report("softmax rows /0.01", ops.softmax_rows_(g(x).clone(), 0.01), F.softmax(x.double() / 0.01, -1), 2e-5)
    fa = torch.randn(1, 512, 64, 64); fb = torch.randn(1, 512, 64, 64)
    from ppst_amd.ppst_model import PPSTModel
    img = torch.randn(2, 3, 64, 64)
    corr = F.softmax(torch.randn(2, 64, 64) * 3, -1)
    pt = ops.unfold_patches(g(img), 8)
    report("unfold_patches", pt, F.unfold(img, 8, stride=8).permute(0, 2, 1), 0)
    report("fold_patches", ops.fold_patches(pt, 3, 64, 64, 8), img, 0)
    wf = ops.fold_patches(ops.gemm_nn(g(corr), pt), 3, 64, 64, 8)
    report("model warp (unfold, gemm, fold)", wf, O.model_warp(img.double(), corr.double()), 2e-6)


def t_guided():
    rng = np.random.default_rng(5)
    H = Wd = 96
    yy, xx = np.meshgrid(np.arange(H), np.arange(Wd), indexing="ij")
    guide = np.stack([(128 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0)) for c in range(3)], -1)
    guide = np.clip(guide + rng.normal(0, 12, guide.shape), 0, 255).astype(np.uint8)
    src = np.clip(guide.astype(np.float64) * 0.6 + 50 + rng.normal(0, 25, guide.shape), 0, 255).astype(np.uint8)
    gb = torch.from_numpy(np.stack([guide, guide[::-1].copy()])).to(dev)
    sb = torch.from_numpy(np.stack([src, src[::-1].copy()])).to(dev)
    out, u8 = ops.guided_filter(gb, sb, 30, (0.02 * 255) ** 2, want_u8=True)
    ref = np.stack([O.guided_filter_color(guide, src, 30), O.guided_filter_color(guide[::-1].copy(), src[::-1].copy(), 30)])
    d = np.abs(u8.cpu().numpy().astype(int) - ref.astype(int))
    print("guided_filter u8: max |diff| %d, frac != %.4f" % (d.max(), (d > 0).mean()), flush=True)
    RES.append(("guided filter u8 within 1 LSB", bool(d.max() <= 1 and (d > 0).mean() < 0.02)))
    reff = (torch.from_numpy(ref).permute(0, 3, 1, 2).float() / 255.0 - 0.5) * 2
    # one uint8 step is 2/255 in [-1,1]; relative to refmax ~0.6
    report("guided_filter fp32 out (vs oracle, 1 LSB)", out, reff, (2.0 / 255 + 1e-6) / reff.abs().max().item())
    # a ragged extent: two column chunks of the H passes (the second 18 wide), W % 4 != 0, a last row segment of 24 rows
    H2, W2 = 600, 530
    yy, xx = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    guide2 = np.stack([(128 + 100 * np.sin(xx / 19.0 + c) * np.cos(yy / 13.0)) for c in range(3)], -1)
    guide2 = np.clip(guide2 + rng.normal(0, 12, guide2.shape), 0, 255).astype(np.uint8)
    src2 = np.clip(guide2.astype(np.float64) * 0.6 + 50 + rng.normal(0, 25, guide2.shape), 0, 255).astype(np.uint8)
    _, u82 = ops.guided_filter(torch.from_numpy(guide2[None]).to(dev), torch.from_numpy(src2[None]).to(dev), 30, (0.02 * 255) ** 2, want_u8=True)
    d2 = np.abs(u82.cpu().numpy().astype(int)[0] - O.guided_filter_color(guide2, src2, 30).astype(int))
    print("guided_filter u8 600x530: max |diff| %d, frac != %.4f" % (d2.max(), (d2 > 0).mean()), flush=True)
    RES.append(("guided filter u8 600x530 within 1 LSB", bool(d2.max() <= 1 and (d2 > 0).mean() < 0.02)))


# ----------------------------------------------------------------- nets ----
def t_networks():
    from ppst_amd.ppst_model import create_model
    sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
    noise = W.make_noise(3, 1)
    imgs = W.synthetic_images(5, 2)
    t0 = time.time()
    with torch.no_grad():
        r = oracle_swap512()
        d_ref = O.discriminator(sd, imgs)
    print("oracle swap %.1fs" % (time.time() - t0), flush=True)
    m = create_model(state_dict=sd, with_D=True)
    m.noise = {k: v.to(dev) for k, v in noise.items()}
    c, s = g(imgs[0:1]), g(imgs[1:2])
    with torch.no_grad():
        sp, gl = m(c, command="encode")
        report("E1 sp", sp, r["sp"], 1e-3)
        for i in range(4):
            report("E2 gl[%d]" % i, gl[i], r["gl"][i], 1e-3)
        fc, fc1 = m(c, command="extract_feat_from_image")
        fs, fs1 = m(s, command="extract_feat_from_image")
        report("G feat (content)", fc, r["fea_c"][:, :256], 1e-3)
        rc = m(fc1, command="Rselfcorr")
        rs = m(fs1, command="Rselfcorr")
        report("Rselfcorr (content)", rc, r["fea_c"][:, 256:], 1e-3)
        fcc = torch.cat((fc, rc), 1)
        fss = torch.cat((fs, rs), 1)
        report("fea_s cat", fss, r["fea_s"], 1e-3)
        corr = m(fss, fcc, command="corrm")
        agree = (corr[0].argmax(-1).cpu() == r["corr"][0].argmax(-1)).float().mean().item()
        print("corr argmax agreement %.5f ; max|dcorr| %.3e" % (agree, (corr.cpu() - r["corr"]).abs().max().item()), flush=True)
        RES.append(("corr argmax", agree > 0.995))
        _, glw = m(s, corr, command="encode2")
        for i in range(4):
            report("E2 gl_w[%d]" % i, glw[i], r["gl_w"][i], 1e-3)
        # E2 warp alone on the oracle's corr (isolates the GEMM from upstream noise)
        _, glw2 = m(s, g(r["corr"]), command="encode2")
        for i in range(4):
            report("E2 gl_w[%d] (oracle corr)" % i, glw2[i], r["gl_w"][i], 1e-3)
        out = m(sp, glw, target=None, command="decode")
        report("decode out", out, r["out"], 1e-3)
        out2 = m(g(r["sp"]), [g(t) for t in r["gl_w"]], target=None, command="decode")
        report("decode out (oracle inputs)", out2, r["out"], 1e-3)
        u8 = O.to_pil_uint8(out2[0].cpu()); u8r = O.to_pil_uint8(r["out"][0])
        print("uint8 image: frac pixels differing %.5f, max diff %d" % ((u8 != u8r).mean(), np.abs(u8.astype(int) - u8r.astype(int)).max()), flush=True)
        report("D(x)", m.discriminate(g(imgs)), d_ref, 1e-3)
        wimg = m(c, g(r["corr"]), command="warp")
        report("model.warp image", wimg, O.model_warp(imgs[0:1], r["corr"]), 1e-4)
        sm = m(g(r["sp"]), [g(t) for t in r["gl_w"]], target=c, command="decode")
        smr = O.smooth(r["out"], imgs[0:1])
        d = ((sm.cpu() - smr).abs() * 127.5).round()
        print("decode+guided filter: max LSB diff %d, frac != %.4f" % (d.max().item(), (d > 0).float().mean().item()), flush=True)
        # <= 1 uint8 LSB on a small fraction of pixels (fp32 rounding across a quantisation step of G's output);
        # the oracle's filter itself is parity-unpinned vs OpenCV
        RES.append(("decode + guided filter 512 within 1 LSB", bool(d.max().item() <= 1 and (d > 0).float().mean().item() < 0.02)))
    # cfg1: 256^2 encode/decode only, init-like weights
    sd0 = W.make_state_dict(0)
    im = W.synthetic_images(0, 2, size=256, smooth=False)
    with torch.no_grad():
        spr = O.encoder_con(sd0, im[0:1]); glr = O.encoder_col(sd0, im[1:2])[0]; outr = O.generator(sd0, spr, glr)
        m0 = create_model(state_dict=sd0)
        sp0, _ = m0(g(im[0:1]), command="encode")
        _, gl0 = m0(g(im[1:2]), command="encode")
        report("cfg1 256^2 sp", sp0, spr, 1e-3)
        report("cfg1 256^2 decode", m0(sp0, gl0, command="decode"), outr, 1e-3)


def t_configs():
    """BASELINE configs 3 and 5 and the train-only E2 mask heads."""
    from ppst_amd.ppst_model import create_model
    from ppst_amd.evaluation import swapping_grid, shard_pairs, simple_swap
    from ppst_amd import glue
    sd = W.make_state_dict(1, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    m = create_model(state_dict=sd)
    # --- E2 with mask + corrmatrix (encoder_col.py:171-245), B=2 so swap(mask) is defined
    imgs = W.synthetic_images(7, 2)
    torch.manual_seed(0)
    labels = torch.randint(0, 3, (2, 512, 512))
    mask = O.one_hot_mask(labels)
    report("one_hot mask exact", glue.one_hot_mask(g(labels)).cpu(), mask, 0)
    corr = F.softmax(torch.randn(2, 4096, 4096) * 4, -1)
    with torch.no_grad():
        v, pm, vw, pmw = O.encoder_col(sd, imgs, mask=mask, corrmatrix=corr)
        hv, hpm, hvw, hpmw = m.E2(g(imgs), mask=g(mask), corrmatrix=g(corr))
    for name, a, b in (("E2 mask vectors", hv, v), ("E2 mask projections_m", hpm, pm), ("E2 mask vectors_w", hvw, vw), ("E2 mask projections_mw", hpmw, pmw)):
        assert len(a) == len(b), name
        worst = max(rel(x, y)[0] for x, y in zip(a, b))
        RES.append((name, worst <= 1e-3)); print("%-46s %s worst rel %.3e over %d tensors" % (name, "ok  " if worst <= 1e-3 else "FAIL", worst, len(a)), flush=True)
    # --- config 3: content x style grid (2 x 2 here), guided filter on, sharded over 2 'ranks'
    m.noise = {k: v.to(dev) for k, v in W.make_noise(3, 1).items()}
    cs, ss_ = W.synthetic_images(11, 2), W.synthetic_images(12, 2)
    orc = O.PPSTOracle(sd, noise=W.make_noise(3, 1))
    with torch.no_grad():
        ref = {}
        for (i, j) in [(0, 1)]:                  # one pair against the CPU oracle; all four go through the sharding checks
            r = orc.simple_swap(cs[i:i + 1], ss_[j:j + 1], alpha=1.0)
            ref[(i, j)] = (r["out"], O.smooth(r["out"], cs[i:i + 1]))
        # the image passes and the pair passes of two simulated ranks (phase 1 -> exchange -> phase 3), then the same grid
        # on one rank: the sharded run must reproduce it bit for bit (kernels are batch-composition invariant)
        from ppst_amd.evaluation import grid_exchange, grid_image_pass, grid_pair_pass
        m.noise = {k: v.to(dev) for k, v in W.make_noise(3, 1).items()}      # one row: applies to every batch row
        outs = [grid_image_pass(m, g(cs), g(ss_), r, 2) for r in range(2)]
        tc, ts = grid_exchange(None, None, 2, 2, 2, gathered=outs)
        got = {}
        for rank in range(2):
            got.update(grid_pair_pass(m, g(cs), g(ss_), tc, ts, rank, 2, smooth=True))
        one = swapping_grid(m, g(cs), g(ss_), rank=0, world=1, smooth=True)
        same = all(torch.equal(one[k_], got[k_]) for k_ in one)
        RES.append(("grid: 2 simulated ranks == 1 rank bit for bit", bool(same and sorted(one) == sorted(got))))
        print("grid: 2 simulated ranks vs 1 rank: %s (max diff %.3e)" % ("identical" if same else "DIFFERENT",
              max((one[k_] - got[k_]).abs().max().item() for k_ in one)), flush=True)
        assert sorted(got) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    for key, (raw, sm) in ref.items():
        d = ((got[key].cpu() - sm[0]).abs() * 127.5).round()
        ok = d.max().item() <= 2 and (d > 0).float().mean().item() < 0.02
        RES.append(("grid pair %s guided-filtered" % (key,), ok))
        print("grid pair %s: guided-filter output max LSB diff %d, frac != %.4f %s" % (key, d.max().item(), (d > 0).float().mean().item(), "ok" if ok else "FAIL"), flush=True)
    # --- config 5: 1024^2 encode/decode + guided filter (fp32-class path and the single-pass bf16 path)
    im = W.synthetic_images(13, 2, size=1024)
    sd0 = W.make_state_dict(0, with_D=False, with_nce=False)
    with torch.no_grad():
        spr = O.encoder_con(sd0, im[0:1]); glr = O.encoder_col(sd0, im[1:2])[0]; outr = O.generator(sd0, spr, glr)
        m0 = create_model(state_dict=sd0)
        sp0, _ = m0(g(im[0:1]), command="encode"); _, gl0 = m0(g(im[1:2]), command="encode")
        out0 = m0(sp0, gl0, command="decode")
        report("cfg5 1024^2 decode (bf16x3)", out0, outr, 1e-3)
        smr = O.smooth(outr, im[0:1])
        sm0 = m0(sp0, gl0, target=g(im[0:1]), command="decode")
        d = ((sm0.cpu() - smr).abs() * 127.5).round()
        print("cfg5 1024^2 guided filter: max LSB diff %d, frac != %.4f" % (d.max().item(), (d > 0).float().mean().item()), flush=True)
        RES.append(("cfg5 guided filter", d.max().item() <= 2))
        ops.set_precision(1)
        m1 = create_model(state_dict=sd0)
        sp1, _ = m1(g(im[0:1]), command="encode"); _, gl1 = m1(g(im[1:2]), command="encode")
        out1 = m1(sp1, gl1, command="decode")
        ops.set_precision(0)
        report("cfg5 1024^2 decode (single-pass bf16, max-norm tol 1e-1)", out1, outr, 1e-1)
        rms = ((out1.cpu() - outr).pow(2).mean().sqrt() / outr.pow(2).mean().sqrt()).item()
        # sanity bound of the reduced-precision mode (2^-9 per operand over ~40 layers), not a parity claim
        print("cfg5 single-pass bf16 relative RMS error %.3e" % rms, flush=True)
        RES.append(("cfg5 bf16 rms", rms < 5e-2))


def t_precision():
    """Reduced-precision modes of BASELINE configs[3]/[4] against the fp32 oracle.  Tolerances stated BEFORE measuring
    (tests/test_gpu_parity.py:test_reduced_precision_modes docstring): relative RMS and max-norm of the output image."""
    from ppst_amd.ppst_model import create_model
    from ppst_amd.evaluation import simple_swap
    # (tag, rms 1024 enc/dec, max, rms 512 swap, max).  fp16x2 = the two-pass experiment: it must pass the fp32 gate itself
    # (1e-3 max-norm on the generator output, BASELINE north_star) to count as an fp32-class mode
    # MEASURED (round 2): fp16x2 is 2.0-2.5e-4 per layer (the fp16 rounding of the weights) and 2.7e-3 RMS / 3.1e-3 max-norm
    # over the whole recipe -- three times OUTSIDE the fp32 gate, for +8 % swaps/s.  It therefore is not an fp32-class
    # mode; the bars below only keep it from regressing (the first-stated bars 5e-4 / 1e-3 are the gate it failed).
    bars = {3: ("fp16", 5e-3, 3e-2, 2e-2, 1e-1), 1: ("bf16", 5e-2, 2e-1, 1e-1, 5e-1), 4: ("fp16x2", 5e-3, 1e-2, 5e-3, 1e-2)}
    if not ops.EXPERIMENTS:
        del bars[4]                      # the two-pass fp16 experiment is compiled only with PPST_EXPERIMENTS=1
    sd0 = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    im = W.synthetic_images(13, 2, size=1024)
    nz1024 = W.make_noise(5, 1, S=128)
    with torch.no_grad():
        ref1024 = O.generator(sd0, O.encoder_con(sd0, im[0:1]), O.encoder_col(sd0, im[1:2])[0], noise=nz1024)
    sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
    imgs = W.synthetic_images(5, 2)
    nz = W.make_noise(3, 1)
    ref512 = oracle_swap512()["out"]

    def errs(a, b):
        a, b = a.double().cpu(), b.double()
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item(), ((a - b).abs().max() / b.abs().max()).item()
    for prec, (tag, r1, m1, r2, m2) in bars.items():
        ops.set_precision(prec)
        try:
            with torch.no_grad():
                m = create_model(state_dict=sd0)
                m.noise = {k: v.to(dev) for k, v in nz1024.items()}
                sp, _ = m(g(im[0:1]), command="encode"); _, gl = m(g(im[1:2]), command="encode")
                rms, mx = errs(m(sp, gl, command="decode"), ref1024)
                ok = rms <= r1 and mx <= m1
                RES.append(("cfg4 1024^2 encode/decode %s" % tag, ok))
                print("cfg4 1024^2 encode/decode %-5s %s rel RMS %.3e (bar %.0e) max-norm %.3e (bar %.0e)" % (tag, "ok  " if ok else "FAIL", rms, r1, mx, m1), flush=True)
                if prec == 3:
                    # configs[4] as ONE workload: fp16 generator + guided filter (decode with target) against the fp32 oracle's
                    # smoothed image.  Bars stated before the first measurement: the filter is linear in its source for a fixed
                    # guide, so the fp16 decode error (bars above: 5e-3 RMS = 0.6 LSB, 3e-2 max = 3.8 LSB of the 8-bit image)
                    # passes through un-amplified, plus one LSB of quantisation: max <= 6 LSB, mean <= 1 LSB, < 5 % of the
                    # pixels off by more than 1 LSB.  (The oracle's filter itself is parity-unpinned: OpenCV-contrib is absent.)
                    sm = m(sp, gl, target=g(im[0:1]), command="decode")
                    d = ((sm.cpu() - O.smooth(ref1024, im[0:1])).abs() * 127.5).round()
                    ok = d.max().item() <= 6 and d.mean().item() <= 1.0 and (d > 1).float().mean().item() < 0.05
                    RES.append(("cfg4 1024^2 fp16 decode + guided filter", ok))
                    print("cfg4 1024^2 fp16 + guided filter  %s max LSB diff %d (bar 6), mean %.3f (bar 1), frac > 1 LSB %.4f (bar 0.05)"
                          % ("ok  " if ok else "FAIL", d.max().item(), d.mean().item(), (d > 1).float().mean().item()), flush=True)
                m2_ = create_model(state_dict=sd, with_D=False)
                m2_.noise = {k: v.to(dev) for k, v in nz.items()}
                out = simple_swap(m2_, g(imgs[0:1]), g(imgs[1:2]), alphas=(1.0,))[1.0]
                rms, mx = errs(out, ref512)
                ok = rms <= r2 and mx <= m2
                RES.append(("512^2 swap %s" % tag, ok))
                print("512^2 full swap recipe     %-5s %s rel RMS %.3e (bar %.0e) max-norm %.3e (bar %.0e)" % (tag, "ok  " if ok else "FAIL", rms, r2, mx, m2), flush=True)
        finally:
            ops.set_precision(0)


def t_train_d():
    """Discriminator update (SURVEY 8 a14, D part): LSGAN losses, every parameter gradient and one
    Adam step against the CPU autograd oracle (oracle/train_oracle.py, pinned to the reference)."""
    import train_oracle as T
    from ppst_amd.networks.discriminator import StyleGAN2Discriminator
    from ppst_amd.train import DiscriminatorTrainer
    # (the 256x256 leg of round 1 went when tests/golden/train512.npz -- the reference's own step at 512x512 -- arrived:
    #  CPU autograd oracle time dominates the GPU suite)
    for size, B in ((128, 2),):
        sd = W.make_state_dict(2, size=size, with_nce=False, bias_std=0.1)
        D = StyleGAN2Discriminator(None, size=size)
        D.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("D.")}, strict=True)
        D = D.to(dev)
        tr = DiscriminatorTrainer(D)
        torch.manual_seed(size)
        real = torch.rand(B, 3, size, size) * 2 - 1
        rec = torch.rand(B // 2, 3, size, size) * 2 - 1
        mix = torch.rand(B, 3, size, size) * 2 - 1
        lo, gr = T.d_step_grads(sd, real, rec, mix, size=size)
        losses = tr.losses_and_grads(g(real), g(rec), g(mix))
        for k in lo:
            report("D-step %d loss %s" % (size, k), losses[k], torch.tensor([lo[k]]), 1e-4)
        worst, worst_k = 0.0, ""
        for k, gref in gr.items():
            got = tr.g(k[2:]).view_as(gref)
            r = rel(got, gref)[0]
            if r > worst:
                worst, worst_k = r, k
        ok = worst <= 5e-3  # deep 4x4 / 8x8 layers: few-term sums with cancellation, error relative to max|grad|
        RES.append(("D-step %d gradients" % size, ok))
        print("D-step %d: worst relative gradient error %.3e (%s) over %d tensors %s" % (size, worst, worst_k, len(gr), "ok" if ok else "FAIL"), flush=True)
        # one Adam step
        keys = list(gr)
        newp = T.adam_reference({k: sd[k] for k in keys}, gr, {}, tr.lr, tr.b1, tr.b2)
        tr.adam()
        w2 = 0.0
        for k in keys:
            off, sz = tr.offsets[k[2:]]
            w2 = max(w2, rel(tr.flat[off:off + sz].view_as(newp[k]), newp[k])[0])
        RES.append(("D-step %d Adam" % size, w2 <= 1e-3)); print("D-step %d: Adam-updated parameters worst rel diff %.3e" % (size, w2), flush=True)


def t_train_ops():
    """Backward building blocks alone (no activation gates): conv input gradient ('dgrad', 'dgrad_s2d'
    = the forward MFMA kernel on a transposed pack) and weight gradient against torch autograd."""
    import torch.nn.functional as F
    torch.manual_seed(5)
    for (B, S, cin, cout) in ((2, 32, 64, 128), (1, 16, 256, 256), (2, 8, 512, 512)):
        x = torch.randn(B, cin, S, S)
        w = torch.randn(cout, cin, 3, 3)
        dy = torch.randn(B, cout, S, S)
        sc = 1.0 / math.sqrt(cin * 9)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        F.conv2d(xr, wr * sc, padding=1).backward(dy)
        wg = g(w)
        dx = ops.ConvPlan(wg, "dgrad", sc)(to_nhwc(g(dy)).contiguous())
        report("dgrad 3x3 %dx%d %d->%d" % (S, S, cout, cin), dx.permute(0, 3, 1, 2), xr.grad, 3e-5)
        dw = ops.conv_wgrad(ops.ConvPlan(wg, "conv", sc), to_nhwc(g(x)).contiguous(), to_nhwc(g(dy)).contiguous())
        report("wgrad 3x3 %dx%d %d->%d" % (S, S, cin, cout), dw, wr.grad, 3e-5)
        # stride-2 conv on the blurred (padded) tensor, as ConvLayer(downsample=True)
        k = torch.tensor([1., 3., 3., 1.]); k2 = k[:, None] * k[None, :]; k2 = k2 / k2.sum()
        xg = to_nhwc(g(x)).contiguous()
        xb, bhw = ops.blur_nhwc(xg, g(k2), 2, 2, ops.PAD_ZERO, s2d=True)
        oh = (bhw[0] - 3) // 2 + 1
        xbr = O.upfirdn2d(x, k2, pad=(2, 2)).detach().requires_grad_(True)
        wr2 = w.clone().requires_grad_(True)
        dy2 = torch.randn(B, cout, oh, oh)
        F.conv2d(xbr, wr2 * sc, stride=2).backward(dy2)
        dy2g = to_nhwc(g(dy2)).contiguous()
        dxb = ops.ConvPlan(wg, "dgrad_s2d", sc)(dy2g, out_hw=bhw)
        report("dgrad_s2d %d->%d out %dx%d" % (cout, cin, bhw[0], bhw[1]), dxb.permute(0, 3, 1, 2), xbr.grad, 3e-5)
        # round 5: the same gradient as ONE stride-1 conv whose output channels stack the four phases + depth_to_space (thin layers)
        ys = ops.ConvPlan(wg, "dgrad_s2ds", sc)(dy2g, out_hw=((bhw[0] + 1) // 2, (bhw[1] + 1) // 2))
        dxs = ops.depth_to_space(ys, bhw)
        report("dgrad_s2ds (phase-stacked) %d->%d out %dx%d" % (cout, cin, bhw[0], bhw[1]), dxs.permute(0, 3, 1, 2), xbr.grad, 3e-5)
        report("dgrad_s2ds vs four-group form", dxs, dxb, 3e-6)
        if cin % 8 == 0:
            prevp = ops.PRECISION["value"]
            ops.set_precision(1)
            try:
                gh = dy2g.to(torch.bfloat16)
                a4 = ops.ConvPlan(wg, "dgrad_s2d", sc)(gh, out_hw=bhw).float()
                as_ = ops.depth_to_space(ops.ConvPlan(wg, "dgrad_s2ds", sc)(gh, out_hw=((bhw[0] + 1) // 2, (bhw[1] + 1) // 2)), bhw).float()
                report("dgrad_s2ds bf16 storage vs four-group form", as_, a4, 8e-3)
            finally:
                ops.set_precision(prevp)
        dw2 = ops.conv_wgrad(ops.ConvPlan(wg, "s2d", sc), xb, dy2g)
        report("wgrad s2d %d->%d" % (cin, cout), dw2, wr2.grad, 3e-5)


def t_train_r1():
    """Lazy R1 penalty: per-sample penalties and the second-order parameter gradients
    against the CPU double-backward oracle.

    Tolerance: the kernels themselves are exact to ~5e-6 (t_train_ops).  End to end, a leaky-ReLU
    whose pre-activation is within the conv rounding error of 0 takes the other slope than in the
    oracle; each such flip changes the image gradient in its receptive-field patch by O(1) of the
    local value (tests/dbg_r1.py shows the isolated patches).  Any two fp32 implementations with
    different summation order differ this way, so the bar is on the L2-relative error per tensor
    (sparse flips average out, a wrong term or scale would give O(1))."""
    import train_oracle as T
    from ppst_amd.networks.discriminator import StyleGAN2Discriminator
    from ppst_amd.train import DiscriminatorTrainer
    oracle_cache = {}
    for size, B, prec in ((128, 2, 0), (128, 2, 2)):      # larger: the reference-generated R1 gradients at 512x512 (train512.npz)
        # prec 2 = exact-fp32 verification convs: the forward then rounds like the oracle's and the tight bar must hold --
        # what remains at prec 0 is gate flips caused by the rounding of the bf16 hi+lo split, not a defect
        sd = W.make_state_dict(3, size=size, with_nce=False, bias_std=0.1)
        D = StyleGAN2Discriminator(None, size=size)
        D.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("D.")}, strict=True)
        D = D.to(dev)
        ops.set_precision(prec)
        tr = DiscriminatorTrainer(D)
        torch.manual_seed(size + 1)
        real = torch.rand(B, 3, size, size) * 2 - 1
        if size not in oracle_cache:
            oracle_cache[size] = T.r1_step_grads(sd, real, size=size)      # CPU double backward: once per size
        pen, gr = oracle_cache[size]
        try:
            losses = tr.r1_losses_and_grads(g(real))
            torch.cuda.synchronize()
        finally:
            ops.set_precision(0)
        report("R1 %d per-sample penalty (precision %d)" % (size, prec), losses["D_R1"], pen, 1e-3)
        worst, worst_k, worst_max = 0.0, "", 0.0
        for k, gref in gr.items():
            got = tr.g(k[2:]).view_as(gref).cpu()
            if float(gref.abs().max()) == 0.0:
                r = rmax = float(got.abs().max())  # biases: exactly no R1 gradient
            else:
                r = float((got - gref).norm() / gref.norm())
                rmax = float((got - gref).abs().max() / gref.abs().max())
            if os.environ.get("PPST_DIAG_VERBOSE"):
                print("   %-50s L2 rel %.3e max rel %.3e" % (k, r, rmax), flush=True)
            worst_max = max(worst_max, rmax)
            if r > worst:
                worst, worst_k = r, k
        # measured (round 2): exact convs 6e-5 / 3.5e-4 (the second-order sweep itself is exact); production convs
        # 2.5e-3 / 3.5e-2 on uniform-noise images, where the rounding of the bf16 hi+lo split flips a few leaky-ReLU gates
        ok = (worst <= 5e-4 and worst_max <= 2e-3) if prec == 2 else (worst <= 5e-3 and worst_max <= 5e-2)
        RES.append(("R1 %d gradients (precision %d)" % (size, prec), ok))
        print("R1 %d precision %d: worst L2-relative gradient error %.3e (%s), worst max-relative %.3e over %d tensors %s"
              % (size, prec, worst, worst_k, worst_max, len(gr), "ok" if ok else "FAIL"), flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("device:", torch.cuda.get_device_name(0), flush=True)
    if which in ("ops", "all"):
        for fn in (t_upfirdn2d, t_fused_act, t_ops_half, t_ops_f64, t_layout_misc, t_conv, t_conv_variants, t_conv_variants_single_pass, t_half_storage, t_conv_k64, t_conv1x1_stream, t_norm_pool, t_corr, t_guided):
            print("== " + fn.__name__, flush=True)
            run(fn)
            torch.cuda.synchronize()
    if which in ("nets", "all"):
        print("== t_networks", flush=True)
        run(t_networks)
    if which == "prec":
        run(t_precision)
    if which == "guided":
        run(t_guided)
    if which == "opshalf":
        run(t_ops_half)
        run(t_ops_f64)
    if which == "trainhalf":
        run(t_train_half)
    if which == "gmp":
        run(t_gmp_multi)
    if which == "ksplit":
        run(t_conv_ksplit)
    if which == "tail":
        run(t_fuse_tail)
    if which == "up9":
        run(t_conv_up9)
    if which == "k64":
        run(t_conv_k64)
    if which == "half":
        run(t_half_storage)
        run(t_conv_dual)
        run(t_conv_variants_single_pass)
    if which == "corr":
        run(t_corr)
    if which == "convv":
        run(t_conv_variants)
        run(t_conv_variants_single_pass)
        run(t_conv1x1_stream)
    if which == "trainops":
        run(t_train_ops)
    if which in ("train", "all"):
        print("== t_train_ops", flush=True)
        run(t_train_ops)
        print("== t_train_d", flush=True)
        run(t_train_d)
        print("== t_train_r1", flush=True)
        run(t_train_r1)
    if which in ("configs", "all"):
        print("== t_configs", flush=True)
        run(t_configs)
    bad = [n for n, ok in RES if not ok]
    print("\nSUMMARY: %d checks, %d failed" % (len(RES), len(bad)))
    for n in bad:
        print("  FAILED:", n)


if __name__ == "__main__":
    main()
