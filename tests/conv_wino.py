"""Tuning aid (GPU): conv_wino.hip (variant 10: Winograd F(2,3) along x) against float64 and, same process, against the direct
kernels (tile kernel / N-256 kernel) on the StyledConv shapes of the 512^2 swap step.

    python tests/conv_wino.py check      # numerics only (small shapes, every epilogue option)
    python tests/conv_wino.py time [B]   # per-shape A/B timing at the swap step's batch (default 16)
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
fails = []


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def report(name, got, ref, tol):
    err = (got.double().cpu() - ref.double().cpu()).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    ok = err <= tol
    print("%-66s err %.3e  (tol %.1e) %s" % (name, err, tol, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        fails.append(name)
    return err


def pad_ref(x, pm):
    return F.pad(x, (1, 1, 1, 1), mode={0: "constant", 1: "reflect", 2: "replicate"}[pm])


def check():
    torch.manual_seed(5)
    ops.WINO["min_blocks"] = 1          # the small shapes here on the Winograd kernel too
    cases = [("64->128 32x32 zero", 2, 64, 128, 32, 32, 0), ("32->128 40x24 ragged zero", 1, 32, 128, 40, 24, 0),
             ("64->256 33x47 reflect", 2, 64, 256, 33, 47, 1), ("128->128 16x16 replicate", 1, 128, 128, 16, 16, 2),
             ("256->384 16x16 zero", 1, 256, 384, 16, 16, 0), ("512->512 8x8 zero", 1, 512, 512, 8, 8, 0),
             ("128->128 64x64 zero", 3, 128, 128, 64, 64, 0), ("96->160 18x50 zero (cout not a multiple of 128)", 1, 96, 160, 18, 50, 0)]
    for name, B, ci, co, H, Wd, pm in cases:
        x = torch.randn(B, ci, H, Wd)
        w = torch.randn(co, ci, 3, 3) / math.sqrt(ci * 9)
        ref = F.conv2d(pad_ref(x.double(), pm), w.double())
        plan = ops.ConvPlan(w.to(dev))
        ops.WINO["value"] = False
        y0 = plan(nhwc(x).to(dev), pad_mode=pm)
        ops.WINO["value"] = True
        y1 = plan(nhwc(x).to(dev), pad_mode=pm)
        e0 = report("direct " + name, nchw(y0), ref, 3e-5)
        e1 = report("wino   " + name, nchw(y1), ref, 3e-5)
    # epilogue options + normalise-on-load
    B, ci, co, H, Wd = 2, 64, 128, 24, 40
    x = torch.randn(B, ci, H, Wd); w = torch.randn(co, ci, 3, 3) / 24.0
    bias = torch.randn(co); noise = torch.randn(B, 1, H, Wd); res = torch.randn(B, co, H, Wd)
    conv = F.conv2d(x.double(), w.double(), padding=1)
    lrelu = lambda t: F.leaky_relu(t, 0.2) * math.sqrt(2.0)
    plan = ops.ConvPlan(w.to(dev))
    g = lambda t: t.to(dev)
    y, st = plan(g(nhwc(x)), bias=g(bias), noise=g(noise), noise_weight=0.3, act=ops.ACT_LRELU, stats=True)
    ref = lrelu(conv + 0.3 * noise.double() + bias.double().view(1, -1, 1, 1))
    report("wino epilogue bias+noise+lrelu", nchw(y), ref, 3e-5)
    s = st.cpu().double().sum(1)
    report("wino tile stats sum", s[..., 0], ref.sum((2, 3)), 1e-5)
    report("wino tile stats sumsq", s[..., 1], (ref ** 2).sum((2, 3)), 1e-5)
    y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_LRELU, residual=g(nhwc(res)), res_after_act=True, out_scale=0.5)
    report("wino residual after act * scale", nchw(y), (lrelu(conv + bias.double().view(1, -1, 1, 1)) + res.double()) * 0.5, 3e-5)
    a = torch.tensor([0.25])
    y = plan(g(nhwc(x)), bias=g(bias), act=ops.ACT_PRELU, prelu=g(a), residual=g(nhwc(res)))
    t = conv + bias.double().view(1, -1, 1, 1) + res.double()
    report("wino residual before prelu", nchw(y), torch.where(t >= 0, t, 0.25 * t), 3e-5)
    big = torch.zeros(B, H, Wd, 200, device=dev)
    plan(g(nhwc(x)), out=big[..., 40:168])
    report("wino out slice", nchw(big[..., 40:168].contiguous()), conv, 3e-5)
    report("wino out slice untouched", big[..., :40], torch.zeros(B, H, Wd, 40), 0)
    # normalise on load (zero and replicate padding of the NORMALISED tensor), with PReLU
    ss = torch.stack([torch.rand(B, ci) + 0.5, torch.randn(B, ci)], -1).contiguous()
    xn = x.double() * ss[..., 0].double().view(B, ci, 1, 1) + ss[..., 1].double().view(B, ci, 1, 1)
    for pm in (0, 2):
        for in_act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_PRELU, lambda t: torch.where(t >= 0, t, 0.25 * t)),
                           (ops.ACT_LRELU, lrelu)):
            ref = F.conv2d(pad_ref(fn(xn), pm), w.double())
            y = plan(g(nhwc(x)), pad_mode=pm, in_ss=g(ss), in_act=in_act, in_prelu=g(a))
            report("wino normalise-on-load pad %d act %d" % (pm, in_act), nchw(y), ref, 3e-5)
    # channel-slice input (pixel stride > channels) and a dgrad plan
    xb = torch.randn(B, H, Wd, 160)
    y = plan(g(xb)[..., 32:96])
    report("wino input slice", nchw(y), F.conv2d(xb[..., 32:96].permute(0, 3, 1, 2).double(), w.double(), padding=1), 3e-5)
    wd = torch.randn(128, 256, 3, 3) / 30.0     # forward (Cout = 128, Cin = 256): its input gradient maps 128 -> 256
    dy = torch.randn(B, 128, H, Wd)
    pd = ops.ConvPlan(g(wd), kind="dgrad")
    yd = pd(g(nhwc(dy)))
    report("wino dgrad 128->256", nchw(yd), F.conv_transpose2d(dy.double(), wd.double(), padding=1), 3e-5)
    ops.WINO["value"] = False
    print("FAILED: %s" % fails if fails else "all ok")
    return 1 if fails else 0


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_shapes(B):
    torch.manual_seed(1)
    shapes = [(128, 128, 512), (256, 256, 256), (512, 512, 128), (256, 256, 64), (512, 512, 64), (256, 128, 256), (512, 128, 128),
              (384, 512, 64), (128, 128, 128)]
    print("%-26s %9s %9s %8s %8s %7s" % ("cin->cout @HxW (B=%d)" % B, "direct ms", "wino ms", "TF/s d", "TF/s w", "ratio"))
    for ci, co, S in shapes:
        x = torch.randn(B, S, S, ci, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
        bias = torch.randn(co, device=dev); noise = torch.randn(B, 1, S, S, device=dev)
        plan = ops.ConvPlan(w)
        out = torch.empty(B, S, S, co, device=dev)
        fl = 2.0 * B * S * S * ci * co * 9

        def run():
            plan(x, bias=bias, noise=noise, noise_weight=0.1, act=ops.ACT_LRELU, stats=True, out=out)
        ops.WINO["value"] = False
        t0 = timeit(run)
        ops.WINO["value"] = True
        t1 = timeit(run)
        ops.WINO["value"] = False
        t0b = timeit(run)
        t0 = min(t0, t0b)
        print("%4d->%-4d @%-4d            %9.3f %9.3f %8.1f %8.1f %7.3f" % (ci, co, S, t0, t1, fl / t0 / 1e9, fl / t1 / 1e9, t0 / t1), flush=True)


def time_wino_only(B):
    """one line per library (PPST_HIP_LIB = an ablation build of tests/build_wino_variant.sh): Winograd kernel only"""
    torch.manual_seed(1)
    ops.WINO["value"] = True
    out = []
    for ci, co, S in [(128, 128, 512), (256, 256, 256), (512, 512, 128), (256, 128, 256)]:
        x = torch.randn(B, S, S, ci, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
        plan = ops.ConvPlan(w)
        y = torch.empty(B, S, S, co, device=dev)
        t = timeit(lambda: plan(x, stats=True, out=y))
        # the StyledConv conv2 call: normalise-on-load + bias + noise + leaky ReLU + statistics
        ss = torch.stack([torch.rand(B, ci, device=dev) + 0.5, torch.randn(B, ci, device=dev)], -1).contiguous()
        bias = torch.randn(co, device=dev); nz = torch.randn(B, 1, S, S, device=dev)
        t1 = timeit(lambda: plan(x, stats=True, out=y, bias=bias, noise=nz, noise_weight=0.1, act=ops.ACT_LRELU))
        t2 = timeit(lambda: plan(x, stats=True, out=y, in_ss=ss, bias=bias, noise=nz, noise_weight=0.1, act=ops.ACT_LRELU))
        out.append("%d->%d@%d %.3f / %.3f / %.3f ms" % (ci, co, S, t, t1, t2))
    print("%-10s %s" % (os.path.basename(os.environ.get("PPST_HIP_LIB", "base")).replace("libppst_hip_", "").replace(".so", ""),
                        " | ".join(out)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(check())
    if mode == "one":          # one shape, a few launches of each kernel: for rocprofv3 --pmc passes
        ci, co, S, B = 512, 512, 128, 16
        x = torch.randn(B, S, S, ci, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
        plan = ops.ConvPlan(w); y = torch.empty(B, S, S, co, device=dev)
        for flag in (False, True):
            ops.WINO["value"] = flag
            for _ in range(6):
                plan(x, stats=True, out=y)
        torch.cuda.synchronize()
        sys.exit(0)
    if mode == "abl":
        time_wino_only(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
        sys.exit(0)
    time_shapes(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
