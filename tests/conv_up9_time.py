"""Tuning aid (GPU): the fused upscale as nine products per input pixel (ops.UP9, variant 11) against the four-phase forms, same
process, on the generator's three upsampling layers with the StyledConv epilogue.   python tests/conv_up9_time.py [batch]"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


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


print("%-24s %10s %10s %8s %8s %7s" % ("cin->cout @in (B=%d)" % B, "4-phase ms", "9-prod ms", "frac 4p", "frac 9p", "ratio"))
for ci, co, S in [(256, 128, 256), (512, 256, 128), (512, 512, 64)]:
    x = torch.randn(B, S, S, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
    bias = torch.randn(co, device=dev); noise = torch.randn(B, 1, 2 * S, 2 * S, device=dev)
    out = torch.empty(B, 2 * S, 2 * S, co, device=dev)
    fl = 2.0 * B * S * S * ci * co * 16
    ts = []
    for on in (False, True, False):
        ops.UP9["value"] = on
        plan = ops.ConvPlan(w, kind="convT")
        ts.append(timeit(lambda: plan(x, bias=bias, noise=noise, noise_weight=0.1, act=ops.ACT_LRELU, stats=True, out=out)))
    t0, t1 = min(ts[0], ts[2]), ts[1]
    print("%4d->%-4d @%-4d         %10.3f %10.3f %8.3f %8.3f %7.3f" % (ci, co, S, t0, t1, fl / t0 / 1e9 / 833.3, fl / t1 / 1e9 / 833.3, t0 / t1), flush=True)
