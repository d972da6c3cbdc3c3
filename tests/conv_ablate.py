"""Tuning aid (GPU): time the plain conv for a few shapes with the library named by PPST_HIP_LIB
(ablation builds from tests/build_variant.sh; their results are numerically wrong on purpose)."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

shapes = [(8, 256, 256, 256, 3, "conv"), (8, 128, 128, 512, 3, "conv"), (8, 512, 512, 128, 3, "conv"), (8, 512, 256, 128, 3, "convT"),
          (8, 32, 32, 512, 3, "conv"), (8, 64, 64, 512, 1, "conv"), (8, 32, 64, 256, 1, "conv"), (8, 64, 64, 256, 3, "conv")]
out = []
for (B, ci, co, H, k, kind) in shapes:
    x = torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda") / math.sqrt(ci * k * k)
    plan = ops.ConvPlan(w, kind=kind)
    flop = 2.0 * 32 * plan.flop_steps * plan.n_groups * co * B * H * H
    y = torch.empty((B, H * (2 if kind == "convT" else 1), H * (2 if kind == "convT" else 1), co), device="cuda")
    med = timeit(lambda: plan(x, out=y))
    out.append("%s%d->%d@%d %.3f ms %4.0f TF" % (kind[4:] or "c", ci, co, H, med, flop / med / 1e9))
print("%-12s %s" % (os.path.basename(os.environ.get("PPST_HIP_LIB", "base")).replace("libppst_hip_", "").replace(".so", ""), " | ".join(out)), flush=True)
