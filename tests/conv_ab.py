"""Tuning aid (GPU): A/B timing of conv variants inside ONE process (devices differ by
several % between gpurun boxes, so only same-process comparisons are meaningful)."""
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
    return ts[len(ts) // 2], ts[0]

shapes = [(8, 512, 512, 128, 3, "conv"), (8, 256, 256, 256, 3, "conv"), (8, 128, 128, 512, 3, "conv"), (8, 512, 256, 128, 3, "convT"),
          (8, 32, 32, 512, 3, "conv"), (8, 256, 256, 64, 3, "conv")]
for (B, ci, co, H, k, kind) in shapes:
    x = torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda") / math.sqrt(ci * k * k)
    plan = ops.ConvPlan(w, kind=kind)
    ss = torch.ones(B, ci, 2, device="cuda"); ss[..., 1] = 0.1
    flop = 2.0 * 32 * plan.flop_steps * plan.n_groups * co * B * H * H
    out = torch.empty((B, H * (2 if kind == "convT" else 1), H * (2 if kind == "convT" else 1), co), device="cuda")
    res = []
    for name, fn in [("plain", lambda: plan(x, out=out)), ("stats", lambda: plan(x, out=out, stats=True)),
                     ("in_ss", lambda: plan(x, out=out, in_ss=ss)), ("in_ss+lrelu+stats", lambda: plan(x, out=out, in_ss=ss, in_act=1, stats=True))]:
        med, mn = timeit(fn)
        res.append("%s %.3f ms (%.0f TF)" % (name, med, flop / med / 1e9))
    print("%-28s %s" % ("%s %d->%d @%d" % (kind, ci, co, H), " | ".join(res)), flush=True)
