"""Diagnostic (GPU): per-phase cycle shares of the fused conv main loop, from the
-DPPST_CONV_STAMP build (ppst_amd/libppst_hip_stamp.so).  Usage:
   PPST_HIP_LIB=ppst_amd/libppst_hip_stamp.so python tests/conv_stamp.py"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
dev = "cuda"
names = ["setup", "prologue", "mainloop", "epilogue", "top+mfma", "stage-store", "barrier", "-"]
for (B, ci, co, H, k, kind) in [(8, 512, 512, 128, 3, "conv"), (8, 128, 128, 512, 3, "conv"), (8, 512, 256, 128, 3, "convT"), (8, 32, 32, 512, 3, "conv"), (8, 256, 256, 64, 1, "conv")]:
    x = torch.randn(B, H, H, ci, device=dev)
    w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(ci * k * k)
    plan = ops.ConvPlan(w, kind=kind)
    n_tiles = (co + plan.bn - 1) // plan.bn
    blocks = plan.n_groups * n_tiles * B * ((H + 15) // 16) ** 2
    nw = 8 if plan.bn == 128 else 4
    dbg = torch.zeros(blocks * nw * 8, dtype=torch.int64, device=dev)
    plan(x)  # warm
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    y = plan(x, prelu=dbg.view(torch.float32)[: 1] if False else None)
    ev1.record()
    # second run with the debug buffer (the prelu slot carries it in the stamp build)
    import ctypes
    from ppst_amd import _lib
    orig = ops._chk
    ops._chk = lambda t, n="t": None
    y = plan(x, prelu=dbg)
    ops._chk = orig
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    d = dbg.view(blocks, nw, 8).double()
    med = d.median(0)[0].median(0)[0]
    tot = med[:4].sum().item()
    print("\n%s B%d %d->%d @%d k%d: blocks %d nsteps %d  kernel %.3f ms; per-block cycles (median) total %.0f" % (kind, B, ci, co, H, k, blocks, plan.nsteps, ms, tot))
    for i in range(7):
        print("   %-12s %10.0f  (%5.1f%%)%s" % (names[i], med[i].item(), 100 * med[i].item() / tot, "   per step %.0f" % (med[i].item() / plan.nsteps) if i in (2, 4, 5, 6) else ""))
    flops = 2.0 * 32 * plan.nsteps * plan.n_groups * co * B * H * H
    print("   algorithmic %.1f TF/s" % (flops / ms / 1e9))
