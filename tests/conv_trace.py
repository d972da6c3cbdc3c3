"""Diagnostic (GPU): per-step timeline of the fused conv main loop from the -DPPST_CONV_TRACE build.
   tests/build_variant.sh trace -DPPST_CONV_TRACE && PPST_HIP_LIB=ppst_amd/libppst_hip_trace.so python tests/conv_trace.py"""
import os, sys, math
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
TR_BLOCKS, TR_STEPS = 8, 160
shapes = [(8, 256, 256, 256, 3, "conv"), (8, 128, 128, 512, 3, "conv")]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:6]) + (sys.argv[6],)]
if len(sys.argv) > 7:
    ops.set_precision(int(sys.argv[7]))     # 1: single-pass bf16 (a probe of the non-MFMA floor of a step)
for (B, ci, co, H, k, kind) in shapes:
    # s2d: H is the OUTPUT size, x the space-to-depth copy of the (2H+1)^2 blurred map
    x = torch.randn(B, H + 1, H + 1, 4 * ci, device="cuda") if kind == "s2d" else torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda") / math.sqrt(ci * k * k)
    plan_ = ops.ConvPlan(w, kind=kind)
    plan = (lambda x_, **kw: plan_(x_, out_hw=(H, H), **kw)) if kind == "s2d" else plan_
    for at in ("bn", "n_groups", "nsteps"):
        setattr(plan, at, getattr(plan_, at)) if kind == "s2d" else None
    nw = 8 if plan.bn == 128 else 4
    n_blocks = plan.n_groups * ((co + plan.bn - 1) // plan.bn) * B * ((H + 15) // 16) ** 2
    dbg = torch.zeros(TR_BLOCKS * nw * TR_STEPS * 8 + 2 * n_blocks + TR_BLOCKS, dtype=torch.int64, device="cuda")
    import time
    t0 = time.time()
    while time.time() - t0 < 2.0:        # steady-state clock: >= 2 s of back-to-back launches on random data
        for _ in range(20):
            plan(x)
        torch.cuda.synchronize()
    chk = ops._chk
    ops._chk = lambda t, n="t": None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan(x, prelu=dbg)
    e1.record()
    ops._chk = chk
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1)
    ns = min(plan.nsteps, TR_STEPS)
    clk = dbg[TR_BLOCKS * nw * TR_STEPS * 8:TR_BLOCKS * nw * TR_STEPS * 8 + 2 * n_blocks].view(n_blocks, 2).cpu().numpy().astype(np.float64)
    ghz = clk[:, 0] / np.maximum(clk[:, 1], 1) * 0.1          # cycles per 10-ns reference tick
    print("in-kernel clock (s_memtime / s_memrealtime over each block): median %.3f GHz  p10 %.3f  p90 %.3f;  block lifetime median %.1f us"
          % (np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90), np.median(clk[:, 1]) * 0.01))
    d = dbg[:TR_BLOCKS * nw * TR_STEPS * 8].view(TR_BLOCKS, nw, TR_STEPS, 8).cpu().numpy()[:, :, :ns].astype(np.int64)
    span = (d[:, 0, ns - 1, 0] + d[:, 0, ns - 1, 6] - d[:, 0, 0, 0]).astype(np.float64)     # main loop ticks per traced tile
    n_tiles = plan.n_groups * ((co + plan.bn - 1) // plan.bn) * B * ((H + 15) // 16) ** 2
    print("kernel %.3f ms for %d tiles (%.1f per CU); main loop of a traced tile: median %.0f ticks -> if tiles ran back to back "
          "a tick is <= %.3f ns" % (kms, n_tiles, n_tiles / 256.0, np.median(span), kms * 1e6 / (n_tiles / 256.0) / np.median(span)))
    life = np.median(clk[:, 0])
    print("block lifetime median %.0f cycles; main loop %.0f cycles; prologue + epilogue %.0f cycles" % (life, np.median(span), life - np.median(span)))
    start = dbg[TR_BLOCKS * nw * TR_STEPS * 8 + 2 * n_blocks:].cpu().numpy().astype(np.int64)
    pro = (d[:, 0, 0, 0] - start).astype(np.float64)
    tl = clk[:TR_BLOCKS, 0]
    print("traced blocks: prologue %s  main loop %s  epilogue %s cycles" % (pro.astype(int).tolist(), span.astype(int).tolist(), (tl - pro - span).astype(int).tolist()))
    if plan.nsteps < 8:
        continue
    names = ["head issued", "MFMA g0-1 issued", "MFMA g2-3 issued", "staging done", "vmcnt done", "barrier passed"]
    print("\n%s %d->%d @%d k%d  nsteps %d  (cycles from step start; median over blocks 1..7, all waves)" % (kind, ci, co, H, k, plan.nsteps))
    for label, sel in (("plain steps", d[..., 7] == 0), ("chunk-staging steps", d[..., 7] == 1)):
        # skip the first/last two steps (pipeline edges)
        m = sel.copy(); m[:, :, :2] = False; m[:, :, ns - 2:] = False; m[0] = False
        if not m.any():
            continue
        print("  %s (%d samples)" % (label, int(m.sum())))
        prev = 0.0
        for i, nme in enumerate(names):
            v = d[..., i + 1][m]
            med = float(np.median(v))
            print("    %-18s at %7.0f   (+%6.0f)   p10 %6.0f  p90 %6.0f" % (nme, med, med - prev, np.percentile(v, 10), np.percentile(v, 90)))
            prev = med
    # wave skew at the step start and duration per step
    st = d[1:, :, 2:ns - 2, 0]
    skew = st.max(axis=1) - st.min(axis=1)
    dur = np.diff(d[1:, 0, :, 0], axis=1)[:, 2:ns - 3]
    print("  step duration (wave 0): median %.0f  p10 %.0f  p90 %.0f cycles;  start skew across the 8 waves: median %.0f  p90 %.0f" % (
        np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.median(skew), np.percentile(skew, 90)))
    for wv in range(nw):
        v = d[1:, wv, 2:ns - 2, :]
        pl = v[..., 7] == 0
        print("    wave %d plain: head %5.0f  g01 %5.0f  g23 %5.0f  stage %5.0f  vmcnt %5.0f  barrier %5.0f" % (
            wv, np.median(v[..., 1][pl]), np.median((v[..., 2] - v[..., 1])[pl]), np.median((v[..., 3] - v[..., 2])[pl]),
            np.median((v[..., 4] - v[..., 3])[pl]), np.median((v[..., 5] - v[..., 4])[pl]), np.median((v[..., 6] - v[..., 5])[pl])))
