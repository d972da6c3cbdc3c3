"""Diagnostic (GPU): per-step timeline of conv_mfma2.hip's main loop (variant 2) from the -DPPST_CONV_TRACE build.
   tests/build_variant.sh trace -DPPST_CONV_TRACE && PPST_HIP_LIB=ppst_amd/libppst_hip_trace.so python tests/conv_trace2.py"""
import os, sys, math, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
TRB, TRS, NW = 8, 160, 8
shapes = [(8, 256, 256, 256, 3, "conv"), (8, 512, 512, 128, 3, "conv"), (8, 512, 256, 128, 3, "convT")]
# (shapes with 128 output channels run the K-split kernel, variant 8, when ops.KSPLIT_128 is on: records are per step PAIR)
if os.environ.get("TILE24"):      # trace variant 9 (24 x 16 px x 128 ch blocks) on the 128-wide plans
    ops.TILE24_128.update(value=True, min_blocks=0, max_waste=10.0)
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:6]) + (sys.argv[6],)]
for (B, ci, co, H, k, kind) in shapes:
    x = torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda") / math.sqrt(ci * k * k)
    plan = ops.ConvPlan(w, kind=kind)
    dbg = torch.zeros(TRB * NW * TRS * 8 + 2 * TRB, dtype=torch.int64, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(10):
            plan(x)
        torch.cuda.synchronize()
    chk = ops._chk
    ops._chk = lambda t, n="t": None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan(x, prelu=dbg); e1.record()
    ops._chk = chk
    torch.cuda.synchronize()
    ksplit = plan.bn == 128 and ops.KSPLIT_128["value"] and getattr(plan, "ksplit_ok", False)
    ns = min((plan.nsteps + 1) // 2 if ksplit else plan.nsteps, TRS)
    d = dbg[:TRB * NW * TRS * 8].view(TRB, NW, TRS, 8).cpu().numpy()[:, :, :ns].astype(np.int64)
    se = dbg[TRB * NW * TRS * 8:].view(TRB, 2).cpu().numpy().astype(np.int64)
    print("\n%s %d->%d @%d k%d nsteps %d: kernel %.3f ms" % (kind, ci, co, H, k, plan.nsteps, e0.elapsed_time(e1)))
    names = ["head issued", "m-tiles 0-3 issued", "all m-tiles issued", "B reload issued", "vmcnt done", "barrier passed"]
    for label, sel in (("plain steps", d[..., 7] == 0), ("chunk-staging steps", d[..., 7] == 1)):
        m = sel.copy(); m[:, :, :2] = False; m[:, :, ns - 2:] = False; m[0] = False
        if not m.any():
            continue
        print("  %s (%d samples)" % (label, int(m.sum())))
        prev = 0.0
        for i, nme in enumerate(names):
            v = d[..., i + 1][m]
            med = float(np.median(v))
            print("    %-20s at %7.0f  (+%6.0f)  p10 %6.0f p90 %6.0f" % (nme, med, med - prev, np.percentile(v, 10), np.percentile(v, 90)))
            prev = med
    dur = np.diff(d[1:, 0, :, 0], axis=1)[:, 2:ns - 3]
    st = d[1:, :, 2:ns - 2, 0]
    skew = st.max(axis=1) - st.min(axis=1)
    print("  step duration (wave 0): median %.0f p10 %.0f p90 %.0f cycles; start skew over the 8 waves: median %.0f p90 %.0f" % (
        np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.median(skew), np.percentile(skew, 90)))
    span = (d[1:, 0, ns - 1, 0] + d[1:, 0, ns - 1, 6] - d[1:, 0, 0, 0]).astype(np.float64)
    pro = d[:, 0, 0, 0] - se[:, 0]
    epi = se[:, 1] - (d[:, 0, ns - 1, 0] + d[:, 0, ns - 1, 6])
    print("  traced blocks (first wave of blocks on an empty chip): prologue %s  epilogue %s  lifetime %s cycles" % (
        pro.tolist(), epi.tolist(), (se[:, 1] - se[:, 0]).tolist()))
    print("  main loop of a tile: median %.0f cycles = %.0f per step; MFMA issue floor 3072 per step (2 waves x 96 MFMAs x 16)" % (np.median(span), np.median(span) / ns))
