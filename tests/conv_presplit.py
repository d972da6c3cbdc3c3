"""Tuning aid (GPU, PPST_EXPERIMENTS build): the tile kernel with PRE-SPLIT activations staged by LDS-DMA (VERDICT r2 lever (i))
against the production form on the same plans -- bit-identical outputs expected (same hi / lo operands, same MFMA order) --
and a same-process timing of the Cout = 128-class layers (the time of the split pass itself is reported separately: in a full
integration the producer's epilogue would write the split layout)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops  # noqa: E402

assert ops.EXPERIMENTS, "build the library with PPST_EXPERIMENTS=1"
dev = torch.device("cuda", 0)
g = lambda t: t.to(dev)
nz_ = torch.randn
ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = 2, 1 << 30       # every 128-wide plan on the tile kernel
bad = 0
for name, B, ci, co, H, Wd, kind, pm, feat in (("3x3 zero 64->128 48x40 full", 2, 64, 128, 48, 40, "conv", 0, "full"),
                                               ("3x3 reflect 32->128 33x47", 2, 32, 128, 33, 47, "conv", 1, "plain"),
                                               ("3x3 replicate 128->256 16x16 res", 1, 128, 256, 16, 16, "conv", 2, "res"),
                                               ("convT 64->128 20x12 full", 2, 64, 128, 20, 12, "convT", 0, "full"),
                                               ("convT 256->128 64x64 full", 2, 256, 128, 64, 64, "convT", 0, "full"),
                                               ("3x3 zero 128->128 512x512 full", 2, 128, 128, 512, 512, "conv", 0, "full")):
    torch.manual_seed(5)
    w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
    plan = ops.ConvPlan(w, kind=kind)
    x = g(nz_(B, H, Wd, plan.max_chan + 32))
    oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
    kw = {}
    if feat == "full":
        kw = dict(bias=g(nz_(plan.cout)), noise=g(nz_(B, 1, oh, ow)), noise_weight=0.3, act=ops.ACT_LRELU)
    elif feat == "res":
        kw = dict(residual=g(nz_(B, oh, ow, plan.cout)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
    y0, s0 = plan(x, pad_mode=pm, stats=True, **kw)
    y1, s1 = plan(ops.presplit(x), pad_mode=pm, stats=True, presplit=True, **kw)
    same = bool(torch.equal(y0, y1)) and bool(torch.equal(s0, s1))
    bad += not same
    print("pre-split %-36s %s max diff %.3e" % (name, "ok  " if same else "FAIL", (y0 - y1).abs().max().item()), flush=True)

print("timing (ms per launch, median of 20): production (fp32 input, split while staging) vs pre-split input (LDS-DMA staging)")
for name, B, ci, co, H, kind in (("128->128 @512 3x3", 16, 128, 128, 512, "conv"), ("convT 256->128 @256->512", 16, 256, 128, 256, "convT"),
                                 ("256->128 @256 3x3", 16, 256, 128, 256, "conv"), ("256->256 @64 3x3 (bn 128)", 16, 256, 256, 64, "conv"),
                                 ("128->128 @512 3x3 B=8", 8, 128, 128, 512, "conv")):
    torch.manual_seed(3)
    w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
    x = g(nz_(B, H, H, ci))
    oh = 2 * H if kind == "convT" else H
    kw = dict(bias=g(nz_(co)), noise=g(nz_(B, 1, oh, oh)), noise_weight=0.3, act=ops.ACT_LRELU)
    plan = ops.ConvPlan(w, kind=kind)
    out = torch.empty((B, oh, oh, co), device=dev)
    xs = ops.presplit(x)
    res = []
    for pre in (False, True, False, True):
        ts = []
        for i in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan(xs if pre else x, stats=True, out=out, presplit=pre, **kw)
            e1.record()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        res.append(ts[len(ts) // 2])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.presplit(x)
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * B * H * H * ci * co * (16 if kind == "convT" else 9)
    print("  %-28s production %.3f / %.3f ms   pre-split %.3f / %.3f ms   (%.0f -> %.0f TFLOP/s, %+.1f %%)   split pass alone %.3f ms" % (
        name, res[0], res[2], res[1], res[3], fl / min(res[0], res[2]) / 1e9, fl / min(res[1], res[3]) / 1e9,
        (min(res[0], res[2]) / min(res[1], res[3]) - 1) * 100, e0.elapsed_time(e1) / 10), flush=True)
sys.exit(1 if bad else 0)
