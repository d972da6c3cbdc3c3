"""Tuning aid (GPU): the tile kernel as two 4-wave blocks per CU (ops.TWO_BLOCK_8ROW: 8 x 16 px x 128 ch tiles) against its
16-row form -- outputs must be bit-identical (same MFMA sequence per output element), tile statistics to rounding -- and a
same-process timing of the Cout = 128-class layers of the swap step."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = lambda t: t.to(dev)
nz_ = torch.randn
cases = [
    ("3x3 zero 64->128 48x40 full", 2, 64, 128, 48, 40, "conv", 0, "full"),
    ("3x3 reflect 32->128 33x47", 2, 32, 128, 33, 47, "conv", 1, "plain"),
    ("3x3 replicate 128->128 16x16 inss", 1, 128, 128, 16, 16, "conv", 2, "inss"),
    ("3x3 zero 256->384 24x24 residual", 1, 256, 384, 24, 24, "conv", 0, "res"),
    ("convT 64->128 20x12 full", 2, 64, 128, 20, 12, "convT", 0, "full"),
    ("convT 256->128 64x64 full", 2, 256, 128, 64, 64, "convT", 0, "full"),
    ("dgrad 3x3 (128<-64) 32x32", 2, 64, 128, 32, 32, "dgrad", 0, "plain"),
    ("3x3 zero 128->128 512x512 full", 2, 128, 128, 512, 512, "conv", 0, "full"),
]
ops.CONV_VARIANT["value"], ops.FAT_MIN_BLOCKS = 2, 1 << 30       # N-256 kernel off: every 128-wide plan on the tile kernel
bad = 0
for name, B, ci, co, H, Wd, kind, pm, feat in cases:
    torch.manual_seed(5)
    w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
    outs = []
    for two in (False, True):
        ops.TWO_BLOCK_8ROW.update(value=two, min_blocks=0)
        plan = ops.ConvPlan(w, kind=kind)
        cin_eff = plan.max_chan + 32
        torch.manual_seed(11)
        x = g(nz_(B, H, Wd, cin_eff))
        oh, ow = (2 * H, 2 * Wd) if kind == "convT" else (H, Wd)
        kw = {}
        if feat == "full":
            kw = dict(bias=g(nz_(plan.cout)), noise=g(nz_(B, 1, oh, ow)), noise_weight=0.3, act=ops.ACT_LRELU)
        elif feat == "inss":
            kw = dict(in_ss=g(torch.rand(B, cin_eff, 2) + 0.5), in_act=ops.ACT_PRELU, in_prelu=g(torch.tensor([0.25])),
                      act=ops.ACT_PRELU, prelu=g(torch.tensor([0.1])))
        elif feat == "res":
            kw = dict(residual=g(nz_(B, oh, ow, plan.cout)), res_after_act=True, act=ops.ACT_LRELU, out_scale=0.7)
        y, st = plan(x, pad_mode=pm, stats=True, **kw)
        outs.append((y.cpu(), st.double().sum(1).cpu()))
    same = bool(torch.equal(outs[0][0], outs[1][0]))
    serr = ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()).item()
    bad += (not same) or serr > 1e-5
    print("8-row two-block %-40s %s  max diff %.3e  stats rel %.2e" % (name, "ok  " if same else "FAIL",
                                                                       (outs[0][0] - outs[1][0]).abs().max().item(), serr), flush=True)

# timing: the Cout = 128 layers of the swap step, batch 16
print("timing (ms per launch, median of 20 after 5 warm-ups)")
for name, B, ci, co, H, kind, inss in (("128->128 @512 3x3", 16, 128, 128, 512, "conv", False),
                                       ("128->128 @512 3x3 in_ss", 16, 128, 128, 512, "conv", True),
                                       ("convT 256->128 @256->512", 16, 256, 128, 256, "convT", False),
                                       ("256->128 @256 3x3", 16, 256, 128, 256, "conv", False),
                                       ("256->256 @64 3x3 (tile kernel)", 16, 256, 256, 64, "conv", True)):
    torch.manual_seed(3)
    w = g(nz_(co, ci, 3, 3) / math.sqrt(ci * 9))
    x = g(nz_(B, H, H, ci))
    oh = 2 * H if kind == "convT" else H
    kw = dict(bias=g(nz_(co)), noise=g(nz_(B, 1, oh, oh)), noise_weight=0.3, act=ops.ACT_LRELU)
    if inss:
        kw["in_ss"] = g(torch.rand(B, ci, 2) + 0.5)
    res = []
    for two in (False, True, False, True):
        ops.TWO_BLOCK_8ROW.update(value=two, min_blocks=0)
        plan = ops.ConvPlan(w, kind=kind)
        out = torch.empty((B, oh, oh, co), device=dev)
        ts = []
        for i in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan(x, stats=True, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        res.append(ts[len(ts) // 2])
    fl = 2.0 * B * (H * H) * ci * co * (16 if kind == "convT" else 9)
    print("  %-34s 16-row %.3f / %.3f ms   8-row two-block %.3f / %.3f ms   (%.0f -> %.0f TFLOP/s)" % (
        name, res[0], res[2], res[1], res[3], fl / min(res[0], res[2]) / 1e9, fl / min(res[1], res[3]) / 1e9), flush=True)
sys.exit(1 if bad else 0)
