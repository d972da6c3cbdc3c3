"""Tuning aid (GPU): the LDS-DMA / transposed-read weight-gradient kernel (ops.WGRAD_TR) against the register-staged bf16x3
kernel of round 2 and against float64 torch autograd, on every plan kind incl. ragged sizes; its fused column sums against
ops.colsum; then a same-process timing of the heavy layers of the train step."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = lambda t: t.to(dev)
bad = 0


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


cases = [("conv 3x3 64->128 48x40", "conv", 2, 64, 128, 48, 40, 3), ("conv 3x3 32->32 33x47 (ragged)", "conv", 2, 32, 32, 33, 47, 3),
         ("conv 1x1 128->64 24x24", "conv", 3, 128, 64, 24, 24, 1), ("conv 3x3 128->36 17x9", "conv", 1, 128, 36, 17, 9, 3),
         ("conv 3x3 256->256 64x64", "conv", 2, 256, 256, 64, 64, 3), ("s2d 32->64 -> 20x23", "s2d", 2, 32, 64, 20, 23, 3),
         ("convT (dgradT) 64->128 20x12", "dgradT", 2, 64, 128, 20, 12, 3), ("conv 3x3 128->128 512x512", "conv", 2, 128, 128, 512, 512, 3)]
for name, kind, B, ci, co, H, Wd, k in cases:
    torch.manual_seed(3)
    w = g(torch.randn(co, ci, k, k) / math.sqrt(ci * k * k))
    plan = ops.ConvPlan(w, kind=kind)
    if kind == "conv":
        x, dy = g(torch.randn(B, H, Wd, ci)), g(torch.randn(B, H, Wd, co))
        xr = x.permute(0, 3, 1, 2).double().cpu()
        wr = w.double().cpu().requires_grad_(True)
        F.conv2d(xr, wr, padding=k // 2).backward(dy.permute(0, 3, 1, 2).double().cpu())
        ref = wr.grad
    elif kind == "s2d":     # forward: 3x3 stride-2 conv (no padding) over a (2H+1, 2W+1) tensor, read as its space-to-depth copy
        xf = torch.randn(B, ci, 2 * H + 1, 2 * Wd + 1)
        x = ops.space_to_depth(g(xf.permute(0, 2, 3, 1).contiguous()))
        dy = g(torch.randn(B, H, Wd, co))
        wr = w.double().cpu().requires_grad_(True)
        F.conv2d(xf.double(), wr, stride=2).backward(dy.permute(0, 3, 1, 2).double().cpu())
        ref = wr.grad
    else:                    # dgradT: dW4 of the fused transposed conv = wgrad(plan, s2d(dY), x)
        xin = torch.randn(B, ci, H, Wd)
        dyf = torch.randn(B, co, 2 * H, 2 * Wd)
        x = ops.space_to_depth(g(dyf.permute(0, 2, 3, 1).contiguous()))
        dy = g(xin.permute(0, 2, 3, 1).contiguous())
        w4 = plan.wsrc.double().cpu().requires_grad_(True)           # (Cin, Cout, 4, 4)
        F.conv_transpose2d(xin.double(), w4, stride=2, padding=1).backward(dyf.double())
        ref = w4.grad
    outs = {}
    for key, on, form in ((False, False, 2), (True, True, 2), ("f1", True, 1)):
        ops.WGRAD_TR["value"], ops.WGRAD_TR["form"] = on, form
        dw, db = ops.conv_wgrad(plan, x, dy, want_bias=True)
        outs[key] = (dw.cpu(), db.cpu())
    ops.WGRAD_TR["value"], ops.WGRAD_TR["form"] = True, 2
    e_ref = rel(outs[True][0] / plan.scale, ref)
    e_old = max(rel(outs[True][0], outs[False][0]), rel(outs["f1"][0], outs[False][0]))
    e_b = rel(outs[True][1], dy.reshape(-1, dy.shape[3]).double().sum(0).cpu())
    # accumulate into a destination
    dst = torch.full_like(outs[True][0], 0.5).to(dev)
    bdst = torch.full((dy.shape[3],), 0.25, device=dev)
    ops.conv_wgrad(plan, x, dy, out=dst, accumulate=True, bias_out=bdst, bias_accumulate=True)
    e_acc = max(rel(dst.cpu() - 0.5, outs[True][0]), rel(bdst.cpu() - 0.25, outs[True][1]))
    # a destination full of NaN, accumulate off: every element must be WRITTEN (plans whose table covers the whole weight skip
    # the zero fill in front of the split reduction)
    nan_dst = torch.full_like(outs[True][0], float("nan")).to(dev)
    ops.conv_wgrad(plan, x, dy, out=nan_dst)
    e_cover = rel(nan_dst.cpu(), outs[True][0]) if torch.isfinite(nan_dst).all() else float("inf")
    e_acc = max(e_acc, e_cover)
    # precision mode 1 (bf16 compute, fp32 master weights): one MFMA pass over the hi halves; the bias sums stay fp32
    ops.set_precision(1)
    try:
        dw1, db1 = ops.conv_wgrad(plan, x, dy, want_bias=True)
    finally:
        ops.set_precision(0)
    e_1 = rel(dw1.cpu() / plan.scale, ref)
    e_b1 = rel(db1.cpu(), outs[True][1])
    ok = e_ref <= 3e-5 and e_old <= 2e-5 and e_b <= 2e-6 and e_acc <= 1e-5 and 1e-4 < e_1 <= 1e-2 and e_b1 == 0.0
    bad += not ok
    print("%-34s %s vs f64 autograd %.2e  vs round-2 kernel %.2e  fused bias %.2e  accumulate / NaN-filled destination %.2e  single-pass bf16 %.2e  (table covers the weight: %s)" % (name, "ok  " if ok else "FAIL", e_ref, e_old, e_b, e_acc, e_1, plan.full_cover), flush=True)

print("timing (ms per weight gradient incl. the split reduction; median of 15)")
for name, kind, B, ci, co, H, k in (("128->128 @512 3x3", "conv", 2, 128, 128, 512, 3), ("256->256 @256 3x3", "conv", 2, 256, 256, 256, 3),
                                    ("512->512 @128 3x3", "conv", 2, 512, 512, 128, 3), ("32->32 @512 3x3", "conv", 2, 32, 32, 512, 3),
                                    ("64->64 @256 3x3", "conv", 2, 64, 64, 256, 3), ("128->64 @512 1x1", "conv", 2, 128, 64, 512, 1),
                                    ("convT 512->256 @128", "dgradT", 2, 512, 256, 128, 3), ("D 64->128 @256 B=5", "conv", 5, 64, 128, 256, 3)):
    torch.manual_seed(3)
    w = g(torch.randn(co, ci, k, k) / math.sqrt(ci * k * k))
    plan = ops.ConvPlan(w, kind=kind)
    if kind == "conv":
        x, dy = g(torch.randn(B, H, H, ci)), g(torch.randn(B, H, H, co))
        fl = 2.0 * B * H * H * ci * co * k * k
    else:
        x, dy = g(torch.randn(B, H, H, 4 * co)), g(torch.randn(B, H, H, ci))
        fl = 2.0 * B * H * H * ci * co * 16
    res = []
    for on, form in ((False, 2), (True, 2), (True, 1), (True, 2)):
        ops.WGRAD_TR["value"], ops.WGRAD_TR["form"] = on, form
        ts = []
        for i in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv_wgrad(plan, x, dy, want_bias=(kind == "conv"))
            e1.record()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        res.append(ts[len(ts) // 2])
    ops.WGRAD_TR["value"], ops.WGRAD_TR["form"] = True, 2
    ops.set_precision(1)
    ts = []
    for i in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv_wgrad(plan, x, dy, want_bias=(kind == "conv"))
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1))
    ops.set_precision(0)
    ts.sort()
    print("  %-24s round-2 kernel (+ colsum) %.3f ms   one block / CU %.3f ms   two blocks / CU (in place) %.3f / %.3f ms   (%.0f -> %.0f TFLOP/s)   single-pass bf16 %.3f ms" % (
        name, res[0], res[2], res[1], res[3], fl / res[0] / 1e9, fl / min(res[1], res[3]) / 1e9, ts[len(ts) // 2]), flush=True)
# timing ablations of the new kernel (results wrong on purpose): what each phase of a tile costs
import ctypes
from ppst_amd._lib import lib
lib.ppst_wgrad_ablate.restype = ctypes.c_int
print("ablations, 128->128 @512 3x3 B=2 (ms incl. split reduction): ", end="")
w = g(torch.randn(128, 128, 3, 3) / 34.0)
plan = ops.ConvPlan(w)
x, dy = g(torch.randn(2, 512, 512, 128)), g(torch.randn(2, 512, 512, 128))
for mask, tag in ((0, "full"), (1, "no MFMA"), (2, "no conversion"), (4, "no DMA"), (3, "DMA only"), (6, "MFMA only"), (5, "conversion only"), (7, "barriers only")):
    lib.ppst_wgrad_ablate(mask)
    ts = []
    for i in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv_wgrad(plan, x, dy)
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("%s %.3f | " % (tag, ts[len(ts) // 2]), end="")
lib.ppst_wgrad_ablate(0)
print()
sys.exit(1 if bad else 0)
