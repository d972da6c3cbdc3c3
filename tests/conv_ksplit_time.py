"""Tuning aid: time of ONE small conv launch against the across-block K split S (ppst_conv_args.ksplit), back-to-back launches of
the same plan timed as a whole (the device stays busy: launch overhead is hidden, the figure is the kernel's).
  python tests/conv_ksplit_time.py [bf16|bf16x3] [wino]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ppst_amd import ops  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
wino = len(sys.argv) > 2 and sys.argv[2] == "wino"
ops.set_precision({"bf16x3": 0, "bf16": 1}[prec])
ops.WINO["value"] = wino
ops.WINO["ksplit_fill"] = 0
ops.BATCH_AWARE["value"] = True
if wino:
    ops.KSPLIT["variants"] = (0, 2, 10)
dev = torch.device("cuda:0")
dt = torch.bfloat16 if prec == "bf16" else torch.float32
forced = {"S": 0}
_choice = ops._ksplit_choice


def _forced_choice(blocks, chunk_starts, mb, ms, max_s=8):
    S = forced["S"]
    if not S or S > max_s or (S - 1) * blocks > 256:
        return 0, None
    # the production cut rule, forced to this S: every share allowed, the block budget lifted
    for cand in (S,):
        nsteps = chunk_starts[-1]
        cuts = [0] + [min(chunk_starts, key=lambda c: (abs(c - i * nsteps / cand), c)) for i in range(1, cand)] + [nsteps]
        if all(b_ > a_ for a_, b_ in zip(cuts[:-1], cuts[1:])):
            return cand, cuts
    return 0, None


ops._ksplit_choice = _forced_choice
N = 300
print("%s%s: us per launch by S (- : not applicable)" % (prec, " wino" if wino else ""))
print("%-28s %8s %8s %8s %8s" % ("layer", "S=1", "S=2", "S=4", "S=8"))
for name, B, ci, co, H in [("256->256 64^2 B2", 2, 256, 256, 64), ("256->256 64^2 B1", 1, 256, 256, 64), ("256->256 64^2 B4", 4, 256, 256, 64),
                           ("512->512 64^2 B2", 2, 512, 512, 64), ("512->512 64^2 B1", 1, 512, 512, 64), ("512->512 32^2 B4", 4, 512, 512, 32),
                           ("512->512 16^2 B4", 4, 512, 512, 16), ("512->512 8^2 B4", 4, 512, 512, 8), ("512->512 4^2 B4", 4, 512, 512, 4),
                           ("384->384 64^2 B2", 2, 384, 384, 64), ("128->128 64^2 B2", 2, 128, 128, 64)]:
    x = torch.randn(B, H, H, ci, device=dev).to(dt)
    w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
    bias = torch.randn(co, device=dev)
    plan = ops.ConvPlan(w)
    out = torch.empty(B, H, H, co, device=dev, dtype=dt)
    row = []
    for S in (1, 2, 4, 8):
        forced["S"] = 0 if S == 1 else S
        ops.KSPLIT["value"] = S > 1
        a_probe = plan(x, bias=bias, act=ops.ACT_LRELU, out=out, stats=True)
        for _ in range(10):
            plan(x, bias=bias, act=ops.ACT_LRELU, out=out, stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            plan(x, bias=bias, act=ops.ACT_LRELU, out=out, stats=True)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / N * 1e3)
    print("%-28s %8.1f %8.1f %8.1f %8.1f" % (name, *row), flush=True)
