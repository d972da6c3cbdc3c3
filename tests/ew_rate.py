"""Tuning aid (GPU): achieved HBM rate of the elementwise passes of a swap step on their largest shapes, beside a plain copy.
   PPST_HIP_LIB=<other build> python tests/ew_rate.py   for a same-box A/B."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def line(name, ms, nbytes):
    print("%-58s %8.1f us  %6.0f GB/s" % (name, ms * 1e3, nbytes / ms / 1e6), flush=True)


g = torch.Generator(device="cuda").manual_seed(0)
for (B, H, C) in ((16, 512, 128), (16, 256, 256), (24, 512, 32)):
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    y = torch.empty_like(x)
    ss = torch.rand(B, C, 2, device="cuda", generator=g) + 0.5
    nb = x.numel() * 4
    line("copy (%d,%d,%d,%d)" % (B, H, H, C), timeit(lambda: y.copy_(x)), 2 * nb)
    line("affine_act ss+lrelu", timeit(lambda: ops.affine_act(x, ss, act=ops.ACT_LRELU, out=y)), 2 * nb)
    r = torch.randn(B, H, H, C, device="cuda", generator=g)
    line("affine_act ss + res", timeit(lambda: ops.affine_act(x, ss, res=r, out_scale=0.7, out=y)), 3 * nb)
    r2 = torch.randn(B, H // 2, H // 2, C, device="cuda", generator=g)
    line("affine_act ss + res_up2", timeit(lambda: ops.affine_act(x, ss, res=r2, out_scale=0.7, res_up2=True, out=y)), 2.25 * nb)
    line("affine_act_stats ss + res_up2", timeit(lambda: ops.affine_act_stats(x, ss, res=r2, out_scale=0.7, res_up2=True)), 2.25 * nb)
    line("in_stats", timeit(lambda: ops.in_stats(x)), nb)
    del x, y, r, r2
