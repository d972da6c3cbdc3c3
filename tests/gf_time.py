"""Tuning aid (GPU): time ops.guided_filter on a batch of 1024^2 images (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppst_amd import ops
from ppst_amd._lib import lib
if os.environ.get("GF_VS"):          # "vs1,vs2": rows per block of the two fused launches
    lib.ppst_guided_filter_tune(*[int(v) for v in os.environ["GF_VS"].split(",")])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device="cuda")
s = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ops.guided_filter(g, s, 30, (0.02 * 255) ** 2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.guided_filter(g, s, 30, (0.02 * 255) ** 2)
e1.record(); torch.cuda.synchronize()
print("guided filter B=%d %dx%d: %.3f ms" % (B, S, S, e0.elapsed_time(e1) / 10))
