"""Tuning aid (GPU): run ONE conv shape a few times (for rocprofv3 --pmc)."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
B, ci, co, H, k, kind = 8, 512, 512, 128, 3, "conv"
if len(sys.argv) > 1:
    B, ci, co, H, k = [int(v) for v in sys.argv[1:6]]; kind = sys.argv[6]
if len(sys.argv) > 7:
    ops.TILE_ROWS["value"] = int(sys.argv[7])
x = torch.randn(B, H, H, ci, device="cuda")
w = torch.randn(co, ci, k, k, device="cuda") / math.sqrt(ci * k * k)
plan = ops.ConvPlan(w, kind=kind)
for _ in range(3):
    y = plan(x, stats=True)
torch.cuda.synchronize()
