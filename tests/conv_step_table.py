"""Tuning aid (GPU): the ppst_conv2d_mfma launches (forward and input-gradient convs) of one train step (D + G iteration,
batch 2) by shape: (B, tile rows, tile cols, steps, cout, groups, halo, bn).  python tests/conv_step_table.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
from ppst_amd import ops, weights as W
from ppst_amd.ppst_model import Options, create_model
from ppst_amd.train_g import PPSTOptimizer

dev = torch.device("cuda", 0)
sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
model.noise = "random"
real = W.synthetic_images(40, 2).to(dev)
g = torch.Generator().manual_seed(7)
lab = torch.randint(0, 3, (2, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
opt = PPSTOptimizer(model)
data = {"real_A": real, "mask_A": mask}
for _ in range(2):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
ops.prof_enable(True)
N = 3
for _ in range(N):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
detail = ops.prof_detail()
ops.prof_collect(); ops.prof_enable(False)
acc = defaultdict(lambda: [0, 0.0, 0.0])
for ms, fl, info in detail:
    if info[7] == 0:
        continue
    key = tuple(info[:8])
    a = acc[key]; a[0] += 1; a[1] += ms; a[2] += fl
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows) / N
print("conv (forward + input-gradient) launches per step: %d, %.2f ms" % (sum(v[0] for _, v in rows) / N, tot))
for k, v in rows[:40]:
    print("  %-52s n/step %4.1f  %6.3f ms/step  %6.1f TFLOP/s" % (k, v[0] / N, v[1] / N, v[2] / (v[1] * 1e-3) / 1e12 if v[1] else 0))
