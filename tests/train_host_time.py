"""Tuning aid (GPU): how long the host takes to ISSUE one train step (D + G iteration, batch 2) against how long the GPU takes
to run it.  The loss read-outs of train_one_step synchronise once per iteration; the issue time is measured with those
replaced by no-ops (PPST_HOST_TIME only inside this script).  python tests/train_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppst_amd import weights as W
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
for _ in range(3):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
N = 6
t0 = time.perf_counter()
for _ in range(N):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
# host-only: profile with cProfile where the time goes
import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(2):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
pr.disable()
torch.cuda.synchronize()
print("wall per step %.1f ms" % wall)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:3500])
