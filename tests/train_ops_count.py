"""Tuning aid (GPU): torch-native operators issued by one D + one G iteration (count, device time), and host time per step."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import weights as W
from ppst_amd.ppst_model import Options, create_model
from ppst_amd.train_g import PPSTOptimizer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
B = 2
sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
model.noise = "random"
real = W.synthetic_images(40, B).to(dev)
g = torch.Generator().manual_seed(7)
lab = torch.randint(0, 3, (B, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
opt = PPSTOptimizer(model, world=1)
data = {"real_A": real, "mask_A": mask}
for _ in range(3):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
opt.train_one_step(data, 0); opt.train_one_step(data, 0)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host time to ISSUE one D + one G iteration: %.1f ms; until the GPU is done: %.1f ms" % (t_host * 1e3, t_all * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total / 1e3, e.self_cpu_time_total / 1e3) for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[1])
print("%-40s %6s %10s %10s" % ("op", "count", "gpu ms", "cpu ms"))
for r in rows[:30]:
    print("%-40s %6d %10.2f %10.2f" % r)
