"""Tuning aid (GPU): cProfile of the host side of one D + one G iteration (the train step is host-bound)."""
import os, sys, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import weights as W
from ppst_amd.ppst_model import Options, create_model
from ppst_amd.train_g import PPSTOptimizer
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
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
