"""Tuning aid (GPU): is the swapping_grid workload host- or GPU-bound?  Host time to issue one folder vs time until the GPU is done,
and a cProfile of the host side."""
import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import weights as W
from ppst_amd.evaluation import swapping_grid
from ppst_amd.ppst_model import create_model
dev = torch.device("cuda", 0)
sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
model = create_model(state_dict=sd, device=dev)
model.noise = {k: v.to(dev) for k, v in W.make_noise(2, 1).items()}
contents, styles = W.synthetic_images(4, 8).to(dev), W.synthetic_images(5, 8).to(dev)
with torch.no_grad():
    for _ in range(2):
        swapping_grid(model, contents, styles, 0, 1, smooth=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    swapping_grid(model, contents, styles, 0, 1, smooth=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue %.1f ms; GPU done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    swapping_grid(model, contents, styles, 0, 1, smooth=True)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
