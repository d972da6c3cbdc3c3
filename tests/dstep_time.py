"""Timing aid (GPU): one discriminator iteration at 512x512 (BASELINE configs[3], D part)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import weights as W
from ppst_amd.ppst_model import create_model
from ppst_amd.train import DiscriminatorTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = W.make_state_dict(0, with_nce=False, bias_std=0.1, noise_weight=0.1)
m = create_model(state_dict=sd, with_D=True)
m.noise = {k: v.cuda() for k, v in W.make_noise(1, B).items()}
tr = DiscriminatorTrainer(m.D)
real = W.synthetic_images(3, B).cuda()
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        losses = tr.train_step(m, real)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("D iteration %d (B=%d, 512^2): %.1f ms  losses %s" % (it, B, dt * 1e3, {k: round(float(v.mean()), 4) for k, v in losses.items()}), flush=True)
    # split: image generation vs D fwd/bwd
    torch.cuda.synchronize(); t0 = time.perf_counter()
    from ppst_amd.train import d_step_images
    rec, mix = d_step_images(m, real)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.losses_and_grads(real, rec, mix)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tr.adam(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("images (E1,E2,G.feat,corrm x2,E2 warp x2,G mix,G rec) %.1f ms | D fwd+bwd x3 %.1f ms | Adam %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r1 = tr.r1_losses_and_grads(real)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("lazy R1 pass (fwd + bwd-to-image + second-order sweep): %.1f ms  penalty %s" % ((t1 - t0) * 1e3, r1["D_R1"].tolist()))
