"""Diagnostic (GPU): in-kernel clock of the fused conv inside the real benchmark loop.
   tests/build_variant.sh clock -DPPST_CONV_CLOCK && PPST_HIP_LIB=ppst_amd/libppst_hip_clock.so python tests/conv_clock.py
Every conv block stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end -- two scalar
instructions per block, no per-step work -- into a sample buffer; the ratio is the clock the chip holds under the
real load (MI355X guide, DVFS item 6)."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ppst_amd import _lib, glue, ops, weights as W
from ppst_amd.ppst_model import create_model
N = 1 << 16
buf = torch.zeros(N * 2, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.ppst_conv_clock_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.ppst_conv_clock_buffer(buf.data_ptr(), N) == 0
B = 8
sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
m = create_model(state_dict=sd)
m.noise = {k: v.cuda() for k, v in W.make_noise(2, B).items()}
imgs = W.synthetic_images(4, 2 * B).cuda()
with torch.no_grad():
    t0 = time.time()
    n = 0
    while time.time() - t0 < 4.0:       # >= 2 s of back-to-back steps before sampling
        bench.swap_step(m, imgs[:B].contiguous(), imgs[B:].contiguous(), 1.0, glue); n += 1
    torch.cuda.synchronize()
    buf.zero_()
    for _ in range(3):
        bench.swap_step(m, imgs[:B].contiguous(), imgs[B:].contiguous(), 1.0, glue)
    torch.cuda.synchronize()
d = buf.view(N, 2).cpu().numpy().astype(np.float64)
d = d[d[:, 1] > 0]
ghz = d[:, 0] / d[:, 1] * 0.1
life = d[:, 1] * 0.01
print("conv blocks sampled %d (steady state, %d warm steps): in-kernel clock median %.3f GHz  p10 %.3f  p90 %.3f" % (len(d), n, np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)))
big = life > 40.0
if big.any():
    print("  blocks living > 40 us (the big MFMA-bound layers, %d): median %.3f GHz  p10 %.3f  p90 %.3f" % (big.sum(), np.median(ghz[big]), np.percentile(ghz[big], 10), np.percentile(ghz[big], 90)))
