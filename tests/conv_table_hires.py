"""Tuning aid (GPU): per-shape table of the fused conv launches of one 1024^2 encode / encode / decode step (BASELINE configs[4]).
   python tests/conv_table_hires.py [batch] [precision]"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops, weights as W
from ppst_amd.ppst_model import create_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.set_precision(prec)
dev = torch.device("cuda", 0)
sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
m = create_model(state_dict=sd, device=dev)
m.noise = {k: v.to(dev) for k, v in W.make_noise(2, 1, S=128).items()}
imgs = W.synthetic_images(4, 2 * B, size=1024).to(dev)


def step():
    sp, _ = m(imgs[:B], command="encode")
    _, gl = m(imgs[B:], command="encode")
    return m(sp, gl, command="decode")


with torch.no_grad():
    step(); step()
    torch.cuda.synchronize()
    ops.prof_enable(True)
    step()
    torch.cuda.synchronize()
    det = ops.prof_detail()
    tot_ms, n, fl = ops.prof_collect()
agg = collections.OrderedDict()
for ms, f, info in det:
    a = agg.setdefault(info, [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += f
peak = 2500.0 / (3 if prec == 0 else 1)
print("%-46s %4s %9s %9s %8s %6s" % ("(B,th,tw,nsteps,cout,groups,halo,bn)", "n", "ms", "GF", "TF/s", "frac"))
for info, (cnt, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %4d %9.3f %9.1f %8.1f %6.3f" % (info, cnt, ms, f / 1e9, f / ms / 1e9, f / ms / 1e9 / peak))
print("total %.2f ms  %.1f GF  %.1f TF/s  frac %.3f" % (tot_ms, fl / 1e9, fl / tot_ms / 1e9, fl / tot_ms / 1e9 / peak))
