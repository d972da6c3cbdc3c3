"""Tuning aid (GPU): per-shape table of the fused conv launches of one swap step."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import glue, ops, weights as W
from ppst_amd.ppst_model import create_model
from bench import swap_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ops.set_precision(prec)
if len(sys.argv) > 3:
    ops.CONV_VARIANT["value"] = int(sys.argv[3])        # 0: 8-wave kernel everywhere; 1: fat-wave kernel where eligible
if len(sys.argv) > 4:
    ops.FAT_MIN_BLOCKS = int(sys.argv[4])
if len(sys.argv) > 5:
    ops.TWO_BLOCK_128["value"] = bool(int(sys.argv[5]))
if len(sys.argv) > 6:
    ops.DIRECT_MAX["cout"] = int(sys.argv[6])
if len(sys.argv) > 7:
    ops.DIRECT_MAX["nsteps"] = int(sys.argv[7])
if len(sys.argv) > 8:
    ops.DIRECT_MAX["cout3x3"] = int(sys.argv[8])
if len(sys.argv) > 9:
    ops.TALL_TILE_128["value"] = bool(int(sys.argv[9]))
if len(sys.argv) > 10:
    ops.KSPLIT_128["value"] = bool(int(sys.argv[10]))
if len(sys.argv) > 11:
    ops.TILE24_128["value"] = bool(int(sys.argv[11]))
if len(sys.argv) > 12:
    ops.TWO_BLOCK_8ROW["value"] = bool(int(sys.argv[12]))
dev = torch.device("cuda", 0)
sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
m = create_model(state_dict=sd, device=dev)
m.noise = {k: v.to(dev) for k, v in W.make_noise(2, B).items()}
imgs = W.synthetic_images(4, 2 * B).to(dev)
with torch.no_grad():
    swap_step(m, imgs[:B], imgs[B:], 1.0, glue)
    swap_step(m, imgs[:B], imgs[B:], 1.0, glue)
    torch.cuda.synchronize()
    ops.prof_enable(True)
    swap_step(m, imgs[:B], imgs[B:], 1.0, glue)
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
