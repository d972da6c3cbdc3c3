"""Tuning aid (GPU): per-shape table of the weight-gradient launches (and, with 'conv', the conv launches) of one train step."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops, weights as W  # noqa: E402
from ppst_amd.ppst_model import Options, create_model  # noqa: E402
from ppst_amd.train_g import PPSTOptimizer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
dev = torch.device("cuda", 0)
B = 2
sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
real = W.synthetic_images(40, B).to(dev)
g = torch.Generator().manual_seed(7)
lab = torch.randint(0, 3, (B, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
opt = PPSTOptimizer(model)
data = {"real_A": real, "mask_A": mask}
for _ in range(2):
    opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
ops.prof_enable(True)
opt.train_one_step(data, 0); opt.train_one_step(data, 0)
torch.cuda.synchronize()
det = ops.prof_detail()
ops.prof_collect()
ops.prof_enable(False)
rows = [r for r in det if (r[2][7] == 0) == (which == "wgrad")]
agg = collections.OrderedDict()
for ms, f, info in rows:
    a = agg.setdefault(info, [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += f
hdr = "(B,oh,ow,nsteps,cout,nchunks,splits,0)" if which == "wgrad" else "(B,th,tw,nsteps,cout,groups,halo,bn)"
print("%-46s %4s %9s %9s %8s %6s" % (hdr, "n", "ms", "GF", "TF/s", "frac"))
for info, (cnt, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %4d %9.3f %9.1f %8.1f %6.3f" % (info, cnt, ms, f / 1e9, f / ms / 1e9, f / ms / 1e9 / 833.3))
tm, tf = sum(v[1] for v in agg.values()), sum(v[2] for v in agg.values())
print("total %.2f ms  %.1f GF  %.1f TF/s  frac %.3f   (%d launches)" % (tm, tf / 1e9, tf / tm / 1e9, tf / tm / 1e9 / 833.3, len(rows)))
