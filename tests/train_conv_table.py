"""Tuning aid: the conv / weight-gradient launches of ONE train step (configs[3], batch 2) grouped by shape, from the HIP-event
brackets of ppst_prof_* -- which launches are small (few blocks, one long serial K chain) and what they cost together.
  python tests/train_conv_table.py [bf16|bf16x3] [batch]"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ppst_amd import ops, weights as W  # noqa: E402
from ppst_amd.ppst_model import Options, create_model  # noqa: E402
from ppst_amd.train_g import PPSTOptimizer  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ops.set_precision({"bf16x3": 0, "bf16": 1}[prec])
dev = torch.device("cuda:0")
sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
model.noise = "random"
real = W.synthetic_images(40, B).to(dev)
g = torch.Generator().manual_seed(7)
lab = torch.randint(0, 3, (B, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
opt = PPSTOptimizer(model, world=1)
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
ops.prof_collect()
ops.prof_enable(False)
groups = defaultdict(lambda: [0, 0.0, 0.0])
for ms, fl, info in detail:
    b, th, tw, nsteps, cout, ng, halo, code = info
    wg = (code & 0xfff) == 0
    key = ("wgrad" if wg else "conv", b, th, tw, nsteps, cout, ng, (code >> 12) & 0xff, code & 0xfff, round(fl / 1e9, 2))
    e = groups[key]
    e[0] += 1; e[1] += ms; e[2] += fl
tot = sum(e[1] for e in groups.values()) / N
print("%s batch %d: %d bracketed launches / step, %.2f ms / step" % (prec, B, len(detail) // N, tot))
print("%-6s %2s %3s %3s %6s %5s %3s %3s %4s %9s | %6s %8s %9s %8s" % ("kind", "B", "th", "tw", "nsteps", "cout", "ng", "var", "bn", "GFLOP", "n/step", "us avg", "ms/step", "TFLOP/s"))
acc = 0.0
for key, e in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    n, ms, fl = e
    acc += ms / N
    print("%-6s %2d %3d %3d %6d %5d %3d %3d %4d %9.2f | %6.1f %8.1f %9.3f %8.1f   cum %.1f" % (*key, n / N, ms / n * 1e3, ms / N, fl / ms / 1e9 if ms > 0 else 0, acc))
