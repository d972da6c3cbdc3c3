"""Debug aid: where does d sum(D(x))/dx differ from the CPU autograd oracle?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ppst_oracle as O
from ppst_amd import weights as W
from ppst_amd.networks.discriminator import StyleGAN2Discriminator
from ppst_amd.train import DiscriminatorTrainer
dev = torch.device("cuda")
size, B = 128, 2
sd = W.make_state_dict(3, size=size, with_nce=False, bias_std=0.1)
D = StyleGAN2Discriminator(None, size=size)
D.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("D.")}, strict=True)
D = D.to(dev)
tr = DiscriminatorTrainer(D)
torch.manual_seed(size + 1)
real = torch.rand(B, 3, size, size) * 2 - 1
x = real.clone().requires_grad_(True)
pred = O.discriminator(sd, x, size).sum()
gref, = torch.autograd.grad(pred, [x])
pr, tape = tr.forward(real.to(dev))
keep = {}
tr.backward(tape, torch.ones_like(pr), param_grads=False, keep=keep)
got = keep["d_img"].permute(0, 3, 1, 2).cpu()
d = (got - gref).abs()
print("max ref", float(gref.abs().max()), "max err", float(d.max()), "mean err", float(d.mean()))
idx = torch.nonzero(d > 0.1 * d.max())
print("n large", idx.shape[0]); print(idx[:20].tolist())
rows = d.amax(dim=(0, 1, 3)); cols = d.amax(dim=(0, 1, 2))
print("row err (first 6, mid, last 6):", rows[:6].tolist(), float(rows[size // 2]), rows[-6:].tolist())
print("col err:", cols[:6].tolist(), float(cols[size // 2]), cols[-6:].tolist())
# gate-flip hypothesis: count sign mismatches of the first-layer activation
import math
w = sd["D.stylegan2_D.convs.0.Conv.weight"]; b = sd["D.stylegan2_D.convs.0.Act.bias"]
x0r = torch.nn.functional.conv2d(real, w / math.sqrt(3)) + b.view(1, -1, 1, 1)
x0g = tape["x0"].permute(0, 3, 1, 2).cpu()
mism = ((x0r > 0) != (x0g > 0))
print("layer-0 activations", x0r.numel(), "sign mismatches", int(mism.sum()), "max |pre| at mismatch", float(x0r[mism].abs().max()) if mism.any() else 0.0)
print("L2 rel err of d_img", float((got - gref).norm() / gref.norm()))
