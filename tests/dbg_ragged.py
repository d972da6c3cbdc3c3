"""Debug aid (GPU): encode/decode at a non-square size vs the CPU oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ppst_oracle as O
from ppst_amd import weights as W
from ppst_amd.ppst_model import create_model
sd = W.make_state_dict(2, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.0)
m = create_model(state_dict=sd)
for (h, w) in ((384, 512), (272, 208), (512, 512)):
    torch.manual_seed(h)
    a, b = torch.rand(1, 3, h, w) * 2 - 1, torch.rand(1, 3, h, w) * 2 - 1
    with torch.no_grad():
        sp, _ = m(a.cuda(), command="encode")
        _, gl = m(b.cuda(), command="encode")
        out = m(sp, gl, command="decode").cpu()
        spr = O.encoder_con(sd, a)
        glr, _ = O.encoder_col(sd, b)
        ref = O.generator(sd, spr, glr)
    print((h, w), "sp", tuple(sp.shape), "out", tuple(out.shape), "rel err %.3e" % float((out - ref).abs().max() / ref.abs().max()), flush=True)
