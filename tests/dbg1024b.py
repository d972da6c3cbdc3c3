import os, sys, math, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ppst_oracle as O
from ppst_amd import ops
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(1, 128, S, S); w = torch.randn(128, 128, 3, 3) / 34.0
ref = F.conv2d(x, w, padding=1)
xn = x.permute(0, 2, 3, 1).contiguous().cuda()
plan = ops.ConvPlan(w.cuda())
y, st = plan(xn, stats=True)
print("conv raw", rel(y.permute(0, 3, 1, 2), ref))
print("stats sum", rel(st.double().sum(1)[..., 0].cpu(), ref.double().sum((2, 3))), "sumsq", rel(st.double().sum(1)[..., 1].cpu(), (ref.double() ** 2).sum((2, 3))))
ss = ops.in_finalize(st, S * S)
yn = ops.affine_act(y, ss)
print("IN", rel(yn.permute(0, 3, 1, 2), O.instance_norm(ref)))
ssin = torch.randn(1, 128, 2)
y2 = plan(xn, in_ss=ssin.cuda())
xa = x * ssin[:, :, 0, None, None] + ssin[:, :, 1, None, None]
print("conv in_ss", rel(y2.permute(0, 3, 1, 2), F.conv2d(xa, w, padding=1)))
lo = torch.randn(1, 128, S // 2, S // 2)
up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False)
y3 = ops.affine_act(xn, None, res=lo.permute(0, 2, 3, 1).contiguous().cuda(), out_scale=0.7, res_up2=True)
print("res_up2", rel(y3.permute(0, 3, 1, 2), (x + up) * 0.7))
y4, part = ops.affine_act_stats(xn, None, res=lo.permute(0, 2, 3, 1).contiguous().cuda(), out_scale=0.7, res_up2=True)
print("res_up2 stats", rel(y4.permute(0, 3, 1, 2), (x + up) * 0.7))
