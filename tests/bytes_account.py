"""Tuning aid (GPU): algorithmic bytes moved by the elementwise wrappers during one swap step,
to compare with the rocprof kernel times (effective GB/s per op family)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ppst_amd import glue, ops, weights as W
from ppst_amd.ppst_model import create_model

acc = collections.defaultdict(lambda: [0, 0, collections.Counter()])
def nbytes(t):
    return t.numel() * t.element_size() if isinstance(t, torch.Tensor) else 0
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        out = fn(*a, **k)
        b = sum(nbytes(t) for t in a) + sum(nbytes(t) for t in k.values())
        outs = out if isinstance(out, (tuple, list)) else (out,)
        b += sum(nbytes(t) for t in outs)
        acc[name][0] += 1; acc[name][1] += b
        shp = tuple(a[0].shape) if a and isinstance(a[0], torch.Tensor) else ()
        acc[name][2][(shp, "res" if k.get("res") is not None else "", "up2" if k.get("res_up2") else "")] += 1
        return out
    setattr(ops, name, w)
for n in ("affine_act", "affine_act_stats", "blur_nhwc", "in_stats", "bilinear", "avgpool", "gap_gmp", "conv1x1_small_cin", "conv1x1_small_cout"):
    if hasattr(ops, n):
        wrap(n)
# fused conv: algorithmic bytes = input channels actually read + output written (+ residual / noise planes)
conv_acc = [0, 0, 0]
_orig_call = ops.ConvPlan.__call__
def _conv_call(self, x, *a, **k):
    out = _orig_call(self, x, *a, **k)
    y = out[0] if isinstance(out, tuple) else out
    B_, H_, W_, _ = x.shape
    cin = self.cin * (4 if self.kind in ("s2d",) else 1)
    rd = B_ * H_ * W_ * cin * 4
    res = k.get("residual")
    if res is not None:
        rd += res.shape[0] * res.shape[1] * res.shape[2] * self.cout * 4
    if k.get("noise") is not None:
        rd += k["noise"].numel() * 4
    wr = y.shape[0] * y.shape[1] * y.shape[2] * self.cout * 4
    conv_acc[0] += 1; conv_acc[1] += rd; conv_acc[2] += wr
    return out
ops.ConvPlan.__call__ = _conv_call
B = 8
sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
model = create_model(state_dict=sd, device=torch.device("cuda"))
model.noise = {k: v.cuda() for k, v in W.make_noise(2, B).items()}
imgs = W.synthetic_images(4, 2 * B).cuda()
with torch.no_grad():
    bench.swap_step(model, imgs[:B].contiguous(), imgs[B:].contiguous(), 1.0, glue)
torch.cuda.synchronize()
for k, (n, b, shapes) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-20s calls %3d  %8.1f MB" % (k, n, b / 1e6))
    for s, c in shapes.most_common(12):
        print("      %3d x %s" % (c, s))
print("fused conv: %d launches, algorithmic read %.1f MB/launch, write %.1f MB/launch (PMC: profiles/r01_pmc_traffic.json conv_mfma)" % (
    conv_acc[0], conv_acc[1] / conv_acc[0] / 1e6, conv_acc[2] / conv_acc[0] / 1e6))
