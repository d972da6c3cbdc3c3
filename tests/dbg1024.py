import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ppst_oracle as O
from ppst_amd import ops, weights as W
from ppst_amd.ppst_model import create_model
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sd0 = W.make_state_dict(0, with_D=False, with_nce=False)
im = W.synthetic_images(13, 2, size=size)
g = lambda t: t.cuda()
with torch.no_grad():
    spr = O.encoder_con(sd0, im[0:1]); glr = O.encoder_col(sd0, im[1:2])[0]
    m0 = create_model(state_dict=sd0)
    sp0, _ = m0(g(im[0:1]), command="encode"); _, gl0 = m0(g(im[1:2]), command="encode")
    print("sp", rel(sp0, spr), [rel(a, b) for a, b in zip(gl0, glr)])
    outr = O.generator(sd0, spr, glr)
    out_o = m0(g(spr), [g(t) for t in glr], command="decode")
    print("decode with oracle inputs", rel(out_o, outr))
    # stage by stage inside G with oracle inputs: replicate oracle generator blocks
    import math, torch.nn.functional as F
    codes = O.normalize(list(glr)); gg = codes[-1]
    q = "G.SpatialCodeModulation."
    x = spr * O.equal_linear(gg, sd0[q + "scale.weight"], sd0[q + "scale.bias"])[:, :, None, None] + O.equal_linear(gg, sd0[q + "bias.weight"], sd0[q + "bias.bias"])[:, :, None, None]
    G = m0.G
    from ppst_amd.networks.base_network import to_nhwc, as_nchw, INV_SQRT2
    hc = [ops.l2norm_rows(g(c), 1e-8, 0) for c in glr]
    styles = G._style_table(hc)
    ws = G.p("SpatialCodeModulation.scale.weight"); inv = 1.0 / math.sqrt(ws.shape[1])
    hx = ops.spatial_modulation(to_nhwc(g(spr)).contiguous(), ops.linear(hc[-1], ws, G.p("SpatialCodeModulation.scale.bias"), wscale=inv),
                                ops.linear(hc[-1], G.p("SpatialCodeModulation.bias.weight"), G.p("SpatialCodeModulation.bias.bias"), wscale=inv))
    print("modulated sp", rel(as_nchw(hx), x))
    for i, (ci, co) in enumerate(O.G_HEAD_CH):
        qq = "G.HeadResnetBlock%d." % i
        skip = x if ci == co else O.conv_layer(x, sd0, qq + "skip.", ci, co, 1, activate=False, bias=False)
        r = O.styled_conv(x, sd0, qq + "conv1.", gg); r2 = O.styled_conv(r, sd0, qq + "conv2.", gg)
        x = (skip + r2) / O.SQRT2
        q2 = "HeadResnetBlock%d." % i
        hskip = hx if ci == co else G.plan(q2 + "skip.Conv.weight", scale=1.0 / math.sqrt(ci))(hx)
        hr, rss = G.styled_conv(hx, q2 + "conv1.", styles, "", None, defer=True)
        print("  head%d conv1 (applied)" % i, rel(as_nchw(ops.affine_act(hr, rss)), r))
        hx = G.styled_conv(hr, q2 + "conv2.", styles, "", None, res=hskip, out_scale=INV_SQRT2, in_ss=rss)
        print("head block %d" % i, rel(as_nchw(hx), x))
    for j, (key, ci, co) in enumerate(O.G_UP):
        qq = "G.UpsamplingResBlock%d." % key; gj = codes[-2 - j]
        skip = x if ci == co else O.conv_layer(x, sd0, qq + "skip.", ci, co, 1, activate=True, bias=True)
        skip = F.interpolate(skip, scale_factor=2, mode="bilinear", align_corners=False)
        r = O.styled_conv(x, sd0, qq + "conv1.", gj, upsample=True); r2 = O.styled_conv(r, sd0, qq + "conv2.", gj)
        x = (skip + r2) / O.SQRT2
        q2 = "UpsamplingResBlock%d." % key
        hskip = hx if ci == co else G.plan(q2 + "skip.Conv.weight", scale=1.0 / math.sqrt(ci))(hx, bias=G.p(q2 + "skip.Act.bias"), act=ops.ACT_LRELU)
        hr, rss = G.styled_conv(hx, q2 + "conv1.", styles, "", None, upsample=True, defer=True)
        print("  up%d conv1 (applied)" % key, rel(as_nchw(ops.affine_act(hr, rss)), r))
        hx = G.styled_conv(hr, q2 + "conv2.", styles, "", None, res=hskip, out_scale=INV_SQRT2, in_ss=rss, res_up2=True)
        print("up block %d" % key, rel(as_nchw(hx), x))
