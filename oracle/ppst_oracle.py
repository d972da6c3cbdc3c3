"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the PPST swap / train-step hot path.

This file is a from-scratch *functional* restatement (plain PyTorch CPU ops, no
nn.Module tree) of the algorithms on the path named by BASELINE.json
``north_star``.  It exists so that the HIP path in ``ppst_amd/`` can be checked
on a GPU box where /root/reference does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path never does (ppst_amd raises if its HIP library is
missing instead of falling back to anything in here).

Parity status: **pinned** against the reference's own Python (imported under the
shims of oracle/ref_loader.py in the build container) through the fixtures that
oracle/gen_golden.py produced from it (tests/golden/*.npz), which
tests/test_oracle_golden.py holds this file to.  Two third-party pieces are *unpinned*
(SURVEY.md section 8c): cv2.ximgproc.guidedFilter (opencv-contrib 4.8.1.78,
photo_gif.py:43) -- restated below from He et al. / the OpenCV contrib
algorithm -- and lpips (ppst_model.py:48,178), which is left out.

Every function cites the reference file:line it follows.  Parameters come from
a flat ``state_dict`` (name -> tensor) with the reference's key names.
All functions are dtype-generic: run them in float64 for a high-precision
reference or float32 to mimic the reference's CPU path bit-for-bit-ish.
"""
import math
import sys

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# --------------------------------------------------------------------------
# stylegan2_op: upfirdn2d / fused_leaky_relu
# --------------------------------------------------------------------------
def make_kernel(k, dtype=torch.float32):
    """stylegan2_layers.py:28-36 -- outer product of a 1-D tap list, sum 1."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    k = k / k.sum()
    return k.to(dtype)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d.py:150-159 -> upfirdn2d_kernel.cu:52-137 (same result as the
    pure-torch upfirdn2d_native, upfirdn2d.py:162-222).

    x (B,C,H,W); zero-insert upsample by ``up``, pad (negative = crop) with
    (pad0 before, pad1 after) on both axes, true 2-D convolution with
    ``kernel`` (i.e. correlation with the flipped kernel), keep every
    ``down``-th sample.  Accumulation order follows the CUDA kernel: y-major,
    x-minor over the taps (upfirdn2d_kernel.cu:124-128).
    """
    return upfirdn2d_full(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_full(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    if up_x > 1 or up_y > 1:
        z = x.new_zeros(B, C, H * up_y, W * up_x)
        z[:, :, ::up_y, ::up_x] = x
        x = z
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0),
          max(-px0, 0): x.shape[3] - max(-px1, 0)]
    oh = x.shape[2] - kh + 1
    ow = x.shape[3] - kw + 1
    kf = torch.flip(kernel, [0, 1]).to(x.dtype)
    out = x.new_zeros(B, C, oh, ow)
    for ky in range(kh):
        for kx in range(kw):
            out = out + x[:, :, ky:ky + oh, kx:kx + ow] * kf[ky, kx]
    return out[:, :, ::down_y, ::down_x].contiguous()


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=SQRT2):
    """fused_act.py:89-96 / fused_bias_act_kernel.cu:19-49 (act=3, grad=0):
    y = (x+b > 0 ? x+b : (x+b)*slope) * scale, bias broadcast over dim 1."""
    if bias is not None:
        x = x + bias.view(1, -1, *([1] * (x.dim() - 2)))
    return torch.where(x > 0, x, x * negative_slope) * scale


def fused_leaky_relu_grad(grad_out, out, negative_slope=0.2, scale=SQRT2):
    """fused_act.py:23-53 (act=3, grad=1): the gate is the sign of the saved
    *output*; returns (grad_input, grad_bias)."""
    gi = torch.where(out > 0, grad_out, grad_out * negative_slope) * scale
    dims = [0] + list(range(2, gi.dim()))
    return gi, gi.sum(dims)


# --------------------------------------------------------------------------
# layer zoo (stylegan2_layers.py)
# --------------------------------------------------------------------------
def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d defaults: no affine, biased variance, eps 1e-5."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def prelu(x, a):
    return torch.where(x >= 0, x, x * a.view(1, -1, 1, 1))


def equal_conv2d(x, w, b=None, stride=1, padding=0):
    """EqualConv2d, stylegan2_layers.py:167-196: runtime scale 1/sqrt(Cin*k*k)."""
    scale = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
    return F.conv2d(x, w * scale, b, stride=stride, padding=padding)


def equal_linear(x, w, b, lr_mul=1.0, activation=False):
    """EqualLinear, stylegan2_layers.py:205-242."""
    scale = (1.0 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), b * lr_mul)
    return F.linear(x, w * scale, b * lr_mul)


def conv_layer(x, sd, p, cin, cout, ks, downsample=False, blur_kernel=(1, 3, 3, 1),
               bias=True, activate=True, pad=None, norm="none", reflection_pad=False):
    """ConvLayer, stylegan2_layers.py:497-555 ([Blur]->[RefPad]->Conv->[IN]->[Act])."""
    w = sd[p + "Conv.weight"]
    assert tuple(w.shape) == (cout, cin, ks, ks), (p, w.shape)
    if downsample:
        if pad is None:
            pad = (len(blur_kernel) - 2) + (ks - 1)
        pad0, pad1 = (pad + 1) // 2, pad // 2
        k = sd[p + "Blur.kernel"]
        if reflection_pad:
            x = F.pad(x, (pad0, pad1, pad0, pad1), mode="reflect")
            x = upfirdn2d(x, k, pad=(0, 0))
        else:
            x = upfirdn2d(x, k, pad=(pad0, pad1))
        stride, padding = 2, 0
    else:
        stride = 1
        padding = ks // 2 if pad is None else pad
        if reflection_pad:
            if padding > 0:
                x = F.pad(x, (padding,) * 4, mode="reflect")
            padding = 0
    b = sd[p + "Conv.bias"] if (bias and not activate) else None
    x = equal_conv2d(x, w, b, stride=stride, padding=padding)
    if norm == "in":
        x = instance_norm(x)
    if activate:
        if bias:
            x = fused_leaky_relu(x, sd[p + "Act.bias"])
        else:
            x = fused_leaky_relu(x, None)
    return x


def res_block(x, sd, p, cin, cout, blur_kernel, reflection_pad=False, norm=None):
    """ResBlock, stylegan2_layers.py:559-579; skip never reflection-pads."""
    n = norm or "none"
    out = conv_layer(x, sd, p + "conv1.", cin, cin, 3, reflection_pad=reflection_pad, norm=n)
    out = conv_layer(out, sd, p + "conv2.", cin, cout, 3, downsample=True,
                     blur_kernel=blur_kernel, reflection_pad=reflection_pad, norm=n)
    skip = conv_layer(x, sd, p + "skip.", cin, cout, 1, downsample=True,
                      blur_kernel=blur_kernel, activate=False, bias=False, norm=n)
    return (out + skip) / SQRT2


def style_mod(x, style, w, b):
    """StyleMod + EqualizedLinear(gain 1, use_wscale), stylegan2_layers.py:249-273,361-374."""
    s = F.linear(style, w * (w.shape[1] ** -0.5), b)
    C = x.shape[1]
    s = s.view(-1, 2, C, 1, 1)
    return x * (s[:, 0] + 1.0) + s[:, 1]


def upscale_weight(w):
    """EqualizedConv2d fused-upscale weight, stylegan2_layers.py:312-319:
    (Cout,Cin,3,3) -> (Cin,Cout,4,4) = sum of the 4 one-pixel shifts of the
    zero-padded 5x5 kernel."""
    w = w.permute(1, 0, 2, 3)
    w = F.pad(w, [1, 1, 1, 1])
    return (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]).contiguous()


def styled_conv(x, sd, p, style, upsample=False, noise=None, use_noise=True):
    """StyledConv, stylegan2_layers.py:439-475 (conv -> noise -> bias ->
    FusedLeakyReLU -> InstanceNorm -> StyleMod) with EqualizedConv2d :275-348."""
    w = sd[p + "conv.weight"]
    cb = sd[p + "conv.bias"]
    if upsample and min(x.shape[2:]) * 2 >= 128:
        x = F.conv_transpose2d(x, upscale_weight(w), stride=2, padding=1)
        x = x + cb.view(1, -1, 1, 1)
    elif upsample:
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        x = F.conv2d(x, w, cb, padding=1)
    else:
        x = F.conv2d(x, w, cb, padding=1)
    if use_noise:
        nw = sd[p + "noise.weight"]
        if noise is None:
            assert float(nw.abs().max()) == 0.0, "explicit noise needed when noise.weight != 0"
        else:
            x = x + nw * noise
    x = x + sd[p + "bias"]
    x = fused_leaky_relu(x, sd[p + "activate.bias"])
    x = instance_norm(x)
    return style_mod(x, style, sd[p + "epi1.style_mod.lin.weight"], sd[p + "epi1.style_mod.lin.bias"])


def to_rgb(x, sd, p, style):
    """ToRGB with skip=None, stylegan2_layers.py:477-495."""
    x = equal_conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"])
    x = x + sd[p + "bias"]
    x = instance_norm(x)
    return style_mod(x, style, sd[p + "epi1.style_mod.lin.weight"], sd[p + "epi1.style_mod.lin.bias"])


# --------------------------------------------------------------------------
# util helpers on the path (util/util.py)
# --------------------------------------------------------------------------
def normalize(v):
    """util/util.py:18-22."""
    if isinstance(v, (list, tuple)):
        return [normalize(vv) for vv in v]
    return v * torch.rsqrt(torch.sum(v ** 2, dim=1, keepdim=True) + 1e-8)


def lerp(a, b, r):
    """util/util.py:32-35."""
    if isinstance(a, (list, tuple)):
        return [lerp(aa, bb, r) for aa, bb in zip(a, b)]
    return a * (1 - r) + b * r


def swap(x):
    """ppst_model.py:59-66 -- flip adjacent pairs of the minibatch."""
    shape = x.shape
    assert shape[0] % 2 == 0, "Minibatch size must be a multiple of 2"
    return torch.flip(x.view(shape[0] // 2, 2, *shape[1:]), [1]).reshape(shape)


def tensor2im(t):
    """util/util.py:98-131 for a (B,3,H,W) tensor with tile=False:
    ((x+1)/2*255) clipped to [0,255] then **truncated** to uint8, HWC."""
    a = t.detach().cpu().double().numpy() if t.dtype == torch.float64 else t.detach().cpu().numpy()
    a = (np.transpose(a, (0, 2, 3, 1)) + 1) / 2.0 * 255.0
    return np.clip(a, 0, 255).astype(np.uint8)


def to_pil_uint8(t):
    """simple_swapping_evaluator.py:61-62: ToPILImage()((x.clamp(-1,1)+1)*0.5)
    == byte = trunc(255*v) (torchvision mul(255).byte()), HWC uint8."""
    v = (t.clamp(-1.0, 1.0) + 1.0) * 0.5
    return (v * 255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()


def gan_loss(pred, real):
    """models/networks/loss.py:11-18 (LSGAN)."""
    return torch.mean((pred - 1) ** 2) if real else torch.mean(pred ** 2)


def one_hot_mask(labels, n=3):
    """CelebAMask_dataset.py:54-60: integer label map {0..n-1} (B,H,W) ->
    one-hot float (B,n,H,W)."""
    return torch.stack([(labels == i) for i in range(n)], dim=1).float()


# --------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------
def encoder_con(sd, x, p="E1."):
    """StyleGAN2ResnetEncodercon.forward, encoder_con.py:82-92."""
    x = conv_layer(x, sd, p + "FromRGB.", 3, 32, 1)
    ch = [32, 64, 128, 256]
    for i in range(3):
        x = res_block(x, sd, p + "DownToSpatialCode.ResBlockDownBy%d." % (2 ** i),
                      ch[i], ch[i + 1], (1, 2, 1), reflection_pad=True, norm="in")
    x = conv_layer(x, sd, p + "ToSpatialCode.0.", 256, 256, 1, activate=True, bias=True, norm="in")
    x = conv_layer(x, sd, p + "ToSpatialCode.1.", 256, 256, 1, activate=False, bias=True, norm="in")
    return x


def _e2_head(sd, p, tag, x):
    """encoder_col.py:162-168: cat(GAP,GMP) -> conv1x1 -> projector -> F.normalize."""
    gap = x.mean(dim=(2, 3))
    # nn.AdaptiveMaxPool2d(1) (encoder_col.py:45): its backward routes the gradient to the FIRST maximal element in
    # row-major order (amax would split it evenly over ties; the x8 bilinear upsampling of the warped features has 4x4
    # plateaus at the image border, so ties do occur)
    gmp = F.adaptive_max_pool2d(x, 1).flatten(1)
    v = torch.cat([gap, gmp], 1)
    w = sd[p + "conv1x1_%s.weight" % tag]
    v = F.linear(v, w.view(w.shape[0], -1), sd[p + "conv1x1_%s.bias" % tag])
    q = p + "projector%s." % tag
    v = F.linear(F.relu(v), sd[q + "1.weight"], sd[q + "1.bias"])
    v = F.linear(F.relu(v), sd[q + "3.weight"], sd[q + "3.bias"])
    v = F.linear(F.relu(v), sd[q + "5.weight"], sd[q + "5.bias"])
    return F.normalize(v)


def e2_warp(fea, corr, resize, scale_factor):
    """StyleGAN2ResnetEncodercol.warp, encoder_col.py:100-138 (square inputs)."""
    b, c, h, w = fea.shape
    assert h == w
    if resize:
        feas = F.adaptive_avg_pool2d(fea, (64, 64)).reshape(b, c, -1).permute(0, 2, 1)
        wf = torch.matmul(corr, feas).permute(0, 2, 1).reshape(b, c, 64, 64)
        return F.interpolate(wf, scale_factor=scale_factor, mode="bilinear")
    f = fea.reshape(b, c, -1).permute(0, 2, 1)
    return torch.matmul(corr, f).permute(0, 2, 1).reshape(b, c, 64, -1)


def encoder_col(sd, x, mask=None, corrmatrix=None, p="E2."):
    """StyleGAN2ResnetEncodercol.forward, encoder_col.py:150-251.
    Returns (vectors, vectors_w) or, with mask,
    (vectors, projections_m, vectors_w, projections_mw)."""
    vectors, vectors_w, pm, pmw = [], [], [], []
    x = conv_layer(x, sd, p + "FromRGB.", 3, 32, 1)
    ch = [32, 64, 128, 256]
    tags = ["9", "0", "1", "2"]
    scales = [8, 4, 2, None]
    for lvl in range(4):
        if lvl > 0:
            x = res_block(x, sd, p + "DownToGlobalCode1.ResBlockDownBy%d." % (2 ** (lvl - 1)),
                          ch[lvl - 1], ch[lvl], (1, 2, 1), reflection_pad=True)
        tag = tags[lvl]
        vectors.append(_e2_head(sd, p, tag, x))
        xx = None
        if corrmatrix is not None:
            # the first level warps with the live matrix (encoder_col.py:165), the deeper ones with corrmatrix.detach() (:197)
            xx = e2_warp(x, corrmatrix if lvl == 0 else corrmatrix.detach(), scales[lvl] is not None, scales[lvl])
            vectors_w.append(_e2_head(sd, p, tag, xx))
        if mask is not None:
            if lvl > 0:
                mask = F.max_pool2d(mask, 2, 2)
            for i in range(3):
                pm.append(_e2_head(sd, p, tag, x * mask[:, i:i + 1]))
                if corrmatrix is not None:
                    pmw.append(_e2_head(sd, p, tag, xx * swap(mask)[:, i:i + 1]))
    if mask is not None:
        return vectors, pm, vectors_w, pmw
    return vectors, vectors_w


def _feat_head(sd, p, x, k):
    """generator.py:174-224: layer32/64/128 (k=3, ReplicationPad before IN!) and
    layer256 (k=1).  Note the quirk: InstanceNorm runs on the *padded* tensor."""
    def pad(t):
        return F.pad(t, (1, 1, 1, 1), mode="replicate") if k == 3 else t
    x = instance_norm(pad(x))
    x = F.conv2d(x, sd[p + "2.weight"], sd[p + "2.bias"])
    x = prelu(instance_norm(x), sd[p + "4.weight"])
    x = F.conv2d(pad(x), sd[p + "6.weight"], sd[p + "6.bias"])
    return prelu(instance_norm(x), sd[p + "8.weight"])


def _residual_block(sd, p, x):
    """generator.py:10-32 (shared PReLU weight)."""
    a = sd[p + "prelu.weight"]
    out = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    out = prelu(instance_norm(out), a)
    out = F.conv2d(F.pad(out, (1, 1, 1, 1), mode="replicate"), sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    out = instance_norm(out) + x
    return prelu(out, a)


G_HEAD_CH = [(256, 256), (256, 256), (256, 384), (384, 512)]
G_UP = [(16, 512, 512), (32, 512, 256), (64, 256, 128)]
G_NOISE_SHAPES = None  # see generator_noise_shapes()


def generator_noise_shapes(B, S=64):
    """(name, (B,1,H,W)) for every NoiseInjection in call order (14 of them)."""
    out = []
    for i in range(4):
        for c in ("conv1", "conv2"):
            out.append(("HeadResnetBlock%d.%s" % (i, c), (B, 1, S, S)))
    s = S
    for key, _, _ in G_UP:
        s *= 2
        for c in ("conv1", "conv2"):
            out.append(("UpsamplingResBlock%d.%s" % (key, c), (B, 1, s, s)))
    return out


def generator(sd, sp, codes, extract_features=False, noise=None, p="G."):
    """StyleGAN2ResnetGenerator.forward, generator.py:244-281.
    ``noise``: optional dict name -> (B,1,H,W) (names from generator_noise_shapes)."""
    nz = (lambda k: None) if noise is None else (lambda k: noise.get(k))
    codes = normalize(list(codes))
    g = codes[-1]
    q = p + "SpatialCodeModulation."
    x = sp * equal_linear(g, sd[q + "scale.weight"], sd[q + "scale.bias"])[:, :, None, None] \
        + equal_linear(g, sd[q + "bias.weight"], sd[q + "bias.bias"])[:, :, None, None]
    for i, (ci, co) in enumerate(G_HEAD_CH):
        q = p + "HeadResnetBlock%d." % i
        skip = x if ci == co else conv_layer(x, sd, q + "skip.", ci, co, 1, activate=False, bias=False)
        r = styled_conv(x, sd, q + "conv1.", g, noise=nz("HeadResnetBlock%d.conv1" % i))
        r = styled_conv(r, sd, q + "conv2.", g, noise=nz("HeadResnetBlock%d.conv2" % i))
        x = (skip + r) / SQRT2
    feas = []
    if extract_features:
        feas.append(_feat_head(sd, p + "layer32.", x.detach(), 3))          # generator.py:256: the heads read x.detach()
    for j, (key, ci, co) in enumerate(G_UP):
        q = p + "UpsamplingResBlock%d." % key
        g = codes[-2 - j]
        skip = x if ci == co else conv_layer(x, sd, q + "skip.", ci, co, 1, activate=True, bias=True)
        skip = F.interpolate(skip, scale_factor=2, mode="bilinear", align_corners=False)
        r = styled_conv(x, sd, q + "conv1.", g, upsample=True, noise=nz("UpsamplingResBlock%d.conv1" % key))
        r = styled_conv(r, sd, q + "conv2.", g, noise=nz("UpsamplingResBlock%d.conv2" % key))
        x = (skip + r) / SQRT2
        if extract_features:
            feas.append(_feat_head(sd, p + "layer%d." % (2 ** (j + 6)), x.detach(), 3 if j < 2 else 1))   # generator.py:267
    rgb = to_rgb(x, sd, p + "ToRGB.", codes[0])
    if not extract_features:
        return rgb
    h, w = feas[0].shape[2:]
    feat = torch.cat([feas[0]] + [F.adaptive_avg_pool2d(f, (h, w)) for f in feas[1:]], dim=1)
    feat1 = torch.cat([F.interpolate(f, (256, 256), mode="bilinear") for f in feas], dim=1)
    for i in range(3):
        feat = _residual_block(sd, p + "layert.%d." % i, feat)
    feat1 = _residual_block(sd, p + "layert1.0.", feat1)
    feat1 = F.conv2d(feat1, sd[p + "layert1.1.weight"], sd[p + "layert1.1.bias"])
    return rgb, feat, feat1


D_CH = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}


def discriminator_block_names(size):
    """stylegan2_layers.py:608-611: block names ('512x512', then '1','2',...)."""
    log_size = int(math.log2(size))
    names = []
    for i in range(log_size, 2, -1):
        names.append(str(9 - i) if i <= 8 else "%dx%d" % (2 ** i, 2 ** i))
    return names


def discriminator(sd, x, size=512, p="D.stylegan2_D."):
    """StyleGAN2Discriminator / Discriminator, discriminator.py:19-21,
    stylegan2_layers.py:582-646."""
    c = D_CH[size]
    x = conv_layer(x, sd, p + "convs.0.", 3, c, 1)
    s = size
    for name in discriminator_block_names(size):
        co = D_CH[s // 2]
        x = res_block(x, sd, p + "convs.%s." % name, c, co, (1, 3, 3, 1))
        c, s = co, s // 2
    x = conv_layer(x, sd, p + "final_conv.", c, 512, 3)
    x = x.reshape(x.shape[0], -1)
    x = equal_linear(x, sd[p + "final_linear.0.weight"], sd[p + "final_linear.0.bias"], activation=True)
    return equal_linear(x, sd[p + "final_linear.1.weight"], sd[p + "final_linear.1.bias"])


# --------------------------------------------------------------------------
# correspondence (ppst_model.py:330-387)
# --------------------------------------------------------------------------
EPS64 = sys.float_info.epsilon


def rselfcorr(fea):
    """PPSTModel.Rselfcorr, ppst_model.py:330-339.  fea (B,64,256,256) ->
    (B,256,64,64): per 4x4 patch, centre over channels, L2-normalise over
    channels, 16x16 Gram over channels; out[b, i*16+j, py, px]."""
    B, C, H, W = fea.shape
    gy, gx = H // 4, W // 4
    x = fea.reshape(B, C, gy, 4, gx, 4).permute(0, 1, 2, 4, 3, 5).reshape(B, C, gy * gx, 16)
    x = x - x.mean(dim=1, keepdim=True)
    x = x / (torch.norm(x, 2, 1, keepdim=True) + EPS64)
    g = torch.einsum("bcpi,bcpj->bpij", x, x)
    return g.reshape(B, gy * gx, 256).permute(0, 2, 1).reshape(B, 256, gy, gx)


def corrm(fea, fea0, match_kernel=1):
    """PPSTModel.corrm, ppst_model.py:341-364.
    fea = style/key (B,512,h,w), fea0 = content/query; returns (B, hw_query, hw_key)
    = softmax over keys of cosine similarity / 0.01.  match_kernel k != 1 (:345-347): both maps are unfolded into
    k x k neighbourhoods first (F.unfold, zero padding k // 2: rows c * k^2 + ky * k + kx) -- the mean is then taken over
    the first 256 ROWS of the unfolded matrix, whatever channels / taps those are, exactly as the reference slices it."""
    def prep(f):
        if match_kernel == 1:
            f = f.reshape(f.shape[0], f.shape[1], -1)
        else:
            f = F.unfold(f, kernel_size=match_kernel, padding=int(match_kernel // 2))
        h1 = f[:, :256]
        h1 = h1 - h1.mean(dim=1, keepdim=True)
        f = torch.cat((h1, f[:, 256:]), dim=1)
        return f / (torch.norm(f, 2, 1, keepdim=True) + EPS64)
    k = prep(fea)
    q = prep(fea0).permute(0, 2, 1)
    return F.softmax(torch.matmul(q, k) / 0.01, dim=-1)


def model_warp(fea, corr):
    """PPSTModel.warp, ppst_model.py:366-387: patch-level soft warp."""
    b, c, h, w = fea.shape
    H = corr.shape[1]
    if H != h * w:
        s = int(((h * w) / H) ** 0.5)
        feas = F.unfold(fea, s, stride=s).permute(0, 2, 1)
        wf = torch.matmul(corr, feas).permute(0, 2, 1)
        return F.fold(wf, (h, w), s, stride=s)
    f = fea.reshape(b, c, -1).permute(0, 2, 1)
    return torch.matmul(corr, f).permute(0, 2, 1).reshape(b, c, h, w)


# --------------------------------------------------------------------------
# guided filter post-process (photo_gif.py:25-46 -> cv2.ximgproc.guidedFilter)
# PARITY UNPINNED: OpenCV is not available; this restates the colour-guide
# guided filter of He et al. as OpenCV-contrib implements it (float32 work
# type, normalised box filter of size 2r+1 with BORDER_REFLECT, eps added to the
# covariance diagonal, symmetric 3x3 inverse by cofactors, round-to-nearest
# saturate to uint8).
# --------------------------------------------------------------------------
def _box_mean(a, r):
    """a (H,W) float; mean over (2r+1)^2 window, border 'reflect' (edge pixel
    repeated: numpy 'symmetric' == cv2.BORDER_REFLECT)."""
    ap = np.pad(a, r, mode="symmetric")
    c = np.cumsum(np.cumsum(ap.astype(np.float64), axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    n = 2 * r + 1
    H, W = a.shape
    s = c[n:n + H, n:n + W] - c[0:H, n:n + W] - c[n:n + H, 0:W] + c[0:H, 0:W]
    return (s / (n * n)).astype(a.dtype)


def guided_filter_color(guide_u8, src_u8, r=30, eps=(0.02 * 255) ** 2, dtype=np.float32):
    """guide_u8, src_u8: (H,W,3) uint8 -> (H,W,3) uint8."""
    I = guide_u8.astype(dtype)
    P = src_u8.astype(dtype)
    mI = [_box_mean(I[..., i], r) for i in range(3)]
    cov = {}
    for i in range(3):
        for j in range(i, 3):
            cov[(i, j)] = _box_mean(I[..., i] * I[..., j], r) - mI[i] * mI[j]
            if i == j:
                cov[(i, j)] = cov[(i, j)] + dtype(eps)
    a00, a01, a02 = cov[(0, 0)], cov[(0, 1)], cov[(0, 2)]
    a11, a12, a22 = cov[(1, 1)], cov[(1, 2)], cov[(2, 2)]
    c00 = a11 * a22 - a12 * a12
    c01 = a02 * a12 - a01 * a22
    c02 = a01 * a12 - a02 * a11
    c11 = a00 * a22 - a02 * a02
    c12 = a02 * a01 - a00 * a12
    c22 = a00 * a11 - a01 * a01
    det = a00 * c00 + a01 * c01 + a02 * c02
    inv = [[c00 / det, c01 / det, c02 / det],
           [c01 / det, c11 / det, c12 / det],
           [c02 / det, c12 / det, c22 / det]]
    out = np.empty_like(P)
    for ch in range(3):
        p = P[..., ch]
        mp = _box_mean(p, r)
        cp = [_box_mean(I[..., i] * p, r) - mI[i] * mp for i in range(3)]
        a = [inv[k][0] * cp[0] + inv[k][1] * cp[1] + inv[k][2] * cp[2] for k in range(3)]
        b = mp - a[0] * mI[0] - a[1] * mI[1] - a[2] * mI[2]
        ma = [_box_mean(a[k], r) for k in range(3)]
        mb = _box_mean(b, r)
        out[..., ch] = ma[0] * I[..., 0] + ma[1] * I[..., 1] + ma[2] * I[..., 2] + mb
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def smooth(out, target, r=30, eps=(0.02 * 255) ** 2):
    """PPSTModel.decode post-process, ppst_model.py:290-305: tensor2im both,
    guided filter per image (guide = target/content), ToTensor, (x-0.5)*2."""
    o8 = tensor2im(out)
    t8 = tensor2im(target)
    res = torch.zeros_like(out)
    for i in range(o8.shape[0]):
        f = guided_filter_color(t8[i], o8[i], r, eps)
        res[i] = (torch.from_numpy(f).permute(2, 0, 1).to(out.dtype) / 255.0 - 0.5) * 2
    return res


# --------------------------------------------------------------------------
# model facade (models/base_model.py:114-123, models/ppst_model.py)
# --------------------------------------------------------------------------
class PPSTOracle:
    """``model(*args, command="encode")`` dispatch like BaseModel.forward."""

    def __init__(self, state_dict, size=512, noise=None):
        self.sd = state_dict
        self.size = size
        self.noise = noise

    def __call__(self, *args, command=None, **kw):
        if command is None:
            raise ValueError(command)
        method = getattr(self, command)
        assert callable(method)
        return method(*args, **kw)

    def encode(self, image):
        return encoder_con(self.sd, image), encoder_col(self.sd, image)[0]

    def encode2(self, image, corrmatrix):
        return encoder_col(self.sd, image, corrmatrix=corrmatrix)

    def extract_feat(self, sp, gl):
        return generator(self.sd, sp, gl, extract_features=True, noise=self.noise)

    def extract_feat_from_image(self, img):
        sp = encoder_con(self.sd, img)
        gl = encoder_col(self.sd, img)[0]
        _, fea, fea1 = generator(self.sd, sp, gl, extract_features=True, noise=self.noise)
        return fea, fea1

    def Rselfcorr(self, fea):
        return rselfcorr(fea)

    def corrm(self, fea, fea0):
        return corrm(fea, fea0)

    def warp(self, fea, corr):
        return model_warp(fea, corr)

    def decode(self, sp, gl, target=None):
        out = generator(self.sd, sp, gl, noise=self.noise)
        if target is not None:
            return smooth(out, target)
        return out

    def discriminate(self, x):
        return discriminator(self.sd, x, self.size)

    # -- the single-pair recipe of simple_swapping_evaluator.py:44-60 --------
    def simple_swap(self, content, style, alpha=1.0):
        sp, gl_c = self.encode(content)
        fea_c, fea_c1 = self.extract_feat_from_image(content)
        fea_s, fea_s1 = self.extract_feat_from_image(style)
        fea_c = torch.cat((fea_c, rselfcorr(fea_c1)), dim=1)
        fea_s = torch.cat((fea_s, rselfcorr(fea_s1)), dim=1)
        corr = corrm(fea_s, fea_c)
        _, gl_w = self.encode2(style, corr)
        code = lerp(gl_c, gl_w, alpha)
        out = self.decode(sp, code)
        return dict(sp=sp, gl=gl_c, fea_c=fea_c, fea_s=fea_s, corr=corr, gl_w=gl_w, out=out)
