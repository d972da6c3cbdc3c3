"""Weight-format contract of the PPST path and a deterministic, name-keyed
random initialiser.

The flat ``state_dict`` key names and shapes are the reference's checkpoint
format (BaseModel.save/load, models/base_model.py:33-112; key list probed from
PPSTModel.state_dict()): prefixes ``E1. E2. G. D.
criterionNCE.`` + ``num_discriminator_iters``.  ``param_specs()`` enumerates
them from the architecture flags; tests pin the list against the reference and
against tests/golden/state_dict_keys.json.

``make_state_dict(seed)`` draws every tensor from a numpy Generator seeded by
(seed, crc32(name)) so any subset can be regenerated identically on a box that
has neither the reference nor a checkpoint (there is no network for
checkpoints: bench/smoke use these random-init weights).  Init rules follow
SURVEY.md Appendix A (file:line of each rule in the table below).
"""
import math
import zlib

import numpy as np
import torch

# init kinds ---------------------------------------------------------------
RANDN = "randn"              # EqualConv2d/EqualLinear/EqualizedLinear(use_wscale): N(0,1)  stylegan2_layers.py:173,211,262
HE = "he"                    # EqualizedConv2d in StyledConv: N(0,1)*sqrt2/sqrt(Cin*k*k)    stylegan2_layers.py:287-297
ZEROS = "zeros"              # every bias / noise weight                                    stylegan2_layers.py:182,215,381,463
N002 = "n002"                # init_net(..., 'normal', 0.02)                                encoder_col.py:90-93
CONV_DEFAULT = "conv_w"      # nn.Conv2d default: U(+-1/sqrt(fan_in))
CONV_DEFAULT_B = "conv_b"
PRELU = "prelu"              # nn.PReLU(): 0.25
BLUR3 = "blur3"              # make_kernel([1,2,1])                                        encoder_con.py:27
BLUR4 = "blur4"              # make_kernel([1,3,3,1])                                      discriminator.py:16
UP4 = "up4"                  # Upsample kernel = make_kernel([1,3,3,1])*4                  stylegan2_layers.py:39-46
QUEUE = "queue"              # rsclLoss queue: column-normalised randn                     rscl.py:24-31
ZERO_I64 = "zero_i64"


def make_kernel(k):
    """stylegan2_layers.py:28-36."""
    k = np.asarray(k, dtype=np.float32)
    k = k[None, :] * k[:, None]
    return (k / k.sum()).astype(np.float32)


def _conv_layer(p, cin, cout, ks, downsample=False, blur=None, bias=True, activate=True):
    """ConvLayer, stylegan2_layers.py:497-555."""
    out = []
    if downsample:
        out.append((p + "Blur.kernel", (len(blur), len(blur)), BLUR3 if len(blur) == 3 else BLUR4, None))
    out.append((p + "Conv.weight", (cout, cin, ks, ks), RANDN, None))
    if bias and not activate:
        out.append((p + "Conv.bias", (cout,), ZEROS, None))
    if activate and bias:
        out.append((p + "Act.bias", (cout,), ZEROS, None))
    return out


def _res_block(p, cin, cout, blur):
    """ResBlock, stylegan2_layers.py:559-579."""
    return (_conv_layer(p + "conv1.", cin, cin, 3)
            + _conv_layer(p + "conv2.", cin, cout, 3, downsample=True, blur=blur)
            + _conv_layer(p + "skip.", cin, cout, 1, downsample=True, blur=blur, bias=False, activate=False))


def _styled_conv(p, cin, cout, style_dim=2048):
    """StyledConv, stylegan2_layers.py:439-465 (registration order)."""
    return [
        (p + "bias", (1, cout, 1, 1), ZEROS, None),
        (p + "conv.weight", (cout, cin, 3, 3), HE, None),
        (p + "conv.bias", (cout,), ZEROS, None),
        (p + "epi1.style_mod.lin.weight", (2 * cout, style_dim), RANDN, None),
        (p + "epi1.style_mod.lin.bias", (2 * cout,), ZEROS, None),
        (p + "noise.weight", (1,), ZEROS, "noise"),
        (p + "activate.bias", (cout,), ZEROS, None),
    ]


def _nn_conv(p, cin, cout, ks):
    fan_in = cin * ks * ks
    return [(p + "weight", (cout, cin, ks, ks), CONV_DEFAULT, fan_in),
            (p + "bias", (cout,), CONV_DEFAULT_B, fan_in)]


def _feat_head(p, cin, ks):
    """generator.py:174-224."""
    return (_nn_conv(p + "2.", cin, 128 if ks == 3 else 64, ks)
            + [(p + "4.weight", (1,), PRELU, None)]
            + _nn_conv(p + "6.", 128 if ks == 3 else 64, 64, ks)
            + [(p + "8.weight", (1,), PRELU, None)])


def _residual_block(p, c=256):
    """generator.py:10-19."""
    return (_nn_conv(p + "conv1.", c, c, 3) + [(p + "prelu.weight", (1,), PRELU, None)]
            + _nn_conv(p + "conv2.", c, c, 3))


G_HEAD_CH = [(256, 256), (256, 256), (256, 384), (384, 512)]
G_UP = [(16, 512, 512), (32, 512, 256), (64, 256, 128)]
D_CH = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}


def discriminator_block_names(size):
    """stylegan2_layers.py:608-611."""
    log_size = int(round(math.log2(size)))
    return [str(9 - i) if i <= 8 else "%dx%d" % (2 ** i, 2 ** i) for i in range(log_size, 2, -1)]


def param_specs(size=512, with_D=True, with_nce=True):
    """Ordered [(name, shape, init_kind, aux)] == PPSTModel.state_dict() order."""
    s = [("num_discriminator_iters", (1,), ZERO_I64, None)]
    # ---- E1 (encoder_con.py:22-58) ----
    s += _conv_layer("E1.FromRGB.", 3, 32, 1)
    s += [("E1.mlp_01.0.weight", (256, 32), N002, None), ("E1.mlp_01.0.bias", (256,), ZEROS, None),
          ("E1.mlp_01.2.weight", (256, 256), N002, None), ("E1.mlp_01.2.bias", (256,), ZEROS, None)]
    ch = [32, 64, 128, 256]
    for i in range(3):
        s += _res_block("E1.DownToSpatialCode.ResBlockDownBy%d." % 2 ** i, ch[i], ch[i + 1], [1, 2, 1])
    s += _conv_layer("E1.ToSpatialCode.0.", 256, 256, 1, activate=True, bias=True)
    s += _conv_layer("E1.ToSpatialCode.1.", 256, 256, 1, activate=False, bias=True)
    # ---- E2 (encoder_col.py:22-93) ----
    s += _conv_layer("E2.FromRGB.", 3, 32, 1)
    for i in range(3):
        s += _res_block("E2.DownToGlobalCode1.ResBlockDownBy%d." % 2 ** i, ch[i], ch[i + 1], [1, 2, 1])
    s += [("E2.ToGlobalCode.0.weight", (2048, 256), RANDN, None), ("E2.ToGlobalCode.0.bias", (2048,), ZEROS, None)]
    for tag, c in zip("9012", ch):
        s += _nn_conv("E2.conv1x1_%s." % tag, 2 * c, c, 1)
    for tag, c in zip("9012", ch):
        q = "E2.projector%s." % tag
        s += [(q + "1.weight", (1024, c), N002, None), (q + "1.bias", (1024,), ZEROS, None),
              (q + "3.weight", (2048, 1024), N002, None), (q + "3.bias", (2048,), ZEROS, None),
              (q + "5.weight", (2048, 2048), N002, None), (q + "5.bias", (2048,), ZEROS, None)]
    # ---- G (generator.py:138-238) ----
    for n in ("scale", "bias"):
        s += [("G.SpatialCodeModulation.%s.weight" % n, (256, 2048), RANDN, None),
              ("G.SpatialCodeModulation.%s.bias" % n, (256,), ZEROS, None)]
    for i, (ci, co) in enumerate(G_HEAD_CH):
        q = "G.HeadResnetBlock%d." % i
        s += _styled_conv(q + "conv1.", ci, co) + _styled_conv(q + "conv2.", co, co)
        if ci != co:
            s += _conv_layer(q + "skip.", ci, co, 1, activate=False, bias=False)
    for key, ci, co in G_UP:
        q = "G.UpsamplingResBlock%d." % key
        s += _styled_conv(q + "conv1.", ci, co) + _styled_conv(q + "conv2.", co, co)
        if ci != co:
            s += _conv_layer(q + "skip.", ci, co, 1, activate=True, bias=True)
    s += [("G.ToRGB.bias", (1, 3, 1, 1), ZEROS, None), ("G.ToRGB.upsample.kernel", (4, 4), UP4, None),
          ("G.ToRGB.conv.weight", (3, 128, 1, 1), RANDN, None), ("G.ToRGB.conv.bias", (3,), ZEROS, None),
          ("G.ToRGB.epi1.style_mod.lin.weight", (6, 2048), RANDN, None),
          ("G.ToRGB.epi1.style_mod.lin.bias", (6,), ZEROS, None)]
    s += _feat_head("G.layer32.", 512, 3) + _feat_head("G.layer64.", 512, 3)
    s += _feat_head("G.layer128.", 256, 3) + _feat_head("G.layer256.", 128, 1)
    for i in range(3):
        s += _residual_block("G.layert.%d." % i)
    s += _residual_block("G.layert1.0.") + _nn_conv("G.layert1.1.", 256, 64, 1)
    # ---- D (stylegan2_layers.py:582-626) ----
    if with_D:
        c = D_CH[size]
        s += _conv_layer("D.stylegan2_D.convs.0.", 3, c, 1)
        sz = size
        for name in discriminator_block_names(size):
            co = D_CH[sz // 2]
            s += _res_block("D.stylegan2_D.convs.%s." % name, c, co, [1, 3, 3, 1])
            c, sz = co, sz // 2
        s += _conv_layer("D.stylegan2_D.final_conv.", c, 512, 3)
        s += [("D.stylegan2_D.final_linear.0.weight", (512, 512 * 16), RANDN, None),
              ("D.stylegan2_D.final_linear.0.bias", (512,), ZEROS, None),
              ("D.stylegan2_D.final_linear.1.weight", (1, 512), RANDN, None),
              ("D.stylegan2_D.final_linear.1.bias", (1,), ZEROS, None)]
    if with_nce:
        for i in range(4):
            s += [("criterionNCE.queue_data_A%d" % i, (2048, 128), QUEUE, None),
                  ("criterionNCE.queue_ptr_A%d" % i, (1,), ZERO_I64, None)]
    return s


def _draw(name, shape, kind, aux, seed, bias_std, noise_weight):
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if kind == RANDN:
        a = rng.standard_normal(shape, dtype=np.float32)
    elif kind == HE:
        a = rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0) / math.sqrt(shape[1] * shape[2] * shape[3]))
    elif kind == ZEROS:
        if aux == "noise":
            a = np.full(shape, noise_weight, dtype=np.float32)
        elif bias_std > 0:
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(bias_std)
        else:
            a = np.zeros(shape, dtype=np.float32)
    elif kind == N002:
        a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
    elif kind in (CONV_DEFAULT, CONV_DEFAULT_B):
        b = 1.0 / math.sqrt(aux)
        a = rng.uniform(-b, b, size=shape).astype(np.float32)
    elif kind == PRELU:
        a = np.full(shape, 0.25, dtype=np.float32)
    elif kind == BLUR3:
        a = make_kernel([1, 2, 1])
    elif kind == BLUR4:
        a = make_kernel([1, 3, 3, 1])
    elif kind == UP4:
        a = make_kernel([1, 3, 3, 1]) * 4
    elif kind == QUEUE:
        a = rng.standard_normal(shape, dtype=np.float32)
        a = a / np.maximum(np.sqrt((a * a).sum(0, keepdims=True)), 1e-12)
    elif kind == ZERO_I64:
        return torch.zeros(shape, dtype=torch.int64)
    else:
        raise KeyError(kind)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_state_dict(seed=0, size=512, with_D=True, with_nce=True, bias_std=0.0,
                    noise_weight=0.0, prefixes=None):
    """Deterministic random-init weights keyed by state-dict name.

    bias_std / noise_weight: the reference initialises every bias and noise
    weight to 0 (which hides indexing bugs); tests use non-zero values, which a
    trained checkpoint also has.
    """
    sd = {}
    for name, shape, kind, aux in param_specs(size, with_D, with_nce):
        if prefixes is not None and not name.startswith(tuple(prefixes)):
            continue
        sd[name] = _draw(name, shape, kind, aux, seed, bias_std, noise_weight)
    return sd


def make_noise(seed, B, S=64):
    """Explicit noise tensors for the 14 NoiseInjection layers
    (stylegan2_layers.py:376-399), keyed '<block>.<conv>' -> (B,1,H,W)."""
    out = {}
    names = []
    for i in range(4):
        for c in ("conv1", "conv2"):
            names.append(("HeadResnetBlock%d.%s" % (i, c), S))
    s = S
    for key, _, _ in G_UP:
        s *= 2
        for c in ("conv1", "conv2"):
            names.append(("UpsamplingResBlock%d.%s" % (key, c), s))
    for name, hw in names:
        rng = np.random.default_rng([seed, zlib.crc32(("noise." + name).encode())])
        out[name] = torch.from_numpy(rng.standard_normal((B, 1, hw, hw), dtype=np.float32))
    return out


def synthetic_images(seed, B, size=512, smooth=True):
    """Synthetic 'portrait' batch in [-1,1]: a smooth low-frequency field (so the
    T=0.01 correspondence softmax is not degenerate) plus uniform noise
    (SURVEY.md section 8d 'Synthetic inputs per config')."""
    rng = np.random.default_rng([seed, 7])
    img = rng.uniform(-1, 1, size=(B, 3, size, size)).astype(np.float32)
    if smooth:
        yy, xx = np.meshgrid(np.linspace(-1, 1, size, dtype=np.float32),
                             np.linspace(-1, 1, size, dtype=np.float32), indexing="ij")
        for b in range(B):
            for c in range(3):
                f = rng.uniform(0.5, 3.0, size=4).astype(np.float32)
                ph = rng.uniform(0, 6.28, size=2).astype(np.float32)
                base = 0.5 * np.sin(f[0] * xx * 3 + ph[0]) * np.cos(f[1] * yy * 3 + ph[1]) \
                    + 0.3 * np.exp(-((xx * f[2]) ** 2 + (yy * f[3]) ** 2))
                img[b, c] = np.clip(0.75 * base + 0.25 * img[b, c], -1, 1)
    return torch.from_numpy(img)
