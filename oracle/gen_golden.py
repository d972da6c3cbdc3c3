"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/* from the *reference*.

Run in the build container (needs /root/reference):  python oracle/gen_golden.py
Imports the reference's own Python under the shims of ref_loader.py, loads the
deterministic name-keyed weights of ppst_amd/weights.py into it, runs the
reference's commands and stores *data only* (inputs are regenerated from seeds;
outputs are stored as sampled values + summary statistics so the fixtures stay
small).  The reference never travels to the GPU box; these fixtures do.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_loader  # noqa: E402
from ppst_amd import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NSAMP = 2048


def sample_idx(name, numel, n=NSAMP):
    import zlib
    rng = np.random.default_rng([99, zlib.crc32(name.encode())])
    return rng.integers(0, numel, size=min(n, numel))


def pack(out, name, t):
    """Store sampled values + stats of tensor t under key prefix name."""
    a = t.detach().cpu().contiguous().view(-1).double().numpy()
    idx = sample_idx(name, a.size)
    out[name + ".shape"] = np.array(t.shape, dtype=np.int64)
    out[name + ".samples"] = a[idx].astype(np.float32)
    out[name + ".stats"] = np.array([a.mean(), a.std(), np.abs(a).max()], dtype=np.float64)


def gen_ops(ns):
    """Per-op goldens from the reference's own fallbacks: upfirdn2d for every
    pad/kernel combination the path uses (+ the up/down modes of the CUDA
    dispatcher, upfirdn2d_kernel.cu:177-211), fused_leaky_relu fwd and the
    autograd of its fallback."""
    out = {}
    rng = np.random.default_rng(11)
    up = ns.stylegan2_op.upfirdn2d
    cases = []
    k3 = ns.layers.make_kernel([1, 2, 1])
    k4 = ns.layers.make_kernel([1, 3, 3, 1])
    k2 = ns.layers.make_kernel([1, 1])
    ka = torch.tensor(rng.standard_normal((4, 4)), dtype=torch.float32)  # asymmetric: catches a missing flip
    kb = torch.tensor(rng.standard_normal((3, 3)), dtype=torch.float32)
    # (kernel, up, down, pad0, pad1, H, W)
    for (k, u, d, p0, p1, H, Wd) in [
        (k3, 1, 1, 0, 0, 35, 35), (k3, 1, 1, 1, 0, 32, 32), (k4, 1, 1, 2, 2, 32, 32),
        (k4, 1, 1, 1, 1, 32, 32), (ka, 1, 1, 2, 1, 19, 70), (kb, 1, 1, 0, 2, 17, 33),
        (ka, 2, 1, 2, 1, 16, 24), (k2 * 4, 2, 1, 1, 0, 9, 11), (ka, 1, 2, 1, 1, 21, 30),
        (k2, 1, 2, 0, 0, 16, 16), (ka, 1, 1, -1, -1, 20, 20), (k4 * 4, 2, 1, 2, 1, 8, 8),
    ]:
        x = torch.tensor(rng.standard_normal((2, 3, H, Wd)), dtype=torch.float32)
        y = up(x, k, up=u, down=d, pad=(p0, p1))
        i = len(cases)
        out["upfirdn2d.%d.x" % i] = x.numpy()
        out["upfirdn2d.%d.k" % i] = k.numpy()
        out["upfirdn2d.%d.cfg" % i] = np.array([u, d, p0, p1], dtype=np.int64)
        out["upfirdn2d.%d.y" % i] = y.numpy()
        cases.append(i)
    out["upfirdn2d.n"] = np.array(len(cases))
    # fused leaky relu (fallback path of fused_act.py:89-96) fwd + grads
    x = torch.tensor(rng.standard_normal((2, 5, 7, 6)), dtype=torch.float32, requires_grad=True)
    b = torch.tensor(rng.standard_normal((5,)), dtype=torch.float32, requires_grad=True)
    y = ns.stylegan2_op.fused_leaky_relu(x, b)
    g = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float32)
    gx, gb = torch.autograd.grad(y, [x, b], g)
    out.update({"flrelu.x": x.detach().numpy(), "flrelu.b": b.detach().numpy(), "flrelu.y": y.detach().numpy(),
                "flrelu.g": g.numpy(), "flrelu.gx": gx.numpy(), "flrelu.gb": gb.numpy()})
    x2 = torch.tensor(rng.standard_normal((3, 8)), dtype=torch.float32)
    b2 = torch.tensor(rng.standard_normal((8,)), dtype=torch.float32)
    out.update({"flrelu2.x": x2.numpy(), "flrelu2.b": b2.numpy(),
                "flrelu2.y": ns.stylegan2_op.fused_leaky_relu(x2, b2, 0.1, 1.5).numpy()})
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops.npz", len(out))


def gen_ops_f64(ns):
    """The two native ops in DOUBLE (the CUDA dispatcher of the reference is AT_DISPATCH_FLOATING_TYPES_AND_HALF,
    upfirdn2d_kernel.cu:225 / fused_bias_act_kernel.cu:79: a gradcheck-style caller hands them float64): the reference's own
    fallbacks on float64 tensors -- upfirdn2d for the two blur forms of the path, an asymmetric kernel and one up / one down
    mode; fused_leaky_relu forward and the autograd of its fallback."""
    out = {}
    rng = np.random.default_rng(12)
    up = ns.stylegan2_op.upfirdn2d
    k3 = ns.layers.make_kernel([1, 2, 1]).double()
    k4 = ns.layers.make_kernel([1, 3, 3, 1]).double()
    ka = torch.tensor(rng.standard_normal((4, 4)), dtype=torch.float64)
    n = 0
    for (k, u, d, p0, p1, H, Wd) in [(k3, 1, 1, 1, 0, 24, 24), (k4, 1, 1, 2, 2, 20, 28), (ka, 1, 1, 2, 1, 19, 33),
                                    (ka, 2, 1, 2, 1, 12, 10), (ka, 1, 2, 1, 1, 21, 18)]:
        x = torch.tensor(rng.standard_normal((2, 3, H, Wd)), dtype=torch.float64)
        y = up(x, k, up=u, down=d, pad=(p0, p1))
        assert y.dtype == torch.float64
        out["upfirdn2d.%d.x" % n] = x.numpy()
        out["upfirdn2d.%d.k" % n] = k.numpy()
        out["upfirdn2d.%d.cfg" % n] = np.array([u, d, p0, p1], dtype=np.int64)
        out["upfirdn2d.%d.y" % n] = y.numpy()
        n += 1
    out["upfirdn2d.n"] = np.array(n)
    x = torch.tensor(rng.standard_normal((2, 5, 7, 6)), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(rng.standard_normal((5,)), dtype=torch.float64, requires_grad=True)
    y = ns.stylegan2_op.fused_leaky_relu(x, b)
    assert y.dtype == torch.float64
    g = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
    gx, gb = torch.autograd.grad(y, [x, b], g)
    out.update({"flrelu.x": x.detach().numpy(), "flrelu.b": b.detach().numpy(), "flrelu.y": y.detach().numpy(),
                "flrelu.g": g.numpy(), "flrelu.gx": gx.numpy(), "flrelu.gb": gb.numpy()})
    np.savez_compressed(os.path.join(GOLD, "ops_f64.npz"), **out)
    print("ops_f64.npz", len(out))


def set_noise(m, noise):
    for name, mod in m.G.named_modules():
        if type(mod).__name__ == "NoiseInjection":
            mod.fixed_noise = None if noise is None else noise[name[:-len(".noise")]]


def gen_swap(ns, m):
    """simple_swapping recipe at 512^2 (simple_swapping_evaluator.py:44-60, with
    the intended tensor corr matrix) + D forward, stress weights (non-zero
    biases and noise weights, explicit noise)."""
    out = {}
    sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
    m.load_state_dict(sd, strict=True)
    noise = W.make_noise(3, 1)
    set_noise(m, noise)
    imgs = W.synthetic_images(5, 2)
    c, s = imgs[0:1], imgs[1:2]
    with torch.no_grad():
        sp, gl = m(c, command="encode")
        fc, fc1 = m(c, command="extract_feat_from_image")
        fs, fs1 = m(s, command="extract_feat_from_image")
        rc = m(fc1, command="Rselfcorr")
        rs = m(fs1, command="Rselfcorr")
        fcc = torch.cat((fc, rc), 1)
        fss = torch.cat((fs, rs), 1)
        corr = m(fss, fcc, command="corrm")
        _, glw = m(s, corr, command="encode2")
        for alpha in (0.0, 0.7, 1.0):
            code = ns.util.lerp(gl, glw, alpha)
            o = m(sp, code, target=None, command="decode")
            pack(out, "out_a%.1f" % alpha, o)
            if alpha == 1.0:
                u8 = ((o[0].clamp(-1.0, 1.0) + 1.0) * 0.5 * 255).to(torch.uint8)  # ToPILImage quantisation
                out["out_u8.hist"] = np.bincount(u8.flatten().numpy(), minlength=256).astype(np.int64)
        d = m.D(imgs)
        wimg = m(c, corr, command="warp")
    pack(out, "sp", sp)
    for i in range(4):
        out["gl%d" % i] = gl[i].numpy()
        out["glw%d" % i] = glw[i].numpy()
    pack(out, "fea_c", fc)
    pack(out, "fea_c1", fc1)
    pack(out, "rself_c", rc)
    pack(out, "fea_s", fs)
    pack(out, "rself_s", rs)
    pack(out, "corr", corr)
    out["corr.argmax"] = corr[0].argmax(-1).numpy().astype(np.int32)
    out["corr.rowmax"] = corr[0].max(-1)[0].numpy()
    rows = sample_idx("corr.rows", 4096, 16)
    out["corr.rows.idx"] = rows
    out["corr.rows.val"] = corr[0, rows].numpy().astype(np.float32)
    pack(out, "warp_img", wimg)
    out["D"] = d.numpy()
    np.savez_compressed(os.path.join(GOLD, "swap512.npz"), **out)
    print("swap512.npz", len(out))


def gen_cfg1(ns, m):
    """BASELINE config 1: 256^2 encode/decode only (the correspondence path is
    impossible at 256^2 in the reference), init-like weights
    (zero biases / noise weights => deterministic without explicit noise)."""
    out = {}
    sd = W.make_state_dict(0)
    m.load_state_dict(sd, strict=True)
    set_noise(m, None)
    imgs = W.synthetic_images(0, 2, size=256, smooth=False)
    with torch.no_grad():
        sp, _ = m(imgs[0:1], command="encode")
        _, gl = m(imgs[1:2], command="encode")
        o = m(sp, gl, target=None, command="decode")
    pack(out, "sp", sp)
    for i in range(4):
        out["gl%d" % i] = gl[i].numpy()
    pack(out, "out", o)
    np.savez_compressed(os.path.join(GOLD, "cfg1_256.npz"), **out)
    print("cfg1_256.npz", len(out))


def gen_glue(ns, m):
    """Exact / integer glue: tensor2im truncation
    (util/util.py:98-131), swap permutation (ppst_model.py:59-66), lerp and
    normalize (util/util.py:18-35)."""
    out = {}
    rng = np.random.default_rng(21)
    x = torch.tensor(rng.uniform(-1.3, 1.3, size=(4, 3, 16, 16)), dtype=torch.float32)
    out["t2i.x"] = x.numpy()
    out["t2i.y"] = ns.util.tensor2im(x, tile=False)
    out["swap.y"] = m.swap(x).numpy()
    a = torch.tensor(rng.standard_normal((2, 64)), dtype=torch.float32)
    b = torch.tensor(rng.standard_normal((2, 64)), dtype=torch.float32)
    out["vec.a"], out["vec.b"] = a.numpy(), b.numpy()
    out["lerp.y"] = ns.util.lerp([a], [b], 0.3)[0].numpy()
    out["normalize.y"] = ns.util.normalize(a).numpy()
    out["gan.real"] = ns.loss.gan_loss(a, True).numpy()
    out["gan.fake"] = ns.loss.gan_loss(a, False).numpy()
    np.savez_compressed(os.path.join(GOLD, "glue.npz"), **out)
    print("glue.npz", len(out))


def gen_keys(m):
    sd = m.state_dict()
    keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    print("state_dict_keys.json", len(keys))


def train_inputs(size=128, B=2):
    """Seeded inputs of the train-step fixture (shared with tests/test_oracle_golden.py)."""
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    rec = torch.rand(B // 2, 3, size, size, generator=g) * 2 - 1
    mix = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    return real, rec, mix


def gen_train(ns, size=128):
    """Discriminator iteration + lazy R1 from the reference's own methods and autograd
    (models/ppst_model.py:68-92, 140-159; optimizers/ppst_optimizer.py:112-125) on the reference's
    own Discriminator at crop_size 128: loss values, every parameter gradient (sampled)."""
    opt = ref_loader.default_opt(crop_size=size, load_size=size)
    D = ns.ppst_model.networks.create_network(opt, opt.netD, "discriminator")
    sd = W.make_state_dict(11, size=size, with_nce=False, bias_std=0.1)
    D.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("D.")}, strict=True)
    D.train()

    class Host:
        pass
    h = Host()
    h.D, h.opt = D, opt
    PM = ns.ppst_model.PPSTModel
    real, rec, mix = train_inputs(size)
    out = {}
    losses = PM.compute_image_discriminator_losses(h, real, rec, mix, None)
    for k, v in losses.items():
        out["loss." + k] = v.detach().numpy().astype(np.float64)
    D.zero_grad()
    sum(v.mean() for v in losses.values()).backward()
    for n, p in D.named_parameters():
        pack(out, "dgrad.D." + n, p.grad)
    D.zero_grad()
    r1 = PM.compute_R1_loss(h, real.clone())
    out["loss.D_R1"] = r1["D_R1"].detach().numpy().astype(np.float64)
    (sum(v.mean() for v in r1.values()) * opt.R1_once_every).backward()
    for n, p in D.named_parameters():
        pack(out, "r1grad.D." + n, p.grad if p.grad is not None else torch.zeros_like(p))
    np.savez_compressed(os.path.join(GOLD, "train%d.npz" % size), **out)
    print("train%d.npz" % size, len(out), {k: v for k, v in out.items() if k.startswith("loss.")})


def gloss_inputs(B=2, size=512):
    """Seeded inputs of the generator-loss fixture: smooth synthetic portraits + an integer label map -> one-hot mask."""
    real = W.synthetic_images(21, B, size)
    g = torch.Generator().manual_seed(77)
    lab = torch.randint(0, 3, (B, size // 16, size // 16), generator=g)
    lab = lab.repeat_interleave(16, 1).repeat_interleave(16, 2)
    mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous()
    return real, mask


def gen_gloss(ns):
    """compute_generator_losses of the reference itself (models/ppst_model.py:161-235) at B = 2, 512x512,
    lambda_Cycwarp = 0 (lpips stubbed), noise weights 0 (the reference's init): every loss / metric value
    and the NCE queues after the step's enqueues."""
    opt = ref_loader.default_opt(lambda_Cycwarp=0.0)
    m = ref_loader.build_reference_model(opt)
    sd = W.make_state_dict(13, bias_std=0.1, noise_weight=0.0)
    own = m.state_dict()
    m.load_state_dict({k: sd[k] for k in own if k in sd}, strict=False)
    m.eval()
    real, mask = gloss_inputs()
    with torch.no_grad():
        losses, metrics = m.compute_generator_losses(real, None, None, mask)
    out = {"loss." + k: np.array(float(v.mean())) for k, v in losses.items()}
    out.update({"metric." + k: np.array(float(v.mean())) for k, v in metrics.items()})
    for i in range(4):
        q = getattr(m.criterionNCE, "queue_data_A%d" % i)
        pack(out, "queue%d" % i, q)
        out["queue_ptr%d" % i] = np.array(int(getattr(m.criterionNCE, "queue_ptr_A%d" % i)))
    np.savez_compressed(os.path.join(GOLD, "gloss512.npz"), **out)
    print("gloss512.npz", {k: float(v) for k, v in out.items() if k.startswith(("loss.", "metric."))})


def gstep_inputs(B=2, size=512):
    """Seeded inputs of the generator-update fixture: the gloss inputs + explicit noise (weights.make_noise)."""
    real, mask = gloss_inputs(B, size)
    return real, mask, W.make_noise(31, B, S=size // 8)


def _noise_hooks(m, noise):
    """NoiseInjection draws normal_() on every call (stylegan2_layers.py:388-390); a forward pre-hook installs the
    fixture's rows [0:batch] as ``fixed_noise`` before each call (the reference file is not edited)."""
    hs = []
    for name, mod in m.G.named_modules():
        if type(mod).__name__ == "NoiseInjection":
            key = name[:-len(".noise")]

            def pre(mod_, args, key=key):
                mod_.fixed_noise = noise[key][:args[0].shape[0]]
            hs.append(mod.register_forward_pre_hook(pre))
    return hs


def _noise_scale_hooks(m, acc):
    """The noise-weight gradient is  sum_p noise[p] * (sum_c dL/dz[c, p])  -- a sum of ~10^5..10^6 zero-mean terms whose
    total can come out far smaller than its own random-walk magnitude (UpsamplingResBlock64.conv2 at stage 2: 5e-4 against
    1e-3), so "relative to the reference's value" is not a meaningful scale for it: a handful of leaky-ReLU gates that fall
    the other way within float32 rounding move it by tens of percent.  This hook records the natural scale
    sqrt(sum_p term[p]^2) of every NoiseInjection over all its calls of the step; the parity bar for these scalars is taken
    relative to max(|gradient|, that scale) (tests/gstep_diag.py)."""
    hs = []
    for name, mod in m.G.named_modules():
        if type(mod).__name__ == "NoiseInjection":
            def fwd(mod_, args, out, name=name):
                if not (torch.is_tensor(out) and out.requires_grad):
                    return
                nz = mod_.fixed_noise.detach()

                def on_grad(g, nz=nz, name=name):
                    t = (g.detach().sum(1, keepdim=True) * nz).double()
                    acc[name] = acc.get(name, 0.0) + float((t * t).sum())
                out.register_hook(on_grad)
            hs.append(mod.register_forward_hook(fwd))
    return hs


def gen_gstep(ns, stage, f64=False):
    """The generator/encoder update of the reference itself: PPSTOptimizer.train_generator_one_step's
    ``sum(v.mean()).backward()`` (optimizers/ppst_optimizer.py:73-94) over compute_generator_losses
    (models/ppst_model.py:161-235) at B = 2, 512x512, lambda_Cycwarp = 0 (lpips stubbed), stress weights (non-zero
    biases and noise weights, explicit noise).  stage 1: L1 + GAN on the reconstruction (lambda_StyleCon = 0);
    stage 2: the full objective.  f64: the same reference code with the model and inputs cast to float64 -- the
    rounding-free value of every gradient, which tells how much of a float32-vs-float32 difference is the reference's own
    summation noise (several of these gradients are heavily cancelling sums over 10^5..10^8 terms).  Stored: every loss, sampled gradients of every G / E1 / E2 parameter, and the
    gradients at the network boundaries (d/d rec, d/d sp, d/d codes) as debugging checkpoints."""
    over = dict(lambda_Cycwarp=0.0, training_stage=stage)
    if stage == 1:
        over["lambda_StyleCon"] = 0.0
    opt = ref_loader.default_opt(**over)
    m = ref_loader.build_reference_model(opt)
    sd = W.make_state_dict(17, bias_std=0.1, noise_weight=0.1)
    own = m.state_dict()
    m.load_state_dict({k: sd[k] for k in own if k in sd}, strict=False)
    m.train()
    real, mask, noise = gstep_inputs()
    if f64:
        m.double()
        real, mask, noise = real.double(), mask.double(), {k: v.double() for k, v in noise.items()}
    hooks = _noise_hooks(m, noise)
    nscale = {}
    hooks += _noise_scale_hooks(m, nscale)
    taps = {"E1": [], "E2": [], "G": []}

    def tap(tag):
        def f(mod, args, out):
            ts = []

            def walk(o):
                if torch.is_tensor(o):
                    if o.requires_grad:
                        o.retain_grad()
                    ts.append(o)
                elif isinstance(o, (list, tuple)):
                    for v in o:
                        walk(v)
            walk(out)
            taps[tag].append(ts)
        return f
    for tag in taps:
        hooks.append(getattr(m, tag).register_forward_hook(tap(tag)))
    for p in m.D.parameters():                      # set_requires_grad(self.Dparams, False)
        p.requires_grad_(False)
    for net in (m.G, m.E1, m.E2):
        net.zero_grad()
    losses, metrics = m.compute_generator_losses(real, None, None, mask)
    sum(v.mean() for v in losses.values()).backward()
    out = {"loss." + k: np.array(float(v.mean())) for k, v in losses.items()}
    out.update({"metric." + k: np.array(float(v.mean())) for k, v in metrics.items()})
    for pre, net in (("G.", m.G), ("E1.", m.E1), ("E2.", m.E2)):
        for n, p in net.named_parameters():
            pack(out, "grad." + pre + n, p.grad if p.grad is not None else torch.zeros_like(p))
    for tag, calls in taps.items():
        for ci, ts in enumerate(calls):
            for ti, t in enumerate(ts):
                if t.grad is not None:
                    pack(out, "tapgrad.%s.%d.%d" % (tag, ci, ti), t.grad)
    for name, v in nscale.items():
        out["nscale.G." + name + ".weight"] = np.array(np.sqrt(v))
    for h in hooks:
        h.remove()
    tag = "gstep512_s%d%s.npz" % (stage, "_f64" if f64 else "")
    np.savez_compressed(os.path.join(GOLD, tag), **out)
    print(tag, len(out), {k: float(v) for k, v in out.items() if k.startswith(("loss.", "metric."))})


def corrm_mk_inputs(h=16, seed=31):
    """Seeded (key, query) feature maps of the match_kernel fixture (shared with the tests): a common direction plus noise, so
    the T = 0.01 softmax stays soft (random 512-vectors would give one-hot rows and pin nothing but the arg-max)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 512, 1, 1, generator=g)
    fea = base + 0.5 * torch.randn(1, 512, h, h, generator=g)
    fea0 = base + 0.5 * torch.randn(1, 512, h, h, generator=g)
    return fea, fea0


def gen_corrm_mk(ns):
    """PPSTModel.corrm with match_kernel = 3 / 5 (ppst_model.py:341-364, the F.unfold branch :345-347) on a 16 x 16 map: the
    reference method itself, called on a stub that carries only ``opt.match_kernel`` (it touches nothing else of the model)."""
    from types import SimpleNamespace
    fea, fea0 = corrm_mk_inputs()
    out = {}
    for k in (1, 3, 5):
        corr = ns.ppst_model.PPSTModel.corrm(SimpleNamespace(opt=SimpleNamespace(match_kernel=k)), fea, fea0)
        out["corr.k%d" % k] = corr.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "corrm_mk.npz"), **out)
    print("corrm_mk.npz", {k: v.shape for k, v in out.items()}, "max prob", {k: float(v.max()) for k, v in out.items()})


def gen_iter_counter():
    """Schedule of the reference's own IterationCounter (util/iter_counter.py, loaded from its file: util/__init__ needs
    packages that are absent): for two option sets, the image count and the save / evaluate / print decisions of the first
    iterations."""
    import importlib.util
    from argparse import Namespace
    spec = importlib.util.spec_from_file_location("ref_iter_counter", os.path.join(ref_loader.REF, "util", "iter_counter.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = []
    for kw in (dict(batch_size=2, save_freq=50, evaluation_freq=30, print_freq=8, total_nimgs=200),
               dict(batch_size=16, save_freq=50000, evaluation_freq=50000, print_freq=480, total_nimgs=120000)):
        opt = Namespace(checkpoints_dir="/nonexistent", name="x", dataset_mode="celebamask", isTrain=True, continue_train=False,
                        resume_iter="latest", pretrained_name=None, display_freq=1600, **kw)
        ic = mod.IterationCounter(opt)
        ic.record_one_iteration.__func__  # noqa: B018 (exists)
        rows = []
        for _ in range(400):
            rows.append([int(ic.steps_so_far), bool(ic.needs_saving()), bool(ic.needs_evaluation()), bool(ic.needs_printing()),
                         bool(ic.completed_training())])
            if ic.completed_training():
                break
            ic.steps_so_far += ic.batch_size            # record_one_iteration without the file write
        out.append({"opt": kw, "rows": rows})
    with open(os.path.join(GOLD, "iter_counter.json"), "w") as f:
        json.dump(out, f)
    print("iter_counter.json", [len(o["rows"]) for o in out])


def main():
    os.makedirs(GOLD, exist_ok=True)
    ns = ref_loader.load_reference()
    if len(sys.argv) > 1:                        # e.g. `gen_golden.py gstep1 gstep2 train512`: only these fixtures
        for a in sys.argv[1:]:
            {"gstep1": lambda: gen_gstep(ns, 1), "gstep2": lambda: gen_gstep(ns, 2),
             "gstep1_f64": lambda: gen_gstep(ns, 1, True), "gstep2_f64": lambda: gen_gstep(ns, 2, True),
             "iter_counter": gen_iter_counter, "corrm_mk": lambda: gen_corrm_mk(ns),
             "ops_f64": lambda: gen_ops_f64(ns), "train512": lambda: gen_train(ns, 512), "train128": lambda: gen_train(ns), "gloss": lambda: gen_gloss(ns)}[a]()
        return
    m = ref_loader.build_reference_model()
    gen_keys(m)
    gen_glue(ns, m)
    gen_ops(ns)
    gen_ops_f64(ns)
    gen_cfg1(ns, m)
    gen_swap(ns, m)
    gen_train(ns)
    gen_train(ns, 512)
    gen_gloss(ns)
    gen_gstep(ns, 1)
    gen_gstep(ns, 2)
    gen_gstep(ns, 1, True)
    # gen_gstep(ns, 2, True) needs more than the build container's 62 GB (killed by the kernel twice): not a fixture
    gen_iter_counter()
    gen_corrm_mk(ns)


if __name__ == "__main__":
    main()
