"""GPU parity tests (pytest -m gpu): every HIP kernel and the assembled networks against
the CPU oracle (oracle/ppst_oracle.py) and the reference-generated golden fixtures, through
the C ABI (ctypes -> libppst_hip.so).  The comparison code lives in tests/gpu_diag.py so the
same checks can be run as a readable table; here they are assertions.

Tolerances (rel = max|a-b| / max|b|):
  integer / index / copy ops ............ 0 (bit exact)
  fp32 elementwise, FIR, fp32-MFMA GEMMs  <= 2e-6 (1e-5 for K=4096 accumulations)
  fused conv, bf16x3 (fp32-class) ....... <= 3e-5 per layer
  assembled networks / full swap recipe . <= 1e-3 (BASELINE.json north_star tolerance)
  correspondence ........................ row arg-max agreement > 99.5 %
  guided filter ......................... <= 1 uint8 LSB (parity unpinned vs OpenCV)
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(fn_name):
    import gpu_diag as D
    D.RES.clear()
    getattr(D, fn_name)()
    torch.cuda.synchronize()
    bad = [n for n, ok in D.RES if not ok]
    assert D.RES, "no checks ran"
    assert not bad, "failed: %s" % bad


def test_native_library_is_loaded():
    from ppst_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libppst_hip.so"
    maps = open("/proc/self/maps").read()
    assert "libppst_hip.so" in maps


@pytest.mark.parametrize("fn", ["t_upfirdn2d", "t_fused_act", "t_ops_half", "t_ops_f64", "t_layout_misc", "t_conv", "t_conv_wino", "t_conv_dual", "t_conv_up9", "t_conv_variants", "t_conv_variants_single_pass", "t_half_storage", "t_conv_k64", "t_conv1x1_stream", "t_conv_ksplit", "t_fuse_tail", "t_gmp_multi", "t_train_half", "t_norm_pool", "t_corr", "t_guided"])
def test_kernels_vs_oracle(fn):
    _run(fn)


def test_networks_and_swap_recipe_vs_oracle():
    """E1, E2, G (+feature heads), D, Rselfcorr, corrm, E2 warp, decode, guided filter at
    512x512 (stress weights: non-zero biases / noise) and the 256x256 encode/decode config."""
    _run("t_networks")


def test_grid_1024_and_mask_configs_vs_oracle():
    """BASELINE configs 3 (grid + guided filter, pair sharding) and 5 (1024x1024 encode/decode +
    guided filter, fp32-class and single-pass bf16) and the E2 mask heads."""
    _run("t_configs")


def test_reduced_precision_modes():
    """fp16 / bf16 single-pass convs (fp32 accumulate, instance-norm statistics and StyleMod in every mode) against the
    fp32 oracle.  Bars, written before the first measurement (relative RMS / max-norm of the output image):
      fp16 (11 significant bits): 1024^2 encode-decode 5e-3 / 3e-2;  512^2 full swap recipe 2e-2 / 1e-1
      bf16 ( 8 significant bits): 1024^2 encode-decode 5e-2 / 2e-1;  512^2 full swap recipe 1e-1 / 5e-1
    (the swap recipe adds the T = 0.01 correspondence softmax, which turns feature rounding into shifted matches; the
    correspondence GEMMs themselves stay exact fp32 in every mode)."""
    _run("t_precision")


def test_discriminator_train_step_vs_autograd_oracle():
    """LSGAN losses, all D parameter gradients (conv wgrad/dgrad, blur, lrelu, linear) and one Adam
    step against oracle/train_oracle.py (CPU autograd, pinned to the reference's own backward)."""
    _run("t_train_d")


def test_backward_kernels_vs_autograd():
    """conv dgrad / dgrad over the space-to-depth input / wgrad alone, 3e-5 against torch autograd."""
    _run("t_train_ops")


def test_lazy_r1_penalty_vs_double_backward_oracle():
    """compute_R1_loss + the second-order parameter gradients (one extra forward sweep on the HIP
    kernels) against CPU double backward."""
    _run("t_train_r1")


@pytest.mark.parametrize("size,precision", [(128, 0), (512, 0), (512, 2)])
def test_train_step_matches_reference_golden(size, precision):
    """The HIP D iteration + lazy R1 against tests/golden/train{128,512}.npz, which hold what the reference's own
    compute_image_discriminator_losses / compute_R1_loss + autograd produced (512x512, batch 2 = BASELINE configs[3]'s
    shape).  precision 2 = the exact-fp32 verification convs: tells rounding of the bf16 hi+lo split from defects."""
    import numpy as np
    from ppst_amd import ops, weights as W
    from ppst_amd.networks.discriminator import StyleGAN2Discriminator
    from ppst_amd.train import DiscriminatorTrainer
    from test_oracle_golden import sample_idx, train_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train%d.npz" % size))
    sd = W.make_state_dict(11, size=size, with_nce=False, bias_std=0.1)
    D = StyleGAN2Discriminator(None, size=size)
    D.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("D.")}, strict=True)
    ops.set_precision(precision)
    tr = DiscriminatorTrainer(D.cuda())
    real, rec, mix = (t.cuda() for t in train_inputs(size))
    worst = {}

    def check(prefix, tol_l2, tol_max):
        for name in tr.names:
            key = prefix + "D." + name
            got = tr.g(name).double().cpu().numpy()
            ref = g[key + ".samples"].astype(np.float64)
            scale = float(g[key + ".stats"][2])
            d = got[sample_idx(key, got.size)] - ref
            if scale == 0.0:
                assert np.abs(got).max() == 0.0, key
                continue
            w_ = worst.setdefault(prefix, [0.0, 0.0, ""])
            e_max, e_l2 = np.abs(d).max() / scale, np.linalg.norm(d) / (np.linalg.norm(ref) + 1e-30)
            if e_max > w_[0]:
                w_[0], w_[2] = e_max, key
            w_[1] = max(w_[1], e_l2)
            assert e_max <= tol_max, (key, e_max, scale)
            assert e_l2 <= tol_l2, (key, e_l2)

    try:
        losses = tr.losses_and_grads(real, rec, mix)
        for k in ("D_real", "D_rec", "D_mix"):
            assert abs(float(losses[k]) - float(g["loss." + k])) <= 1e-4 * max(1.0, abs(float(g["loss." + k]))), k
        check("dgrad.", 5e-3, 5e-3)
        r1 = tr.r1_losses_and_grads(real)
        assert np.allclose(r1["D_R1"].cpu().numpy(), g["loss.D_R1"], rtol=1e-3)
        # second-order (R1) gradients at the same 5e-3 bar as the first-order ones (measured, round 2: worst 1.4e-3 at
        # 512x512 with the production convs, 6.7e-4 with the exact-fp32 convs; round 1 had accepted 2e-2 / 1e-1 here)
        check("r1grad.", 5e-3, 5e-3)
    finally:
        ops.set_precision(0)
        print("D step %d precision %d: worst (max, l2, key) %s" % (size, precision, worst))


def test_generator_loss_values_match_reference_golden():
    """model(real, None, None, mask, command="compute_generator_losses") -- forward values of the generator
    iteration on the HIP path (E1/E2 incl. mask heads, G x4, correspondence, mask warp, rscl NCE with queue
    updates, D) -- against tests/golden/gloss512.npz, produced by the reference's own method."""
    import numpy as np
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import create_model
    from test_oracle_golden import gloss_inputs, sample_idx
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gloss512.npz"))
    sd = W.make_state_dict(13, bias_std=0.1, noise_weight=0.0)
    m = create_model(state_dict=sd, with_D=True, with_nce=True)
    real, mask = (t.cuda() for t in gloss_inputs())
    with torch.no_grad():
        losses, metrics = m(real, None, None, mask, command="compute_generator_losses")
    assert set(losses) == {k[5:] for k in g.files if k.startswith("loss.")}
    for k, v in list(losses.items()) + [("L1_dist", metrics["L1_dist"])]:
        ref = float(g[("metric." if k == "L1_dist" else "loss.") + k])
        tol = 5e-3 if "styleCont" in k else 1e-3      # NCE logits are divided by T = 0.07
        assert abs(float(v) - ref) <= tol * max(1.0, abs(ref)), (k, float(v), ref)
    for i in range(4):
        q = getattr(m.criterionNCE, "queue_data_A%d" % i).double().cpu().numpy().reshape(-1)
        name = "queue%d" % i
        d = q[sample_idx(name, q.size)] - g[name + ".samples"]
        assert np.abs(d).max() <= 1e-3 * float(g[name + ".stats"][2]), name
        assert int(getattr(m.criterionNCE, "queue_ptr_A%d" % i)) == int(g["queue_ptr%d" % i])
    # the other two train-step commands run and agree with the trainer path
    with torch.no_grad():
        dl, _, sp, gl = m(real, mask, command="compute_discriminator_losses")
        r1 = m(real, command="compute_R1_loss")
    assert set(dl) == {"D_real", "D_rec", "D_mix"} and all(torch.isfinite(v).all() for v in dl.values())
    assert tuple(sp.shape) == (2, 256, 64, 64) and len(gl) == 4 and r1["D_R1"].shape == (2,) and (r1["D_R1"] >= 0).all()


def test_swap_matches_reference_golden():
    """The HIP path against the fixtures produced by the *reference itself*
    (oracle/gen_golden.py): sampled activations of the full recipe."""
    import zlib
    import numpy as np
    from ppst_amd import weights as W
    from ppst_amd.evaluation import simple_swap
    from ppst_amd.ppst_model import create_model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "swap512.npz"))
    sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
    m = create_model(state_dict=sd, with_D=True)
    m.noise = {k: v.cuda() for k, v in W.make_noise(3, 1).items()}
    imgs = W.synthetic_images(5, 2).cuda()
    with torch.no_grad():
        out = simple_swap(m, imgs[0:1], imgs[1:2], alphas=(0.0, 0.7, 1.0))
        d = m.discriminate(imgs)
    for alpha, t in out.items():
        name = "out_a%.1f" % alpha
        a = t.cpu().contiguous().view(-1).double().numpy()
        rng = np.random.default_rng([99, zlib.crc32(name.encode())])
        idx = rng.integers(0, a.size, size=2048)
        scale = g[name + ".stats"][2]
        assert np.abs(a[idx] - g[name + ".samples"]).max() <= 1e-3 * scale, name
    assert np.abs(d.cpu().numpy() - g["D"]).max() <= 1e-3 * np.abs(g["D"]).max()


def test_bench_step_batch8_rows_match_oracle():
    """bench.py's exact ``swap_step`` at the benchmarked shape (batch 8, 512x512, distinct images and noise per
    row, non-zero noise weights): rows 0 and 7 against two CPU-oracle swaps at the 1e-3 bar.  Catches per-row
    indexing mistakes (noise rows, [B][C][2] scale/shift tables, statistics tiles) that a batch-1 test cannot see."""
    import ppst_oracle as O
    import bench
    from ppst_amd import glue, weights as W
    from ppst_amd.ppst_model import create_model
    B = 8
    sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    m = create_model(state_dict=sd)
    noise = W.make_noise(2, B)
    m.noise = {k: v.cuda() for k, v in noise.items()}
    imgs = W.synthetic_images(4, 2 * B)
    with torch.no_grad():
        out = bench.swap_step(m, imgs[:B].cuda().contiguous(), imgs[B:].cuda().contiguous(), 1.0, glue).cpu()
        for row in (0, 7):
            orc = O.PPSTOracle(sd, noise={k: v[row:row + 1] for k, v in noise.items()})
            ref = orc.simple_swap(imgs[row:row + 1], imgs[B + row:B + row + 1], alpha=1.0)["out"]
            err = float((out[row:row + 1] - ref).abs().max() / ref.abs().max())
            assert err < 1e-3, (row, err)
    # rows must differ from each other (distinct inputs): a wrong-row read would often go unnoticed otherwise
    assert float((out[0] - out[7]).abs().max()) > 1e-2


def test_checkpoint_save_load_then_swap_matches_golden(tmp_path):
    """SURVEY 8(f1) on the GPU: PPSTModel.save() -> a fresh model -> load() (the reference's file layout and key
    walk, base_model.py:33-112) -> the swap recipe still matches the reference-generated golden."""
    import numpy as np
    import zlib
    from ppst_amd import weights as W
    from ppst_amd.evaluation import simple_swap
    from ppst_amd.ppst_model import Options, PPSTModel, create_model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "swap512.npz"))
    sd = W.make_state_dict(1, bias_std=0.1, noise_weight=0.1)
    opt = Options(checkpoints_dir=str(tmp_path), name="rt", isTrain=False)
    m = create_model(opt, state_dict=sd, with_D=True)
    path = m.save(50000)
    assert os.path.basename(path) == "50k_checkpoint.pth" and os.path.islink(os.path.join(str(tmp_path), "rt", "latest_checkpoint.pth"))
    m2 = PPSTModel(Options(checkpoints_dir=str(tmp_path), name="rt", isTrain=False), with_D=False).cuda()
    assert m2.load(verbose=False)
    m2.noise = {k: v.cuda() for k, v in W.make_noise(3, 1).items()}
    imgs = W.synthetic_images(5, 2).cuda()
    with torch.no_grad():
        out = simple_swap(m2, imgs[0:1], imgs[1:2], alphas=(1.0,))
    name = "out_a1.0"
    a = out[1.0].cpu().contiguous().view(-1).double().numpy()
    idx = np.random.default_rng([99, zlib.crc32(name.encode())]).integers(0, a.size, size=2048)
    assert np.abs(a[idx] - g[name + ".samples"]).max() <= 1e-3 * g[name + ".stats"][2]


def test_image_preprocessing_is_pillow_bit_exact():
    """Device bicubic resample (two integer passes) + ToTensor/Normalize against the oracle
    (oracle/resize_oracle.py, itself pinned to Pillow) and, when Pillow is importable, Pillow itself at
    a full-size case (768x1024 -> short side 512 -> multiple of 16)."""
    import numpy as np
    import resize_oracle as R
    from ppst_amd import imageio
    rng = np.random.default_rng(11)
    for (h, w), (oh, ow) in (((37, 53), (21, 29)), ((64, 48), (128, 96)), ((50, 50), (50, 31)), ((31, 77), (64, 77)), ((9, 200), (16, 16))):
        a = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        a[0, :3, :3] = 255; a[1, -3:, -3:] = 0
        got = imageio.resize_bicubic_u8(torch.from_numpy(a).cuda(), oh, ow).cpu().numpy()
        for b in range(2):
            assert np.array_equal(got[b], R.resize_u8(a[b], oh, ow)), ((h, w), (oh, ow), b)
    a = rng.integers(0, 256, (2, 90, 130, 3), dtype=np.uint8)
    t = imageio.preprocess(torch.from_numpy(a).cuda(), 64).cpu().numpy()
    for b in range(2):
        assert np.array_equal(t[b], R.preprocess(a[b], 64))
    assert imageio.resize_bicubic_u8(torch.zeros((0, 8, 8, 3), dtype=torch.uint8, device="cuda"), 4, 4).shape == (0, 4, 4, 3)
    try:
        from PIL import Image
    except ImportError:
        return
    yy, xx = np.mgrid[0:768, 0:1024]
    big = np.stack([(xx * 255 // 1023), (yy * 255 // 767), ((xx * 7 + yy * 13) % 256)], -1).astype(np.uint8)
    big = (big.astype(np.int32) + rng.integers(-20, 21, big.shape)).clip(0, 255).astype(np.uint8)
    t = imageio.preprocess(torch.from_numpy(big[None]).cuda(), 512)[0].cpu().numpy()
    w1, h1 = R.scale_shortside_size(1024, 768, 512)
    im = Image.fromarray(big).resize((w1, h1), Image.BICUBIC)
    w2, h2 = R.make_power_2_size(w1, h1)
    if (w2, h2) != (w1, h1):
        im = im.resize((w2, h2), Image.BICUBIC)
    ref = (np.asarray(im).astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    assert t.shape == (3, h2, w2) and np.array_equal(t, ref.transpose(2, 0, 1))


def test_ragged_image_sizes_encode_decode():
    """Non-square images whose sides are multiples of 16 but not of the 16x16-pixel conv tile times 8
    (ragged tiles at every stage): E1 / E2 / G encode-decode against the oracle, 1e-3."""
    import ppst_oracle as O
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import create_model
    sd = W.make_state_dict(2, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.0)
    m = create_model(state_dict=sd)
    for (h, w) in ((272, 208), (384, 512)):
        torch.manual_seed(h)
        a, b = torch.rand(1, 3, h, w) * 2 - 1, torch.rand(1, 3, h, w) * 2 - 1
        with torch.no_grad():
            sp, _ = m(a.cuda(), command="encode")
            _, gl = m(b.cuda(), command="encode")
            out = m(sp, gl, command="decode").cpu()
            ref = O.generator(sd, O.encoder_con(sd, a), O.encoder_col(sd, b)[0])
        assert tuple(sp.shape) == (1, 256, h // 8, w // 8) and out.shape == ref.shape
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-3, (h, w)


def test_full_size_properties():
    """Size-independent properties at the BASELINE sizes (512x512, batch 8)."""
    from ppst_amd import ops
    from ppst_amd.stylegan2_op import upfirdn2d
    torch.manual_seed(0)
    x = torch.randn(8, 32, 512, 512, device="cuda")
    y = torch.randn_like(x)
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]], device="cuda") / 16
    # linearity of the FIR
    lhs = upfirdn2d(2.0 * x + y, k, pad=(1, 1))
    rhs = 2.0 * upfirdn2d(x, k, pad=(1, 1)) + upfirdn2d(y, k, pad=(1, 1))
    assert (lhs - rhs).abs().max().item() < 1e-4
    # a normalised blur preserves constants away from the border
    c = upfirdn2d(torch.ones(1, 1, 512, 512, device="cuda"), k, pad=(1, 1))
    assert torch.allclose(c[:, :, 1:-1, 1:-1], torch.ones_like(c[:, :, 1:-1, 1:-1]), atol=1e-6)
    # correspondence rows are probability distributions
    f = torch.randn(2, 4096, 512, device="cuda")
    q = ops.corr_prep(f, 256)
    corr = ops.softmax_rows_(ops.gemm_nt(q, q), 0.01)
    assert (corr.sum(-1) - 1).abs().max().item() < 1e-4
    # self-correspondence peaks on the diagonal
    assert (corr.argmax(-1) == torch.arange(4096, device="cuda")).float().mean().item() > 0.999
    # instance norm output has zero mean / unit variance per (b, c)
    a = ops.nchw_to_nhwc(x)
    z = ops.affine_act(a, ops.in_finalize(ops.in_stats(a), 512 * 512))
    assert z.mean((1, 2)).abs().max().item() < 1e-4
    assert (z.var((1, 2), unbiased=False) - 1).abs().max().item() < 1e-3


def test_grid_8x8_at_size_properties():
    """BASELINE configs[2] at the shape it names (content_style_grid_generation_evaluator.py:81-93: 8 x 8 folder at 512^2, 64 pairs,
    guided filter on) -- properties only, no oracle at this size: every pair is written exactly once across the ranks, the run
    sharded over 2 simulated ranks (image passes -> exchange -> pair passes) is BIT-EQUAL to one rank's, the filtered outputs are
    finite and inside [-1, 1] (the filter's output is a uint8 image mapped back), and distinct styles give distinct images."""
    from ppst_amd import weights as W
    from ppst_amd.evaluation import grid_exchange, grid_image_pass, grid_pair_pass, swapping_grid
    from ppst_amd.ppst_model import create_model
    sd = W.make_state_dict(1, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    m = create_model(state_dict=sd, device="cuda")
    m.noise = {k: v.cuda() for k, v in W.make_noise(3, 1).items()}      # one fixed row = every batch row
    cs, ss_ = W.synthetic_images(21, 8).cuda(), W.synthetic_images(22, 8).cuda()
    with torch.no_grad():
        one = swapping_grid(m, cs, ss_, rank=0, world=1, smooth=True)
        assert sorted(one) == [(i, j) for i in range(8) for j in range(8)]
        outs = [grid_image_pass(m, cs, ss_, r, 2) for r in range(2)]
        tc, ts = grid_exchange(None, None, 8, 8, 2, gathered=outs)
        got, owners = {}, {}
        for rank in range(2):
            part = grid_pair_pass(m, cs, ss_, tc, ts, rank, 2, smooth=True)
            for k_ in part:
                assert k_ not in owners, "pair %s written by ranks %d and %d" % (k_, owners[k_], rank)
                owners[k_] = rank
            got.update(part)
    assert sorted(got) == sorted(one) and sorted(owners.values()).count(0) == 32
    for k_, v in one.items():
        assert v.shape == (3, 512, 512)
        assert torch.isfinite(v).all() and v.min().item() >= -1.0 and v.max().item() <= 1.0
        assert torch.equal(v, got[k_]), "pair %s: 2 simulated ranks differ from 1 rank" % (k_,)
    assert not torch.equal(one[(0, 0)], one[(0, 1)]) and not torch.equal(one[(0, 0)], one[(1, 0)])


def test_edge_cases():
    from ppst_amd import ops
    from ppst_amd.stylegan2_op import fused_leaky_relu, upfirdn2d
    dev = "cuda"
    # empty batch
    assert upfirdn2d(torch.zeros(0, 3, 8, 8, device=dev), torch.ones(3, 3, device=dev)).shape == (0, 3, 6, 6)
    assert fused_leaky_relu(torch.zeros(0, 4, device=dev), torch.zeros(4, device=dev)).numel() == 0
    # ragged sizes: 1-pixel images, odd widths
    x = torch.randn(1, 2, 1, 7, device=dev)
    k = torch.randn(3, 3, device=dev)
    import ppst_oracle as O
    assert torch.allclose(upfirdn2d(x, k, pad=(1, 1)).cpu(), O.upfirdn2d(x.cpu(), k.cpu(), pad=(1, 1)), atol=1e-5)
    # batch-odd swap must fail like the reference
    from ppst_amd import glue
    with pytest.raises(AssertionError):
        glue.swap(torch.zeros(3, 2, device=dev))
    # integer input is rejected; the library takes float32, float16, bfloat16 (t_ops_half) and float64 (t_ops_f64) like the
    # reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF
    with pytest.raises(RuntimeError):
        upfirdn2d(torch.zeros(1, 1, 8, 8, device=dev, dtype=torch.int32), k)
    assert upfirdn2d(torch.zeros(1, 1, 8, 8, device=dev, dtype=torch.float16), k).dtype == torch.float16
    assert upfirdn2d(torch.zeros(1, 1, 8, 8, device=dev, dtype=torch.float64), k).dtype == torch.float64
    assert fused_leaky_relu(torch.zeros(2, 4, device=dev, dtype=torch.float64), torch.zeros(4, device=dev)).dtype == torch.float64


def test_get_visuals_for_snapshot_intended_semantics():
    """models/ppst_model.py:237-248 (the reference's body calls an undefined self.E; the intended semantics are restated in
    ppst_amd/ppst_model.py): rec == decode(encode(real)), mix == decode(sp, swap(gl)), layout = 3-component PCA picture of the spatial
    code at the image size in [-1, 1], at most 4 images while training."""
    from ppst_amd import glue
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import Options, create_model
    sd = W.make_state_dict(3, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    m = create_model(Options(isTrain=True), state_dict=sd, device="cuda")
    m.noise = {k: v.cuda() for k, v in W.make_noise(5, 1, S=16).items()}      # one fixed row = every batch row
    real = W.synthetic_images(9, 6, size=128).cuda()
    with torch.no_grad():
        vis = m(real, command="get_visuals_for_snapshot")
        assert set(vis) == {"real", "layout", "rec", "mix"}
        assert vis["real"].shape[0] == 4 and torch.equal(vis["real"], real[:4])
        sp, gl = m(real[:4], command="encode")
        rec = m(sp, gl, command="decode")
        mix = m(sp, [glue.swap(g) for g in gl], command="decode")
    assert torch.equal(vis["rec"], rec) and torch.equal(vis["mix"], mix)
    lay = vis["layout"]
    assert lay.shape == (4, 3, 128, 128) and lay.is_cuda and torch.isfinite(lay).all()
    assert -1.0 - 1e-5 <= lay.min().item() and lay.max().item() <= 1.0 + 1e-5 and lay.max().item() - lay.min().item() > 0.5
    m.opt.isTrain = False
    with torch.no_grad():
        assert m(real, command="get_visuals_for_snapshot")["rec"].shape[0] == 6


def test_corrm_match_kernel_vs_reference_golden():
    """``model(fea, fea0, command="corrm")`` with opt.match_kernel in {1, 3, 5} against the reference method's own output
    (tests/golden/corrm_mk.npz, ppst_model.py:341-364 with the F.unfold branch :345-347) on the fixture's seeded 16 x 16 maps:
    1e-4 of the largest probability (logits are cosines / 0.01; the kernel path measured 2e-6 at k = 1), rows sum to one;
    the F.unfold rows themselves bit-equal to F.unfold; an even kernel is refused."""
    import numpy as np
    import torch.nn.functional as F
    from test_oracle_golden import corrm_mk_inputs
    from ppst_amd import ops
    from ppst_amd.ppst_model import Options, create_model
    from ppst_amd import weights as W
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "corrm_mk.npz"))
    fea, fea0 = corrm_mk_inputs()
    x = torch.randn(2, 5, 7, 12)
    for k in (3, 5):
        rows = ops.unfold_rows(x.permute(0, 2, 3, 1).contiguous().cuda(), k)
        assert torch.equal(rows.cpu(), F.unfold(x, k, padding=k // 2).permute(0, 2, 1))
    sd = W.make_state_dict(0, with_D=False, with_nce=False)
    for k in (1, 3, 5):
        m = create_model(Options(match_kernel=k), state_dict=sd, device="cuda")
        with torch.no_grad():
            corr = m(fea.cuda(), fea0.cuda(), command="corrm")
        ref = torch.from_numpy(g["corr.k%d" % k])
        assert corr.shape == ref.shape
        err = (corr.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * ref.max().item(), (k, err)
        assert (corr.sum(-1) - 1).abs().max().item() < 1e-5
    m = create_model(Options(match_kernel=2), state_dict=sd, device="cuda")
    with pytest.raises(ValueError):
        m(fea.cuda(), fea0.cuda(), command="corrm")


def test_smooth_filter_local_affine_vs_oracle():
    """SURVEY section 8 f4 (smooth_filter.py:149-378): the two HIP launches against the numpy restatement of the three
    reference kernels -- per-pixel affine model, smoothed model and reconstructed image; r = 15 (LDS-tiled path), a small
    radius, and r = 17 (global-memory path); a ragged size that does not divide the 16 x 16 tile; batch of 2 == 2 singles.
    Bars (images in [0, 1]): model / filtered model 1e-6 relative to the largest coefficient (fp64 inverse of an
    fp32-product normal matrix; the oracle and the kernel order the cofactor sums differently), result 1e-6 absolute
    (first run measured 6e-8 / 8e-8 / 1.8e-7; the bars before it were 2e-5 / 2e-6) + agreement of the uint8 image up to
    1 LSB on <= 0.1 % of the samples.
    Property (no oracle): a stylised image that IS a global affine map of the content comes back unchanged up to the 1e-3
    regulariser."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import smooth_filter_oracle as SO
    from ppst_amd import smooth_filter as SF
    rng = np.random.default_rng(3)
    H, W = 45, 52
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = np.stack([0.5 + 0.4 * np.sin(xx / 7 + c) * np.cos(yy / 9 - c) for c in range(3)]).astype(np.float32)
    inp = np.clip(base + 0.05 * rng.standard_normal((3, H, W)).astype(np.float32), 0, 1).astype(np.float32)
    out = np.clip(0.8 * inp[::-1] ** 1.3 + 0.1 + 0.03 * rng.standard_normal((3, H, W)).astype(np.float32), 0, 1).astype(np.float32)
    dev = torch.device("cuda", 0)
    o_t, i_t = torch.from_numpy(out).to(dev), torch.from_numpy(inp).to(dev)
    for f_r, f_e in ((15, 0.1), (3, 0.05), (17, 0.1)):
        res, filt, model = SF.smooth_local_affine_tensor(o_t, i_t, 3, f_r, f_e, return_model=True)
        m_ref = SO.best_local_affine(out, inp, 1)
        f_ref = SO.bilateral_smooth(m_ref, inp, f_r, f_r / 3, f_e)
        r_ref = SO.reconstruction(inp, f_ref)
        em = np.abs(model[0].cpu().numpy() - m_ref).max() / np.abs(m_ref).max()
        ef = np.abs(filt.cpu().numpy() - f_ref).max() / np.abs(f_ref).max()
        er = np.abs(res.cpu().numpy() - r_ref).max()
        print("smooth_filter f_r=%d: model %.2e  filtered %.2e  result %.2e" % (f_r, em, ef, er))
        assert em < 1e-6 and ef < 1e-6 and er < 1e-6, (f_r, em, ef, er)
        u_hip = np.uint8(np.clip(res.cpu().numpy() * 255., 0, 255.))
        u_ref = np.uint8(np.clip(r_ref * 255., 0, 255.))
        d = np.abs(u_hip.astype(int) - u_ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() <= 1e-3
    # numpy front end of the reference's signature == tensor entry
    r_np = SF.smooth_local_affine(out, inp, 1e-7, 3, H, W, 15, 0.1)
    assert np.array_equal(r_np, SF.smooth_local_affine_tensor(o_t, i_t, 3, 15, 0.1).cpu().numpy())
    # batch == singles, bit for bit
    o2 = torch.stack([o_t, o_t.flip(2)]); i2 = torch.stack([i_t, i_t.flip(2)])
    rb = SF.smooth_local_affine_tensor(o2, i2, 3, 15, 0.1)
    assert torch.equal(rb[0], SF.smooth_local_affine_tensor(o_t, i_t, 3, 15, 0.1))
    assert torch.equal(rb[1], SF.smooth_local_affine_tensor(o_t.flip(2), i_t.flip(2), 3, 15, 0.1))
    # global affine map is a fixed point (channel order reversed by the kernels' indexing, smooth_filter.py:296-317)
    # (a textured content image: on smooth patches the 3x3 normal matrix is near-singular and the regulariser dominates)
    M = rng.standard_normal((3, 3)) * 0.2 + np.eye(3)
    tex = rng.random((3, H, W)).astype(np.float32)
    aff = (np.einsum("ij,jhw->ihw", M, tex) + 0.05).astype(np.float32)
    ra = SF.smooth_local_affine_tensor(torch.from_numpy(aff).to(dev), torch.from_numpy(tex).to(dev), 3, 15, 0.1).cpu().numpy()
    assert np.abs(ra - aff[::-1]).max() < 1e-2
    # full-size run (512 x 512, r = 15): finite, and a constant stylised image stays constant
    big_i = torch.rand(1, 3, 512, 512, device=dev)
    big_o = torch.full_like(big_i, 0.25)
    rbig = SF.smooth_local_affine_tensor(big_o, big_i, 3, 15, 0.1)
    assert torch.isfinite(rbig).all() and (rbig - 0.25).abs().max().item() < 1e-3
    # PIL front end (smooth_filter.py:381-405) == oracle on the same uint8 arrays up to 1 LSB
    from PIL import Image
    a8 = np.uint8(np.clip(out.transpose(1, 2, 0) * 255, 0, 255)); c8 = np.uint8(np.clip(inp.transpose(1, 2, 0) * 255, 0, 255))
    pil = np.asarray(SF.smooth_filter(Image.fromarray(a8), Image.fromarray(c8)))
    ref8 = SO.smooth_filter_arrays(a8, c8)
    d = np.abs(pil.astype(int) - ref8.astype(int))
    assert pil.shape == ref8.shape and d.max() <= 1 and (d > 0).mean() <= 1e-3


def test_evaluator_folder_front_ends(tmp_path):
    """The file-level front ends (evaluation.evaluate_grid_folder / evaluate_swap_files): folder layout and file names of the
    reference's evaluators, threaded PNG I/O; the files decode to exactly the uint8 images the tensor-level recipes give."""
    import numpy as np
    from PIL import Image
    from ppst_amd import evaluation as EV, glue, imageio, weights as W
    from ppst_amd.ppst_model import create_model
    dev = torch.device("cuda", 0)
    root = tmp_path / "data"
    (root / "content").mkdir(parents=True); (root / "style").mkdir()
    base = W.synthetic_images(21, 4, size=512)                       # (4,3,512,512) in [-1,1]
    u8 = ((base.clamp(-1, 1) + 1) * 127.5).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    names = [("content", "c0.png"), ("content", "c1.jpg"), ("style", "s0.png"), ("style", "s1.png")]
    for (sub, fn), a in zip(names, u8):
        Image.fromarray(a).resize((600, 600), Image.BICUBIC).save(str(root / sub / fn))       # not 512: exercises the device resize
        # (square: the correspondence path of the reference itself only closes at 512 x 512)
    sd = W.make_state_dict(3, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.0)
    model = create_model(state_dict=sd, device=dev)
    out_dir = tmp_path / "out"
    with torch.no_grad():
        written = EV.evaluate_grid_folder(model, str(root), str(out_dir), load_size=512, workers=4)
    files = sorted(os.listdir(str(out_dir / "images")))
    assert files == sorted(["c0.png", "c1.png", "s0.png", "s1.png", "c0_s0.png", "c0_s1.png", "c1_s0.png", "c1_s1.png"]), files
    assert len(written) == 8
    # the same pair through the tensor-level recipe
    cp = [str(root / "content" / "c0.png"), str(root / "content" / "c1.jpg")]
    sp_ = [str(root / "style" / "s0.png"), str(root / "style" / "s1.png")]
    with torch.no_grad():
        imgs = EV.load_images(cp + sp_, 512, dev)
        contents, styles = torch.cat(imgs[:2], 0), torch.cat(imgs[2:], 0)
        ref = EV.swapping_grid(model, contents, styles)
    got = np.asarray(Image.open(str(out_dir / "images" / "c1_s0.png")))
    want = glue.tensor2im(ref[(1, 0)][None])[0].cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(np.asarray(Image.open(str(out_dir / "images" / "c0.png"))), glue.tensor2im(contents[0:1])[0].cpu().numpy())
    # simple_swapping front end: names <structure>_<texture>_<alpha>.png
    with torch.no_grad():
        paths = EV.evaluate_swap_files(model, cp[0], sp_[1], str(tmp_path / "swap"), alphas=(0.5, 1.0))
    assert [os.path.basename(p) for p in paths] == ["c0_s1_0.50.png", "c0_s1_1.00.png"]
    with torch.no_grad():
        o = EV.simple_swap(model, imgs[0], imgs[3], (1.0,))[1.0]
    assert np.array_equal(np.asarray(Image.open(paths[1])), EV.to_uint8_image(o)[0].cpu().numpy())
