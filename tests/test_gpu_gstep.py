"""GPU parity of the generator / encoder update (BASELINE configs[3]) -- pytest -m gpu.

Oracle: the REFERENCE's own ``compute_generator_losses`` + ``sum(v.mean()).backward()`` (oracle/gen_golden.py:gen_gstep)
at B = 2, 512x512, run twice: in float32 (what the reference computes) and in float64 (the rounding-free value).
Fixtures: tests/golden/gstep512_s{1,2}.npz and ..._f64.npz -- every loss, sampled entries of every G / E1 / E2 gradient.

Bars (error of a gradient tensor = max|d| / max|truth| and ||d||_2 / ||truth||_2 over the sampled entries, truth = the
float64 run; floor = the same two numbers for the reference's float32 run against its float64 run):
  * losses: 1e-3 (5e-3 for the NCE terms, logits / 0.07);
  * exact-fp32 convs (ops.set_precision(2), conv_f32.hip): every tensor within max(5e-3, 2 x floor) in l2; in max-norm the
    same, except the parameters whose gradient is a sum over 10^5..10^8 *gated* terms that cancel to 10^-3..10^-5 of their
    magnitude (a bias that feeds a leaky-ReLU in front of an instance norm, a noise weight, a PReLU slope, the first
    layers behind the global max pooling): one leaky-ReLU gate or arg-max that falls the other way within float32
    rounding moves such a sum by O(condition number / #terms) -- the reference's own float32 run is off by up to 3e-2
    there.  Bar for that class: 5e-2 max-norm (scalars) / 3e-2.
  * production convs (bf16 hi+lo split, 16 mantissa bits per operand): 1.5e-2 in l2, 5e-2 in max-norm, 1.5e-1 for the
    scalar cancelling sums -- the rounding of the split amplified by the same cancellation, bounded here so that a defect
    (which shows as O(1)) cannot hide.
Parameters whose exact gradient is zero (a bias in front of an instance norm) must come out at rounding-noise size.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run_blocks(fn):
    import gstep_diag as D
    D.RES.clear()
    getattr(D, fn)()
    torch.cuda.synchronize()
    bad = [n for n, ok in D.RES if not ok]
    assert D.RES and not bad, "failed: %s" % bad


def test_autograd_blocks_vs_torch_autograd():
    """Every differentiable block (conv / transposed conv / blur+stride-2 conv with each padding mode, instance norm +
    StyleMod / bias / PReLU, resize, pooling, padding, linear, normalise, modulation, FromRGB / ToRGB, L1 / LSGAN):
    forward and every input gradient against torch autograd of the oracle's ops in float64."""
    _run_blocks("t_blocks")


def test_correspondence_and_nce_blocks_vs_torch_autograd():
    _run_blocks("t_blocks2")


def test_networks_backward_vs_oracle_autograd():
    """E2 (masked heads + correspondence warp, live matrix), E1 and G (+ feature heads): every parameter gradient and the
    input gradients against torch autograd of the CPU oracle in float64 (exact-fp32 convs on the GPU side)."""
    _run_blocks("t_nets")


@pytest.mark.parametrize("stage", [1, 2])
@pytest.mark.parametrize("precision", [2, 0])
def test_generator_update_matches_reference_gradients(stage, precision):
    import gstep_diag as D
    res = D.compare_gstep(stage, precision=precision, verbose=True, assert_mode=True)
    bad = [n for n, ok in res if not ok]
    assert not bad, "outside the bar: %s" % bad[:20]


@pytest.mark.parametrize("stage", [1, 2])
def test_generator_update_gate_replay(stage):
    """Rounding vs gate flips, separated: the backward of the production convs on the gates RECORDED from the
    exact-fp32-conv run (ppst_amd/gates.py: every leaky-ReLU / ReLU / PReLU branch, the global-max-pool arg-max, the L1 sign).
    What is left is operand rounding: every parameter tensor within 5e-3 (max-norm and l2) of the exact run -- one bar, no list of
    cancelling sums; measured 3.6e-4 (stage 1) / 3.5e-3 (stage 2) at worst over the tensors, and the one-element tensors are held
    to 5e-3 or twice their float32 floor measured on the spot (gstep_diag.compare_gstep_replay).  The class bars of
    test_generator_update_matches_reference_gradients remain for the un-replayed run only; the flipped-gate counts it suffers
    are printed per site (1e-5 of the gates)."""
    import gstep_diag as D
    res = D.compare_gstep_replay(stage, verbose=True)
    bad = [n for n, ok in res if not ok]
    assert not bad, bad


def test_train_step_reduced_precision():
    """bf16 / fp16 compute with fp32 master weights (BASELINE configs[3]); bars in gstep_diag.t_train_precision."""
    _run_blocks("t_train_precision")


def test_facade_training_commands_are_differentiable_and_single_sourced():
    """``model(real, None, None, mask, command="compute_generator_losses")`` returns tensors with grad_fn, and the reference's
    ``sum(v.mean()).backward()`` (ppst_optimizer.py:86-88) on them leaves gradients BIT-IDENTICAL to
    GeneratorTrainer.losses_and_grads in the networks' ``p.grad`` (one composition, two entry points); the same for
    compute_discriminator_losses (:105-111) and compute_R1_loss (:119-123) against the discriminator trainer."""
    import gstep_diag as D
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import Options, create_model
    from ppst_amd.train import d_step_images
    real, mask, noise = D.gstep_inputs()
    real, mask = real.cuda(), mask.cuda()
    models = []
    for _ in range(2):
        sd = W.make_state_dict(17, bias_std=0.1, noise_weight=0.1)
        m = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True)
        m.noise = {k: v.cuda() for k, v in noise.items()}
        models.append(m)
    m1, m2 = models
    out1 = m1.trainer().losses_and_grads(real, mask)
    tr2 = m2.trainer()
    tr2.zero_grad()
    with torch.enable_grad():
        g_losses, g_metrics = m2(real, None, None, mask, command="compute_generator_losses")
        assert all(v.grad_fn is not None for v in g_losses.values())
        g_loss = sum([v.mean() for v in g_losses.values()])
        g_loss.backward()
    for k in ("G", "E2", "E1"):
        assert float(tr2.fp[k].grad.abs().max()) > 0.0
        assert torch.equal(m1.trainer().fp[k].grad, tr2.fp[k].grad), k
        # and they sit in p.grad of the networks' own parameters
        p = next(iter(getattr(m2, k).parameters()))
        assert p.grad is not None and p.grad.data_ptr() == tr2.fp[k].grad.data_ptr()
    for k, v in g_losses.items():
        assert torch.equal(v.detach().reshape(-1), out1[k].reshape(-1)), k
    with torch.no_grad():                                    # value path: no graph
        lv, _ = m2(real, None, None, mask, command="compute_generator_losses")
    assert all(v.grad_fn is None for v in lv.values())
    # discriminator iteration
    d1, d2 = m1.trainer().d_trainer, m2.trainer().d_trainer
    with torch.no_grad():
        rec, mix = d_step_images(m1, real, 1.0)
    ref = d1.losses_and_grads(real, rec, mix)
    d2.zero_grad()
    m2.criterionNCE.load_state_dict(m1.criterionNCE.state_dict())
    for net in ("E1", "E2", "G"):                            # the generator step above did not change weights; same images
        assert torch.equal(m1.trainer().fp[net].flat, tr2.fp[net].flat)
    with torch.enable_grad():
        d_losses, d_metrics, sp, gl = m2(real, mask, command="compute_discriminator_losses")
        assert all(v.grad_fn is not None for v in d_losses.values())
        sum([v.mean() for v in d_losses.values()]).backward()
    assert float(d2.grad.abs().max()) > 0.0 and torch.equal(d1.grad, d2.grad)
    assert all(torch.equal(ref[k].reshape(-1), d_losses[k].detach().reshape(-1)) for k in ref)
    pD = next(iter(m2.D.parameters()))
    assert pD.grad.data_ptr() == d2.grad.data_ptr()
    # lazy R1
    r1_ref = d1.r1_losses_and_grads(real, 10.0, 16)
    d2.zero_grad()
    with torch.enable_grad():
        r1 = m2(real, command="compute_R1_loss")
        (sum([v.mean() for v in r1.values()]) * 16).backward()
    assert torch.equal(r1_ref["D_R1"], r1["D_R1"].detach()) and torch.equal(d1.grad, d2.grad) and float(d2.grad.abs().max()) > 0.0


def test_cycwarp_branch_with_injected_metric():
    """ppst_model.py:175-179 (default lambda_Cycwarp = 5 in the reference): warp(real, corr) -> warp(., swap(corr)) -> metric.
    LPIPS weights are unavailable (parity of that term unpinned); the branch runs with an injected differentiable metric, adds
    ``image_warp_reg`` to the losses, back-propagates through both warps into the correspondence (and from there into G's
    feature heads), and refuses to run without a metric.  The block-level gradient check against torch autograd is in t_blocks2."""
    import gstep_diag as D
    from ppst_amd import autograd as A, weights as W
    from ppst_amd.ppst_model import Options, create_model
    real, mask, noise = D.gstep_inputs()
    real, mask = real.cuda(), mask.cuda()
    grads = {}
    for lam in (0.0, 5.0):
        sd = W.make_state_dict(17, bias_std=0.1, noise_weight=0.1)
        m = create_model(Options(training_stage=2, lambda_Cycwarp=lam), state_dict=sd, with_D=True, with_nce=True)
        m.noise = {k: v.cuda() for k, v in noise.items()}
        if lam > 0.0:
            with pytest.raises(RuntimeError, match="perceptual_metric"):
                m.trainer().losses_and_grads(real, mask)
            m.set_perceptual_metric(lambda a, b: A.L1LossFn.apply(a, b, 1.0))
            assert "perceptual_metric" not in dict(m.named_modules()) and len(m.state_dict()) == len(sd)
        out = m.trainer().losses_and_grads(real, mask)
        grads[lam] = {k: f.grad.clone() for k, f in m.trainer().fp.items()}
        if lam > 0.0:
            v = float(out["image_warp_reg"])
            assert v == v and v > 0.0
            # value = lambda * L1(double warp, real) with the inference-path kernels on the same correspondence
            with torch.no_grad():
                fea, fea1 = m.extract_feat_from_image(real)
                sps = torch.cat((fea, m.Rselfcorr(fea1)), dim=1)
                corr = m.corrm(sps, m.swap(sps))
                rec = m.warp(m.warp(real, corr), m.swap(corr))
                ref = 5.0 * float((rec - real).abs().mean())
            assert abs(v - ref) <= 2e-3 * ref, (v, ref)
        else:
            assert "image_warp_reg" not in out
    # the extra term reaches G's correspondence feature heads through the matrix -- and stops there: the heads read
    # x.detach() (generator.py:256,267), so the encoders' gradients are untouched, bit for bit
    assert not torch.equal(grads[0.0]["G"], grads[5.0]["G"])
    for k in ("E1", "E2"):
        assert torch.equal(grads[0.0][k], grads[5.0][k]), k


def test_generator_adam_step_and_alternation():
    """PPSTOptimizer mirror: first call = discriminator iteration (+ D_total), second = generator iteration; the
    parameter update equals torch.optim.Adam(lr 1e-3, betas (0, 0.99)) applied to the gradients of that step."""
    import numpy as np
    import train_oracle as TO
    import gstep_diag as D
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import Options, create_model
    from ppst_amd.train_g import PPSTOptimizer
    sd = W.make_state_dict(17, bias_std=0.1, noise_weight=0.1)
    m = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True)
    real, mask, noise = D.gstep_inputs()
    m.noise = {k: v.cuda() for k, v in noise.items()}
    opt = PPSTOptimizer(m)
    data = {"real_A": real.cuda(), "mask_A": mask.cuda()}
    d_before = opt.dis.flat.clone()
    dl = opt.train_one_step(data, 0)
    assert set(dl) == {"D_real", "D_rec", "D_mix", "D_total"}
    assert abs(dl["D_total"] - (dl["D_real"] + dl["D_rec"] + dl["D_mix"])) < 1e-5
    assert (opt.dis.flat != d_before).any()
    before = {k: f.flat.clone() for k, f in opt.gen.fp.items()}
    gl = opt.train_one_step(data, 2)
    assert {"G_L1", "G_GAN_rec", "G_GAN_mix", "G_styleContmix", "G_styleContrec", "Mask_warp", "G_L1_cyc", "L1_dist"} <= set(gl)
    for k, f in opt.gen.fp.items():
        g = f.grad.cpu().double()
        p0 = before[k].cpu().double()
        ref = TO.adam_reference({"p": p0}, {"p": g}, {}, 1e-3, 0.0, 0.99)["p"]
        live = g.abs() > 1e-12
        assert float((f.flat.cpu().double() - ref)[live].abs().max()) < 2e-6, k
        assert f.step_count == 1
    # the step's small-grid conv launches split K across blocks (ops.KSPLIT, round 5): they ran, and none gave up on its partner blocks
    from ppst_amd import ops
    assert ops.KSPLIT["value"] and ops.lib.ppst_conv_ksplit_check(ops._stream()) == 0


def test_training_loop_runs_saves_and_resumes(tmp_path):
    """train.py:24-60 on the HIP path: four iterations (D, G, D, G) on synthetic data with the image-count schedule, the
    model checkpoint in the reference's layout + optimiser state + iter.txt; a second process-equivalent resumes from them."""
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import Options, create_model
    from ppst_amd.train_g import PPSTOptimizer
    from ppst_amd.training import IterationCounter, SyntheticMaskDataset, load_optimizer_state, train_loop
    opt = Options(training_stage=2, lambda_Cycwarp=0.0, checkpoints_dir=str(tmp_path), name="run", isTrain=True, batch_size=2,
                  total_nimgs=8, save_freq=4, evaluation_freq=1000, print_freq=2, local_rank=0, continue_train=False, dataset_mode="celebamask")
    sd = W.make_state_dict(3, bias_std=0.1, noise_weight=0.1)
    m = create_model(opt, state_dict=sd, with_D=True, with_nce=True)
    optim = PPSTOptimizer(m)
    logs = []
    w0 = optim.gen.fp["G"].flat.clone()
    tracker = train_loop(opt, m, SyntheticMaskDataset(batch_size=2), optim, log=logs.append)
    mets = tracker.current_metrics()
    assert {"D_total", "G_L1", "G_GAN_rec", "Mask_warp"} <= set(mets) and all(v == v for v in mets.values())
    assert (optim.gen.fp["G"].flat != w0).any() and optim.gen.fp["G"].step_count == 2 and optim.dis.step_count == 2
    run = os.path.join(str(tmp_path), "run")
    assert os.path.islink(os.path.join(run, "latest_checkpoint.pth")) and os.path.exists(os.path.join(run, "latest_optimizer.pth"))
    assert any("Training finished" in s for s in logs)
    # resume: same weights, same Adam state, image count from iter.txt
    opt2 = Options(**{**opt.__dict__, "continue_train": True})
    m2 = create_model(opt2, state_dict=W.make_state_dict(99), with_D=True, with_nce=True)
    assert m2.load(verbose=False)
    optim2 = load_optimizer_state(PPSTOptimizer(m2), os.path.join(run, "latest_optimizer.pth"))
    assert torch.equal(optim2.gen.fp["G"].m, optim.gen.fp["G"].m) and optim2.gen.fp["G"].step_count == 2
    assert torch.allclose(optim2.gen.fp["G"].flat, optim.gen.fp["G"].flat) and IterationCounter(opt2).steps_so_far >= 4


def test_celebamask_dataset_loader(tmp_path):
    """image + integer label map -> {'real_A' in [-1, 1], 'mask_A' exact one-hot} (CelebAMask_dataset.py:40-60)."""
    import numpy as np
    Image = pytest.importorskip("PIL.Image")
    import resize_oracle as R
    from ppst_amd.training import CelebAMaskDataset
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "labels")
    imgs, labs = [], []
    for i in range(3):
        a = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
        l = rng.integers(0, 3, (64, 64), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / "images" / ("%d.png" % i))
        Image.fromarray(l).save(tmp_path / "labels" / ("%d.png" % i))
        imgs.append(a); labs.append(l)
    ds = CelebAMaskDataset(str(tmp_path), size=64, batch_size=3, preprocess="scale_shortside", flip=False)   # (the evaluators' transform)
    order = list(ds._order[:3])          # (read before next(): the prefetch thread already walks into the next epoch's order)
    batch = next(ds)
    assert tuple(batch["real_A"].shape) == (3, 3, 64, 64) and tuple(batch["mask_A"].shape) == (3, 3, 64, 64)
    for n, idx in enumerate(order):
        assert np.array_equal(batch["real_A"][n].cpu().numpy(), R.preprocess(imgs[idx], 64))
        onehot = np.stack([(labs[idx] == c) for c in range(3)]).astype(np.float32)
        assert np.array_equal(batch["mask_A"][n].cpu().numpy(), onehot)
    # second batch comes from the prefetch thread: same content rules
    batch2 = next(ds)
    assert tuple(batch2["real_A"].shape) == (3, 3, 64, 64) and torch.isfinite(batch2["real_A"]).all()
    # the launcher's training transform: square resize whatever the aspect (preprocess="resize") + ONE flip draw for image and mask
    a = rng.integers(0, 256, (48, 80, 3), dtype=np.uint8)
    l = rng.integers(0, 3, (48, 80), dtype=np.uint8)
    d2 = tmp_path / "ns"
    os.makedirs(d2 / "images"); os.makedirs(d2 / "labels")
    Image.fromarray(a).save(d2 / "images" / "0.png"); Image.fromarray(l).save(d2 / "labels" / "0.png")
    ds2 = CelebAMaskDataset(str(d2), size=64, batch_size=1, preprocess="resize", flip=True, seed=5)
    seen = set()
    for _ in range(8):
        b = next(ds2)
        x, mk = b["real_A"][0].cpu().numpy(), b["mask_A"][0].cpu().numpy()
        assert x.shape == (3, 64, 64)
        for flipped in (False, True):
            src, lab = (a[:, ::-1], l[:, ::-1]) if flipped else (a, l)
            ref = np.asarray(Image.fromarray(np.ascontiguousarray(src)).resize((64, 64), Image.BICUBIC)).astype(np.float32) / np.float32(255)
            ref = (ref - np.float32(0.5)) / np.float32(0.5)          # ToTensor, Normalize(0.5, 0.5)
            labr = np.asarray(Image.fromarray(np.ascontiguousarray(lab)).resize((64, 64), Image.NEAREST))
            if np.array_equal(x, ref.transpose(2, 0, 1)):
                assert np.array_equal(mk, np.stack([(labr == c) for c in range(3)]).astype(np.float32))   # mask mirrored WITH the image
                seen.add(flipped)
                break
        else:
            raise AssertionError("batch matches neither the plain nor the mirrored Pillow resize")
    assert seen == {False, True}
