#!/usr/bin/env python3
"""bench.py -- 512x512 portrait swaps / s on MI355X (BASELINE.json metric).

One "step" = the reference's single-pair swap recipe
(evaluation/simple_swapping_evaluator.py:44-60: encode(content),
extract_feat_from_image(content), extract_feat_from_image(style), Rselfcorr x2,
corrm, encode2(style, corr), lerp, decode) over a batch of 8 (content, style)
pairs at 512x512 = BASELINE.json configs[1] ("simple_swapping 512x512 batch=8,
generator+encoders forward only").  Inputs, weights and noise tensors are
resident in HBM before the timed region.  Weights are random-init (name-keyed
generator, ppst_amd/weights.py) with non-zero noise weights so the noise path is
exercised; data is synthetic.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8] [--precision bf16x3|bf16]
    N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
          (or just `python bench.py --gpus N`: with WORLD_SIZE unset the program starts its N ranks itself -- child processes
          through torch.distributed.run, like the reference's launcher, experiments/tmux_launcher.py:84-90 -- before it touches
          a GPU, relays rank 0's line and exits with the children's code)

Multi-GPU: pairs are independent -> each rank swaps its own batch (weak scaling),
no data-path collective; only the timing barrier / max-reduction use RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SWAP = 2189e9           # SURVEY.md section 8d / BASELINE.md section 3 (dense conv/GEMM only)
PEAK_BF16_DENSE_TF = 2500.0      # MI355X_MICROARCH.md: BF16 MFMA dense peak
HBM_PEAK_GBS = 8000.0


def _variant(info):
    return (info[7] >> 12) & 0xff


def _is_wgrad(info):
    return (info[7] & 0xfff) == 0


def issued(rows, passes_default):
    """(algorithmic flop, MFMA flop the matrix pipe ISSUED, ms) over prof_detail rows.  Per launch: algorithmic x the MFMA passes of
    its precision mode (bf16x3: 3; single-pass modes and K64: 1; a weight gradient: ``passes_default``) x the factor of the kernel
    form that ran it -- conv_wino_kernel (variant 10) 2/3 (F(2,3) along x: 12 products per output pair instead of 18); the
    nine-product fused upscale (variant 11) 9/16 of the four-phase form's products over blocks that cover 16 x 16 grid positions
    for 15 x 15 outputs; every other kernel 1 (zero-weight pad steps and ragged last tiles are not counted: < 2 % on this path)."""
    alg = iss = ms = 0.0
    for t, fl, info in rows:
        alg += fl
        ms += t
        if _is_wgrad(info):
            iss += fl * passes_default
            continue
        v, prec = _variant(info), (info[7] >> 24) & 0xf
        npass = {0: 3, 4: 2}.get(prec, 1)
        f = 1.0
        if v == 10:
            f = 2.0 / 3.0
        elif v == 11:
            th, tw = info[1], info[2]
            f = 9.0 * 256 * ((th + 14) // 15) * ((tw + 14) // 15) / (16.0 * th * tw)
        iss += fl * npass * f
    return alg, iss, ms


def issued_fields(rows, passes_default=3):
    """the keys every roofline object carries since round 5 (round-4 verdict, weak #2): what the matrix pipe did."""
    alg, iss, ms = issued(rows, passes_default)
    t = iss / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"mfma_issued_tflops": t, "frac_issued": t / PEAK_BF16_DENSE_TF,
            "frac_issued_note": "PRIMARY: MFMA flop the kernels issued (algorithmic x passes per MAC of the mode x 2/3 for the Winograd "
                                "launches x the nine-product upscale's 9/16 over its 16 x 16-position blocks) / kernel time / the guide's dense "
                                "bf16 peak 2500 TFLOP/s -- a fraction of the pipe that cannot exceed 1; `frac` (algorithmic flop against "
                                "2500 / passes) is kept as a secondary figure and can exceed what the pipe did since round 4"}


def swap_step(model, content, style, alpha, glue):
    """The reference recipe, batched; returns the output image tensor."""
    # The recipe's encoder passes -- encode(content), and the E1 / E2 inside extract_feat_from_image(content) and (style) -- run
    # as ONE batch of 3B images, its two generator feature passes as one batch of 2B (command "extract_feat", the reference's
    # own).  Same work as the three separate commands (content is encoded twice, as there), same results bit for bit (no
    # kernel choice depends on the batch size): fewer launches, fuller grids on the thin encoder layers and the 64x64 stage.
    B = content.shape[0]
    sp3, gl3 = model(torch.cat((content, content, style), 0), command="encode")
    sp, gl_c = sp3[:B], [g[:B] for g in gl3]
    _, fea, fea1 = model(sp3[B:], [g[B:] for g in gl3], command="extract_feat")
    fea = torch.cat((fea, model(fea1, command="Rselfcorr")), dim=1)
    fea_c, fea_s = fea[:B], fea[B:]
    corr = model(fea_s, fea_c, command="corrm")
    _, gl_w = model(style, corr, command="encode2")
    code = glue.lerp(gl_c, gl_w, alpha)
    return model(sp, code, target=None, command="decode")


def _pmc_file(what="swap"):
    """Newest committed PMC summary of one workload (profiles/rNN_bench_<what>_pmc_traffic.json, written by tests/profile_round.sh;
    rounds 2-3 committed the swap line's as profiles/rNN_pmc_traffic.json).  PMC counters cannot be read from inside the process:
    the roofline quotes the committed measurement and says which commit / command it was taken at."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_%s_pmc_traffic.json" % what)))
    if not files and what == "swap":
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if "_bench_" not in f)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_traffic(what, prefixes=None):
    """{"hbm_bytes_per_launch", "launches", "file", "measured_at_commit"} of the kernels whose names contain one of `prefixes`
    (None: the conv forward / input-gradient group, "conv_mfma" in the summary) from the committed PMC passes of workload
    `what`; None when no summary is committed."""
    d, name = _pmc_file(what)
    if d is None:
        return None
    try:
        if prefixes is None:
            g = d["conv_mfma"]
            n, b = g["launches"], g["hbm_bytes_per_launch"]
        else:
            rows = [v for k, v in d["kernels"].items() if any(p in k for p in prefixes)]
            n = sum(r["launches"] for r in rows)
            b = sum(r["fetch_bytes_corrected"] + r["write_bytes"] for r in rows) / max(n, 1)
        return {"hbm_bytes_per_launch": b, "launches": n, "file": "profiles/" + name, "measured_at_commit": d.get("commit")}
    except Exception:
        return None


def gf_traffic():
    """the guided filter's PMC summary (profiles/rNN_gf_pmc.json, tests/gf_prof.sh): bytes per pixel of a batch of four 1024^2 images"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gf_pmc.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {"bytes_per_pixel": d["bytes_per_pixel"], "bytes_per_pixel_fetch_doubled": d["bytes_per_pixel_fetch_doubled"],
                "file": "profiles/" + os.path.basename(files[-1]), "note": d.get("note"), "command": d.get("command")}
    except Exception:
        return None


def conv_traffic():
    """HBM bytes per conv launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs of this
    same command; summary committed under profiles/); None if absent."""
    d, _ = _pmc_file()
    try:
        return d["conv_mfma"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def conv_mfma_busy():
    """Matrix-pipe busy fraction of the conv launches from the committed SQ_VALU_MFMA_BUSY_CYCLES pass (profiles/rNN_pmc_mfma.json,
    tests/pmc_mfma_summary.py): a hardware-counter companion of roofline.frac, at the clock the kernels ran at."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_swap_pmc_mfma.json")))
    if not files:       # (rounds 2-3 committed the swap line's as profiles/rNN_pmc_mfma.json)
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")) if "_bench_" not in f)
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {"all_conv_launches": d["all_conv_launches"]["mfma_busy_frac"], "file": "profiles/" + os.path.basename(files[-1]),
                "measured_at_commit": d.get("commit")}
    except Exception:
        return None


def conv_traffic_source():
    d, name = _pmc_file()
    if d is None:
        return None
    return {"file": "profiles/" + name, "measured_at_commit": d.get("commit"), "command": d.get("command")}


def upfirdn2d_rate(B, dev):
    """North-star side metric: achieved HBM GB/s of the Blur (upfirdn2d) on the largest encoder
    blur of the step -- (B,32,512,512) NHWC, 3x3 [1,2,1] taps, reflection pad, space-to-depth
    output -- algorithmic bytes 4*B*C*(Hin*Win + Hout*Wout) / HIP-event time."""
    from ppst_amd import ops
    x = torch.randn(B, 512, 512, 32, device=dev)
    k = torch.tensor([1., 2., 1.], device=dev)
    k2 = (k[:, None] * k[None, :] / 16).contiguous()
    for _ in range(3):
        y, (oh, ow) = ops.blur_nhwc(x, k2, 2, 1, ops.PAD_REFLECT, s2d=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.blur_nhwc(x, k2, 2, 1, ops.PAD_REFLECT, s2d=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = 4.0 * B * 32 * (512 * 512 + oh * ow)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "upfirdn2d_chan<3,3,1,s2d> (B,512,512,32) reflect pad", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "us_per_launch": ms * 1e3}


def conv_pmc():
    """Effective clock of the conv launches from the GRBM_GUI_ACTIVE pass (profiles/, see conv_traffic)."""
    d, _ = _pmc_file()
    try:
        return d["conv_mfma"].get("effective_clock_ghz")
    except Exception:
        return None


def cpu_baseline(seed):
    """The CPU oracle (a port of the reference's PyTorch-CPU path, pinned to it by tests/golden) timed on the host
    cores of this box, fp32: one untimed warm-up of the 512x512 recipe, then ONE timed run of one pair
    per thread count of a small sweep ({8, 16, 32, 64, 128} up to the cores this process may use); the best is reported with its
    thread count.  A batch-8 run costs the same per pair on the host (the oracle's convs are already multi-threaded over
    pixels), so the batch-1 rate is the bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ppst_oracle as O
    from ppst_amd import weights as W
    try:
        nproc = len(os.sched_getaffinity(0))     # the cores this process may run on (not the whole host's)
    except AttributeError:
        nproc = os.cpu_count() or 1
    base_threads = torch.get_num_threads()
    sd = W.make_state_dict(seed, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    noise = W.make_noise(seed + 2, 1)
    imgs = W.synthetic_images(seed + 4, 2)
    orc = O.PPSTOracle(sd, noise=noise)
    # thread sweep (round-4 verdict, weak #6: 128 torch threads on per-layer fp32 convs of this size oversubscribe a 2 x 64-core
    # host -- the number beside the GPU line is the BEST of a small sweep): one untimed warm-up of the same 512x512 recipe (thread
    # pool, allocator arenas, oneDNN primitive caches are per shape), then ONE timed run per thread count, fewest first
    sweep = sorted({n for n in (8, 16, 32, 64, 128) if n <= nproc} | {min(nproc, base_threads)})
    runs = {}
    with torch.no_grad():
        torch.set_num_threads(sweep[len(sweep) // 2])
        t0 = time.time()
        orc.simple_swap(imgs[0:1], imgs[1:2], alpha=1.0)
        warm = time.time() - t0
        print("[bench] cpu_baseline warm-up: %.1f s on %d threads" % (warm, torch.get_num_threads()), file=sys.stderr, flush=True)
        for n in sweep:
            torch.set_num_threads(n)
            t0 = time.time()
            orc.simple_swap(imgs[0:1], imgs[1:2], alpha=1.0)
            runs[n] = time.time() - t0
            print("[bench] cpu_baseline %d threads: %.1f s" % (n, runs[n]), file=sys.stderr, flush=True)
    torch.set_num_threads(base_threads)
    best = min(runs, key=runs.get)
    dt = runs[best]
    model, phys = host_cpu()
    return {"value": 1.0 / dt, "unit": "swaps/s", "cores": best, "kind": "port",
            "cpu_model": model, "physical_cores_host": phys, "cores_available_to_process": nproc, "torch_threads": best,
            "gflops": FLOP_PER_SWAP / dt / 1e9,
            "thread_sweep_s_per_swap": {str(k): round(v, 2) for k, v in runs.items()},
            "sample": "thread sweep {%s}: 1 untimed warm-up run (%.1f s) + ONE timed run of 1 pair (batch 1) of the same 512x512 recipe per "
                      "thread count, fp32; best = %d threads, %.1f s per swap (all: %s); a batch-8 run costs the same per pair on the host "
                      "(the oracle's convs are already threaded over pixels), so the batch-1 rate is the bounded sample of BASELINE.md "
                      "section 4's B = 8 leg" % (", ".join(map(str, sweep)), warm, best, dt,
                                                 ", ".join("%d: %.1f s" % (k, v) for k, v in runs.items()))}


def host_cpu():
    """(model name, physical core count of the host) from /proc/cpuinfo -- BASELINE.md section 4 asks for both."""
    try:
        model, cores = None, set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        return model, (len(cores) or None)
    except OSError:
        return None, None


DTYPE_NOTE = {"bf16x3": "bf16x3 (fp32 split into hi+lo bf16, 3 MFMA passes, fp32 accumulate)",
              "bf16": "bf16 (single MFMA pass, fp32 accumulate / statistics / modulation)",
              "fp16": "fp16 (single MFMA pass, fp32 accumulate / statistics / modulation)",
              "fp16x2": "fp16x2 (activation fp16 hi+lo, weight fp16: 2 MFMA passes, fp32 accumulate) -- measured experiment"}
FLOP_PER_IMAGE_PASS = (23.7 + 22.7 + 753.9) * 1e9     # E1 + E2 + G with feature heads
FLOP_PER_PAIR_PASS = (17.2 + 38.8 + 486.2) * 1e9      # corrm + E2 with warp + G decode


def bench_grid(args, rank, world, dev, barrier, max_over_ranks):
    """BASELINE configs[2]: swapping_grid over an 8 x 8 folder at 512x512 with the guided-filter post-process
    (content_style_grid_generation_evaluator.py:36-99).  One step = the whole grid: 16 per-image passes + one exchange
    + 64 pair passes, images and pairs sharded over the ranks (strong scaling: the grid is fixed as N grows)."""
    from ppst_amd import weights as W
    from ppst_amd.evaluation import swapping_grid
    from ppst_amd.ppst_model import create_model
    sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    model = create_model(state_dict=sd, device=dev)
    model.noise = {k: v.to(dev) for k, v in W.make_noise(2, 1).items()}      # one fixed row for every batch row
    contents, styles = W.synthetic_images(4, 8).to(dev), W.synthetic_images(5, 8).to(dev)
    from ppst_amd import glue, ops
    with torch.no_grad():
        for _ in range(args.warmup):
            out = swapping_grid(model, contents, styles, rank, world, smooth=True)
        barrier()
        ops.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = swapping_grid(model, contents, styles, rank, world, smooth=True)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        detail = ops.prof_detail()
        conv_ms, conv_launches, conv_flop = ops.prof_collect()
        ops.prof_enable(False)
        # the guided filter of one pair batch (8 images at 512^2), HIP events on the launch stream
        gu = glue.tensor2im(contents)
        for _ in range(2):
            ops.guided_filter(gu, gu, 30, (0.02 * 255) ** 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.guided_filter(gu, gu, 30, (0.02 * 255) ** 2)
        e1.record()
        torch.cuda.synchronize()
        gf_ms = e0.elapsed_time(e1) / 5
    assert all(torch.isfinite(v).all() for v in out.values())
    pairs = 64 * args.steps
    flop = args.steps * (16 * FLOP_PER_IMAGE_PASS + 64 * FLOP_PER_PAIR_PASS)
    passes = {"bf16x3": 3, "fp16x2": 2}.get(args.precision, 1)
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    gf_bytes = 9.0 * 512 * 512 * 8
    roof = {"roofline": {"kernel": "ppst_conv2d_mfma launches of the grid (16 image passes + 64 pair passes on this rank's share)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16_DENSE_TF / passes, "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16_DENSE_TF / passes),
                         "frac_vs_dense_bf16": achieved / PEAK_BF16_DENSE_TF, **issued_fields(detail), "launches": conv_launches,
                         "kernel_ms_total": conv_ms, "share_of_step_time": conv_ms * 1e-3 / dt, "traffic": None},
            "roofline_guided_filter": {"kernel": "gf_* kernels<30> (guided_filter.hip) on one batch of 8 images at 512^2", "bound": "hbm",
                                       "achieved": gf_bytes / (gf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": gf_bytes / (gf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_batch": gf_ms,
                                       "bytes_note": "algorithmic minimum 9 B / pixel (uint8 guide + source in, uint8 out)", "traffic": None}}
    return {**roof, "metric": "512x512 grid swaps/sec (8x8 folder, guided filter on)", "value": pairs / dt, "unit": "swaps/s (all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": "swapping_grid 8x8 folder at 512 (BASELINE configs[2]): 16 image passes + 64 pair passes + guided filter",
                       "sharding": "images k mod N, one all_gather of sp | fea||Rselfcorr (12.6 MB per image), pairs (i*8+j) mod N"},
            "algorithmic_tflops_whole_job": flop / dt / 1e12}


def bench_train(args, rank, world, dev, barrier, max_over_ranks):
    """BASELINE configs[3]: CelebAMaskHQ_default train step at 512x512, batch 2 per GPU (the reference's default), random
    init, synthetic images + label maps.  One step = one discriminator iteration (with the lazy R1 pass every 16th) and
    one generator iteration (PPSTOptimizer.train_one_step twice), data parallel: flat gradient all-reduce per network."""
    from ppst_amd import weights as W
    from ppst_amd.ppst_model import Options, create_model
    from ppst_amd.train_g import PPSTOptimizer
    B = args.train_batch
    sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
    model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
    model.noise = "random"
    real = W.synthetic_images(40 + rank, B).to(dev)
    g = torch.Generator().manual_seed(7 + rank)
    lab = torch.randint(0, 3, (B, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
    mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
    # PPST_BENCH_FORCE_DIST on one GPU: a 1-rank RCCL group is up (main()); run the data-parallel code path against it -- the
    # asynchronous flat all-reduces fired from inside backward, the deferred D step, the NCE all_gather -- by telling the
    # optimizer the world is 2 (gradients are halved: a rehearsal of the collectives on the device, not a measurement)
    rehearsal = bool(os.environ.get("PPST_BENCH_FORCE_DIST")) and world == 1
    opt = PPSTOptimizer(model, world=2 if rehearsal else world)
    data = {"real_A": real, "mask_A": mask}
    from ppst_amd import ops
    for _ in range(args.warmup):
        opt.train_one_step(data, 0); opt.train_one_step(data, 0)
    # (the lazy-R1 pass builds plans of its own -- the double-backward convs -- the first time it runs: one untimed pass belongs to the
    #  warm-up like the first D + G iterations do; the iteration counters, hence the 1-in-16 schedule of the timed region, are untouched)
    if args.warmup > 0 and opt.dis is not None and float(getattr(model.opt, "lambda_R1", 10.0)) > 0.0:
        opt.r1_iteration(real)
    # The lazy R1 pass (optimizers/ppst_optimizer.py:116-126) fires on every R1_once_every-th (16th) discriminator iteration.  The
    # timed region holds its cost one of two ways (round-4 verdict, missing #3): with --steps a multiple of 16 the region CONTAINS
    # steps / 16 R1 passes (any 16 consecutive iterations hold exactly one), ms_per_step is the plain quotient; otherwise one R1
    # pass is timed separately behind the region and ms_per_step = step + r1 / 16.  Both terms are printed either way.
    r1_every = opt.R1_once_every
    barrier()
    ops.prof_enable(True)        # HIP events around every conv (forward / input-gradient) and weight-gradient launch, on their stream
    c0 = opt.discriminator_iter_counter
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dl = opt.train_one_step(data, 0)
        gl = opt.train_one_step(data, 0)
    barrier()
    dt_raw = max_over_ranks(time.perf_counter() - t0)
    r1_in_region = opt.discriminator_iter_counter // r1_every - c0 // r1_every
    detail = ops.prof_detail()
    ops.prof_collect()
    ops.prof_enable(False)
    assert all(v == v for v in list(dl.values()) + list(gl.values())), "NaN loss"
    assert ops.ksplit_check(), "a K-split conv launch gave up waiting for its partner blocks (ppst_conv_ksplit_check)"
    # one R1 pass alone (zero_grad -> compute_R1_loss -> x16 -> backward -> Adam on D), warmed up once
    r1_ms = None
    if opt.dis is not None and float(getattr(model.opt, "lambda_R1", 10.0)) > 0.0:
        opt.r1_iteration(real)
        barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            opt.r1_iteration(real)
        barrier()
        r1_ms = max_over_ranks(time.perf_counter() - t1) / 2 * 1e3
    exact = args.steps % r1_every == 0
    step_ms_plain = (dt_raw * 1e3 - r1_in_region * (r1_ms or 0.0)) / args.steps        # the D + G iterations without any R1 pass
    step_ms = dt_raw * 1e3 / args.steps if exact else step_ms_plain + (r1_ms or 0.0) / r1_every
    dt = step_ms * 1e-3 * args.steps
    imgs = world * B * args.steps
    passes = {"bf16x3": 3, "fp16x2": 2}.get(args.precision, 1)

    pmc_what = ({"bf16x3": "train", "bf16": "train_bf16"}.get(args.precision) if B == 2 else
                {"bf16x3": "train_b8"}.get(args.precision) if B == 8 else None)

    def roof(rows, kernel, npass, prefixes=None):
        ms, fl = sum(r[0] for r in rows), sum(r[1] for r in rows)
        tr = pmc_traffic(pmc_what, prefixes) if pmc_what else None
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_BF16_DENSE_TF / npass
        return {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "peak_note": "bf16 dense MFMA 2500 TF / %d MFMA passes per algorithmic MAC" % npass,
                "frac_vs_dense_bf16": ach / PEAK_BF16_DENSE_TF, **issued_fields(rows, npass), "launches_per_step": len(rows) / args.steps,
                "kernel_ms_per_step": ms / args.steps, "algorithmic_tflop_per_step": fl / args.steps / 1e12,
                "share_of_step_time": ms * 1e-3 / dt_raw, "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_source": tr}
    wg = [r for r in detail if _is_wgrad(r[2])]
    cv = [r for r in detail if not _is_wgrad(r[2])]
    # MFMA passes per algorithmic MAC of the weight gradient AS IT RAN: three (bf16 hi / lo split) -- or one in precision mode 1,
    # where ops.conv_wgrad multiplies the hi halves only (ops.WGRAD_TR["bf16_single_pass"], the transposed-read kernel, form 2)
    wtr = ops.WGRAD_TR
    npass_wg = 1 if (args.precision == "bf16" and wtr["bf16_single_pass"] and wtr["value"] and wtr["form"] == 2) else 3
    return {"metric": "512x512 train images/sec (one D + one G iteration per step)", "value": imgs / dt, "unit": "images/s (all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "ms_per_step_terms": {"d_plus_g_iteration_ms": step_ms_plain, "r1_pass_ms": r1_ms, "r1_once_every": r1_every,
                                  "r1_passes_inside_timed_region": r1_in_region, "timed_region_ms": dt_raw * 1e3,
                                  "how": ("timed region of %d steps contains %d lazy-R1 passes: ms_per_step = region / steps" % (args.steps, r1_in_region))
                                         if exact else "ms_per_step = d_plus_g_iteration_ms + r1_pass_ms / r1_once_every (R1 pass timed separately)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": "CelebAMaskHQ_default train step 512x512, batch %d per GPU, training stage 2, lambda_Cycwarp 0 "
                                   "(lpips unavailable), random init (BASELINE configs[3])" % B,
                       "collectives": "flat gradient all-reduce per network (D 29.0 M, G 42.5 M, E2 27.0 M, E1 0.83 M fp32) + one "
                                      "[24, 2048] all_gather of the NCE keys",
                       **({"collectives_rehearsal": "1-rank RCCL group, optimizer told world = 2 (not a measurement)"} if rehearsal else {})},
            # the dominant kernel of the step by time: the weight gradient
            "roofline": roof(wg, "conv_wgrad_tr2_kernel / conv_wgrad_x3_kernel (conv weight gradients on the bf16 matrix pipe, %s; all launches of the step)"
                             % ("hi halves only: ONE MFMA pass" if npass_wg == 1 else "hi + lo split: three MFMA passes"), npass_wg,
                             ("conv_wgrad",)),
            "roofline_conv": roof(cv, "ppst_conv2d_mfma launches of the step: forward and input-gradient convs (same kernels as the swap line)", passes),
            "algorithmic_tflops_conv_and_wgrad": (sum(r[1] for r in detail)) / dt_raw / 1e12,
            "losses": {**dl, **gl}}


def bench_hires(args, rank, world, dev, barrier, max_over_ranks):
    """BASELINE configs[4]: 1024x1024 swap + guided-filter post-process, fp16 generator with fp32 style modulation (precision
    mode 3: fp16 MFMA operands, fp32 accumulate / instance-norm statistics / StyleMod), one GPU per batch.  The correspondence
    recipe does not exist at 1024^2 in the reference (its matching assumes 64 x 64 feature maps), so a
    swap here is the reference's plain command sequence: encode(content) -> sp, encode(style) -> gl, decode(sp, gl,
    target=content) = G + GIFSmoothing(r = 30, eps = (0.02 * 255)^2) (models/ppst_model.py:288-306, photo_gif.py:25-46).
    One step = one batch of pairs; nothing is cached (both encode commands run E1 and E2 like the reference's)."""
    from ppst_amd import glue, ops, weights as W
    from ppst_amd.ppst_model import create_model
    B, S = args.batch, 1024
    sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    model = create_model(state_dict=sd, device=dev)
    model.noise = {k: v.to(dev) for k, v in W.make_noise(2 + rank, 1, S=S // 8).items()}     # one fixed row for every batch row
    imgs = W.synthetic_images(4 + rank, 2, size=S)
    content = imgs[0:1].expand(B, -1, -1, -1).contiguous().to(dev)
    style = imgs[1:2].expand(B, -1, -1, -1).contiguous().to(dev)

    def step():
        sp, _ = model(content, command="encode")
        _, gl = model(style, command="encode")
        return model(sp, gl, target=content, command="decode")
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        barrier()
        ops.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        detail = ops.prof_detail()
        conv_ms, conv_launches, conv_flop = ops.prof_collect()
        ops.prof_enable(False)
        assert torch.isfinite(out).all()
        # the guided filter alone, HIP events on the launch stream (= torch's current stream): uint8 guide + uint8 source in,
        # float image out of the command; algorithmic minimum 9 bytes per pixel with uint8 I/O
        sp, _ = model(content, command="encode")
        _, gl = model(style, command="encode")
        raw = model(sp, gl, target=None, command="decode")
        gu, su = glue.tensor2im(content), glue.tensor2im(raw)
        for _ in range(2):
            ops.guided_filter(gu, su, 30, (0.02 * 255) ** 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ops.guided_filter(gu, su, 30, (0.02 * 255) ** 2)
        e1.record()
        torch.cuda.synchronize()
        gf_ms = e0.elapsed_time(e1) / n
    passes = {"bf16x3": 3, "fp16x2": 2}.get(args.precision, 1)
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_BF16_DENSE_TF / passes
    gf_bytes = 9.0 * S * S * B
    pmc_what = "hires" if (args.precision == "fp16" and B == 4) else None
    conv_tr = pmc_traffic(pmc_what) if pmc_what else None
    gf_tr = gf_traffic() if B == 4 else None
    swaps = world * B * args.steps
    return {"metric": "1024x1024 swaps/sec (encode + decode + guided filter)", "value": swaps / dt, "unit": "swaps/s (all GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": "1024x1024 swap + guided-filter post-process, batch %d per GPU (BASELINE configs[4]): encode(content), "
                                   "encode(style), decode(sp, gl, target=content)" % B,
                       "precision": "conv operands %s; accumulation, instance-norm statistics and StyleMod fp32" % args.precision},
            "roofline": {"kernel": "ppst_conv2d_mfma launches of the step (E1, E2, G at 1024^2)", "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_note": "bf16 / fp16 dense MFMA 2500 TF / %d MFMA passes per algorithmic MAC" % passes,
                         "frac_vs_dense_bf16": achieved / PEAK_BF16_DENSE_TF, **issued_fields(detail), "launches": conv_launches, "kernel_ms_total": conv_ms, "share_of_step_time": conv_ms * 1e-3 / dt,
                         "traffic": conv_tr["hbm_bytes_per_launch"] if conv_tr else None, "traffic_source": conv_tr},
            "roofline_guided_filter": {"kernel": "gf_s1_fused_kernel<30, 32> + gf_s2_fused_kernel<30, 64> (guided_filter.hip, round 5): two launches, every box "
                                                 "sum kept on the chip (column sums in registers, row windows through LDS), (a, b) between them as 12 half planes",
                                       "bound": "hbm", "achieved": gf_bytes / (gf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": gf_bytes / (gf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_batch": gf_ms,
                                       "bytes_note": "algorithmic minimum 9 B / pixel (uint8 guide + source in, uint8 out); the implementation moves "
                                                     "uint8 rows + halo in, 24 B / pixel of half planes out and back in (x halo factors), 12 (+ 3) out",
                                       "share_of_step_time": gf_ms * args.steps * 1e-3 / dt,
                                       "traffic": gf_tr["bytes_per_pixel"] * S * S * B if gf_tr else None,
                                       "traffic_note": "HBM bytes per batch = the two kernels of one call (PMC; narrow loads: raw FETCH_SIZE, see traffic_source.note)",
                                       "traffic_bytes_per_pixel": gf_tr["bytes_per_pixel"] if gf_tr else None,
                                       "traffic_bytes_per_pixel_fetch_doubled": gf_tr["bytes_per_pixel_fetch_doubled"] if gf_tr else None,
                                       "traffic_source": gf_tr}}


def extras(args, dev):
    """The other BASELINE configs, measured in this same process behind the headline line so that the driver's default run
    carries them (round-3 verdict: only the swap line was driver-timed): configs[3] train step in the fp32-class mode and in
    --precision bf16, configs[4] 1024^2 swap + guided filter in fp16.  Each is the same code path as `--workload train|hires`
    with few steps; every sub-object carries its own roofline objects."""
    import copy
    from ppst_amd import ops
    out = {}

    def run(name, fn, precision, steps, warmup, **kw):
        a = copy.copy(args)
        a.precision, a.steps, a.warmup = precision, steps, warmup
        for k, v in kw.items():
            setattr(a, k, v)
        ops.set_precision({"bf16x3": 0, "bf16": 1, "fp16": 3}[precision])
        t0 = time.time()
        try:
            r = fn(a, 0, 1, dev, torch.cuda.synchronize, lambda dt: dt)
            r["wall_s_incl_setup"] = time.time() - t0
            out[name] = r
        except Exception as e:      # an extra must not take the headline line down
            out[name] = {"error": repr(e)}
        finally:
            ops.set_precision(0)
            torch.cuda.empty_cache()
    # 16 steps: the timed region of a train line contains exactly one lazy-R1 pass (ppst_optimizer.py:116-126)
    run("train_bf16x3", bench_train, "bf16x3", 16, 1)
    run("train_bf16", bench_train, "bf16", 16, 1)
    run("train_bf16x3_batch8", bench_train, "bf16x3", 16, 1, train_batch=8)      # B = 8 per GPU beside the reference's 2
    run("hires_fp16", bench_hires, "fp16", 5, 1, batch=4)
    run("grid", bench_grid, "bf16x3", 3, 1)                                      # configs[2] on one GPU: 8 x 8 folder, guided filter on
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (python -m torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1 at a free port) with this program's own arguments, let rank 0's single JSON line
    through on the inherited stdout, and return the launcher's exit code (non-zero if any rank failed).  The reference starts
    its ranks the same way (experiments/tmux_launcher.py:84-90: python -m torch.distributed.launch --nproc_per_node ...
    train.py).  Never exec: the parent has not touched a GPU and only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """RCCL / gloo print connection banners on stdout when a communicator comes up; this program's stdout is ONE JSON line:
    bring the communicator up with fd 1 pointed at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def bench_stub(args, rank, world):
    """--workload stub: the harness of the line without a GPU -- process group (gloo), barrier on both sides of the timed
    region, max over ranks, ONE JSON line from rank 0.  For the launcher test (tests/test_bench_launcher_cpu.py)."""
    import torch.distributed as dist
    if world > 1:
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
            dist.barrier()

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(0.001)
    if os.environ.get("PPST_BENCH_STUB_FAIL_RANK") == str(rank):      # (launcher test: a rank that dies must fail the whole run)
        raise SystemExit(3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))                    # ranks differ: the line must carry the slowest
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "stub steps/sec", "value": world * args.steps / dt, "unit": "steps/s (all ranks)", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "launcher / barrier / max-reduce harness only (gloo, no GPU)"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "fp16", "fp16x2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-batch", type=int, default=2, help="--workload train: images per GPU (the reference's default is 2; SURVEY 8d also asks for 8)")
    ap.add_argument("--no-extras", action="store_true", help="swap workload on one GPU: skip the train / hires sub-measurements of the default line")
    ap.add_argument("--workload", default="swap", choices=["swap", "grid", "train", "hires", "stub"],
                    help="swap: BASELINE configs[1] (the headline line); grid: configs[2], 8x8 folder at 512 with the guided filter, "
                         "images and pairs sharded over the ranks; train: configs[3], one D (+ lazy R1) and one G iteration per step; "
                         "hires: configs[4], 1024x1024 encode / decode + guided filter (use --precision fp16)")
    ap.add_argument("--conv-variant", type=int, default=None, help="2 (default): N-256 / 128x64-wave-tile kernel where eligible; 0: the 64x64-wave-tile kernel everywhere; 1: one-wave-per-SIMD experiment")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # (before anything touches a GPU; the parent only waits)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if args.workload == "stub":
        return bench_stub(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("PPST_BENCH_FORCE_DIST"):   # the override exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        with stdout_to_stderr():      # (the communicator comes up at the first barrier)
            dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()

    from ppst_amd import glue, ops, weights as W
    from ppst_amd.ppst_model import create_model
    ops.set_precision({"bf16x3": 0, "bf16": 1, "fp16": 3, "fp16x2": 4}[args.precision])
    if args.conv_variant is not None:
        ops.CONV_VARIANT["value"] = args.conv_variant

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.workload != "swap":
        res = {"grid": bench_grid, "train": bench_train, "hires": bench_hires}[args.workload](args, rank, world, dev, barrier, max_over_ranks)
        if rank == 0:
            print(json.dumps(res))
        if dist is not None:
            dist.destroy_process_group()
        return

    B = args.batch
    sd = W.make_state_dict(0, with_D=False, with_nce=False, bias_std=0.1, noise_weight=0.1)
    model = create_model(state_dict=sd, device=dev)
    model.noise = {k: v.to(dev) for k, v in W.make_noise(2 + rank, B).items()}
    imgs = W.synthetic_images(4 + rank, 2 * B).to(dev)
    content, style = imgs[:B].contiguous(), imgs[B:].contiguous()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = swap_step(model, content, style, 1.0, glue)
        barrier()
        # which of the bracketed conv launches belong to StyledConv (the "modulated_conv2d" of BASELINE's metric (ii)): every
        # ConvPlan call is one ppst_conv2d_mfma launch, counted in order; those issued inside Generator.styled_conv are tagged
        tag = {"n": 0, "depth": 0, "styled": set()}
        plan_call, g_cls = ops.ConvPlan.__call__, type(model.G)
        styled_conv = g_cls.styled_conv

        def counted_call(self, *a, **k):
            if tag["depth"]:
                tag["styled"].add(tag["n"])
            tag["n"] += 1
            return plan_call(self, *a, **k)

        def tagged_styled_conv(self, *a, **k):
            tag["depth"] += 1
            try:
                return styled_conv(self, *a, **k)
            finally:
                tag["depth"] -= 1
        ops.ConvPlan.__call__, g_cls.styled_conv = counted_call, tagged_styled_conv
        ops.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = swap_step(model, content, style, 1.0, glue)
        barrier()
        dt = time.perf_counter() - t0
        ops.ConvPlan.__call__, g_cls.styled_conv = plan_call, styled_conv
        detail = ops.prof_detail()
        conv_ms, conv_launches, conv_flop = ops.prof_collect()
        ops.prof_enable(False)
    assert torch.isfinite(out).all()
    assert len(detail) == tag["n"] == conv_launches, (len(detail), tag["n"], conv_launches)
    st_ms = sum(detail[i][0] for i in tag["styled"])
    st_flop = sum(detail[i][1] for i in tag["styled"])

    dt = max_over_ranks(dt)

    if rank == 0:
        swaps = world * B * args.steps
        passes = {"bf16x3": 3, "fp16x2": 2}.get(args.precision, 1)
        achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_BF16_DENSE_TF / passes
        res = {
            "metric": "512x512 portrait swaps/sec/GPU", "value": swaps / dt, "unit": "swaps/s (all GPUs)",
            "per_gpu": swaps / dt / world, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NOTE[args.precision],
            "data": "synthetic",
            "config": {"workload": "simple_swapping 512x512 batch=%d per GPU, generator+encoders forward only (BASELINE configs[1])" % B,
                       "recipe": "encode(content) + the E1/E2 of extract_feat_from_image(content | style) as one batch of 3B images; the two "
                                 "generator feature passes (extract_feat) and Rselfcorr as one batch of 2B; corrm + encode2 + decode.  Same "
                                 "work as the reference's command sequence, batched: nothing is skipped or cached",
                       "image_parallel": "1 batch per rank, no data-path collective"},
            "algorithmic_tflops_whole_job": swaps * FLOP_PER_SWAP / dt / 1e12,
            "roofline": {
                "kernel": "ppst_conv2d_mfma: conv_wino_kernel (3x3 stride-1 layers, Winograd F(2,3) along x) + conv_mfma_kernel + "
                          "conv_mfma2_kernel + conv1x1_stream_kernel + conv3x3_direct_kernel (StyledConv / EqualConv2d / nn.Conv2d "
                          "implicit GEMM, all launches)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_note": "bf16 dense MFMA 2500 TF / %d MFMA passes per algorithmic MAC (a builder-defined ceiling of the DIRECT form: "
                             "secondary since round 5, see frac_issued)" % passes,
                "frac_note": "achieved = ALGORITHMIC conv flop (2 * MACs of the direct form) / kernel time.  conv_wino_kernel issues 2/3 "
                             "of the direct form's MFMAs for the same algorithmic MACs (and the nine-product fused upscale, "
                             "ppst_conv_args.variant 11, 0.64 of its four-phase form's), so since round 4 this fraction and the matrix-pipe "
                             "busy counter (mfma_busy_frac_pmc) no longer move together: the counter is what the pipe did, frac is what "
                             "the path got done against the direct form's ceiling",
                "frac_vs_dense_bf16": achieved / PEAK_BF16_DENSE_TF, **issued_fields(detail),
                "launches": conv_launches, "kernel_ms_total": conv_ms,
                "share_of_step_time": conv_ms * 1e-3 / dt, "traffic": conv_traffic(), "traffic_source": conv_traffic_source(),
                "effective_clock_ghz_pmc": conv_pmc(), "mfma_busy_frac_pmc": conv_mfma_busy(),
            },
        }
        st_ach = st_flop / (st_ms * 1e-3) / 1e12 if st_ms > 0 else 0.0
        res["roofline_modulated_conv2d"] = {
            "kernel": "the same kernels, only the launches issued by StyledConv (stylegan2_layers.py:439-475: the generator's 3x3 and "
                      "fused-upscale convs with noise / bias / leaky-ReLU epilogue and instance-norm statistics) -- BASELINE metric (ii)",
            "bound": "mfma", "achieved": st_ach, "peak": peak, "unit": "TFLOP/s", "frac": st_ach / peak,
            "frac_vs_dense_bf16": st_ach / PEAK_BF16_DENSE_TF, **issued_fields([detail[i] for i in sorted(tag["styled"])]),
            "launches": len(tag["styled"]), "kernel_ms_total": st_ms,
            "share_of_step_time": st_ms * 1e-3 / dt}
        res["roofline_upfirdn2d"] = upfirdn2d_rate(B, dev)
        if world == 1 and not args.no_extras and args.precision == "bf16x3":
            res["extra"] = extras(args, dev)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(0)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
