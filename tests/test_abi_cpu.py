"""CPU: the C-ABI library builds, loads, and exports every symbol include/ppst_hip.h
declares; argument errors are reported without a GPU; host-side sharding logic."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ppst_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(ppst_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from ppst_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libppst_hip.so does not export %s" % n
    # and the binding covers the header one-to-one
    assert sorted(_lib.exported_symbols()) == names


def test_argument_errors_need_no_gpu():
    from ppst_amd._lib import lib
    assert lib.ppst_version() == 2          # PPST_ABI_VERSION (include/ppst_hip.h): bumped with the round-4 struct changes
    # unsupported dtype / null pointers / bad sizes are rejected before any launch
    assert lib.ppst_upfirdn2d(None, None, None, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 0, 0, 0, 0, 7, None) == -2    # no such dtype
    assert lib.ppst_upfirdn2d(None, None, None, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 0, 0, 0, 0, 1, None) == -3    # PPST_F16: valid, null data
    assert lib.ppst_fused_bias_act(None, None, None, None, 4, 1, 1, 3, 0, 0.2, 1.0, 9, None) == -2
    assert lib.ppst_upfirdn2d(None, None, None, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 0, 0, 0, 0, 0, None) == -3
    assert lib.ppst_fused_bias_act(None, None, None, None, -1, 1, 1, 3, 0, 0.2, 1.0, 0, None) == -1
    assert lib.ppst_fused_bias_act(None, None, None, None, 0, 1, 1, 3, 0, 0.2, 1.0, 0, None) == 0  # empty input
    assert lib.ppst_gemm_nt_f32(None, None, None, 1, 4, 4, 7, 1.0, None) == -1                      # K % 16
    # split-bf16 GEMMs: passes in {3, 6}, K % 32 (3 passes) / K % 16 (6 passes), empty batch is a no-op, null data after that
    assert lib.ppst_gemm_nt_split(None, None, None, 1, 4, 4, 32, 1.0, 4, None) == -1
    assert lib.ppst_gemm_nt_split(None, None, None, 1, 4, 4, 16, 1.0, 3, None) == -1
    assert lib.ppst_gemm_nt_split(None, None, None, 0, 4, 4, 16, 1.0, 6, None) == 0
    assert lib.ppst_gemm_nt_split(None, None, None, 1, 4, 4, 16, 1.0, 6, None) == -3
    assert lib.ppst_gemm_nn_split(None, None, None, 1, 4, 6, 32, 6, 6, 3, None) == -1                 # N % 4
    assert lib.ppst_gemm_nn_split(None, None, None, 1, 4, 8, 32, 8, 8, 3, None) == -3
    assert lib.ppst_softmax_rows(None, 0, 4096, 0.01, None) == 0
    assert lib.ppst_conv_tiles(512, 512, 16) == 1024 and lib.ppst_conv_tiles(17, 16, 8) == 3
    # entry points added later in the round: same contract
    assert lib.ppst_blur_nhwc(None, None, None, 1, 8, 8, 4, 3, 1, 1, 0, 1, 0, None, 3, None) == -3      # null data
    d = ctypes.c_void_p(16)   # a non-null token: validation happens before anything is dereferenced or launched
    assert lib.ppst_blur_nhwc(d, d, d, 1, 8, 8, 4, 3, 1, 1, 0, 1, 0, None, 3, None) == -1               # in_act without a table
    assert lib.ppst_blur_nhwc(d, d, d, 1, 8, 8, 6, 3, 1, 1, 0, 1, 0, None, 0, None) == -1               # C % 4
    assert lib.ppst_resample_u8(None, None, 1, 8, 8, 3, 4, 1, None, None, 5, None) == -3
    assert lib.ppst_resample_u8(None, None, 0, 8, 8, 3, 4, 1, None, None, 5, None) == 0                 # empty batch
    assert lib.ppst_u8_to_tensor(d, d, 1, 8, 8, 3, 0.5, 0.0, None) == -1                               # std == 0
    assert lib.ppst_head_tail(d, d, None, d, d, 1, 512, 512, 64, 64, 64, 64, 8, 3, 0, None) == -1       # D not in {1, 2}
    assert lib.ppst_l1_mean(None, None, None, None, 16, 1.0, None) == -3 and lib.ppst_l1_mean_ws(1 << 20) == 256 * 4
    assert lib.ppst_rscl_loss(d, d, d, d, d, d, 65, 0, 2048, 128, 0.07, None) == -1                     # n > 64
    assert lib.ppst_rscl_loss(d, d, None, d, d, d, 6, 6, 2048, 128, 0.07, None) == -3                   # k0 missing


def test_conv_entry_rejects_images_beyond_32bit_offsets():
    """The conv epilogues address one image with 32-bit element offsets: the entry point must refuse an image of >= 2^31
    elements (validation runs before anything is dereferenced or launched, so this needs no GPU)."""
    from ppst_amd import _lib
    a = _lib.ConvArgs()
    tok = ctypes.c_void_p(16)
    a.x = a.wpack = a.steps = a.y = tok
    a.B, a.in_h, a.in_w, a.in_ld = 0, 64, 64, 32          # B = 0: a valid call returns PPST_OK without launching
    a.out_h, a.out_w, a.out_ld, a.cout = 64, 64, 64, 64
    a.nsteps, a.n_groups, a.pad_mode, a.out_sy, a.out_sx = 9, 1, 0, 1, 1
    a.tile_h, a.tile_w, a.halo, a.bn, a.tile_rows, a.variant, a.precision = 64, 64, 1, 64, 16, 0, 0
    a.out_scale = 1.0
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == 0
    a.in_h = a.in_w = a.out_h = a.out_w = a.tile_h = a.tile_w = 8192
    a.in_ld = 4                                            # (input side small: 8192 * 8192 * 4 floats = 2^30 bytes)
    a.out_ld = 64                                          # 8192 * 8192 * 64 = 2^32 elements in one image
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == -1
    a.out_ld, a.cout = 28, 28                              # 8192 * 8192 * 28 < 2^31: accepted again
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == 0
    # the input side: the kernels address one input image with 32-bit BYTE offsets (buffer loads), also through the pixel stride
    # of a channel slice of a wide tensor
    a.in_ld = 8                                            # 8192 * 8192 * 8 floats = 2^31 bytes
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == -1
    a.in_ld = 4
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == 0
    # the exact-fp32 twin mirrors the checks (and wants 16-row tiles: its statistics layout assumes them)
    w = ctypes.c_void_p(16)
    f32 = lambda: _lib.lib.ppst_conv2d_f32(ctypes.byref(a), w, 9, 1, 3, 1, ctypes.c_float(1.0), w, w, w, None)
    assert f32() == 0
    a.in_ld = 8
    assert f32() == -1
    a.in_ld, a.out_ld, a.cout = 4, 64, 64
    assert f32() == -1
    a.out_ld, a.cout, a.tile_rows = 28, 28, 8
    assert f32() == -1
    a.tile_rows = 16
    a.variant, a.tile_rows, a.bn, a.early_a = 9, 16, 128, 1   # variant 9 needs tile_rows 24
    a.out_ld, a.cout = 128, 128
    a.in_h = a.in_w = a.out_h = a.out_w = a.tile_h = a.tile_w = 64
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == -1
    a.tile_rows = 24
    # ... and is an experiment kernel: accepted only by a PPST_EXPERIMENTS=1 build, refused by the production library
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == (0 if _lib.lib.ppst_has_experiments() else -1)
    a.variant, a.tile_rows, a.precision, a.halo = 0, 8, 0, 1          # the 8-row two-block form is a production kernel
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == 0
    for variant, rows, prec in ((1, 16, 0), (3, 16, 0), (7, 32, 0), (8, 16, 0), (0, 16, 4)):
        a.variant, a.tile_rows, a.precision, a.halo = variant, rows, prec, 1
        rc = _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None)
        assert rc == (0 if _lib.lib.ppst_has_experiments() else -1), (variant, rows, prec, rc)
    a.variant, a.tile_rows, a.precision = 0, 16, 0
    assert _lib.lib.ppst_conv2d_mfma(ctypes.byref(a), None) == 0


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: a non-GPU tensor raises like the reference's CHECK_CUDA."""
    from ppst_amd.stylegan2_op import fused_leaky_relu, upfirdn2d
    with pytest.raises(RuntimeError, match="CUDA"):
        upfirdn2d(torch.zeros(1, 1, 8, 8), torch.ones(3, 3))
    with pytest.raises(RuntimeError, match="CUDA"):
        fused_leaky_relu(torch.zeros(1, 4, 8, 8), torch.zeros(4))


def test_facade_dispatch_and_key_contract():
    from ppst_amd.ppst_model import PPSTModel
    m = PPSTModel(with_D=True)
    with pytest.raises(ValueError):
        m(torch.zeros(1), command=None)
    with pytest.raises(AttributeError):
        m(torch.zeros(1), command="no_such_command")
    from ppst_amd import weights as W
    sd = W.make_state_dict(3, with_nce=False)
    m.load_weights(sd)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_shard_pairs_partition():
    from ppst_amd.evaluation import shard_pairs
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += shard_pairs(8, 8, r, world)
        assert sorted(seen) == [(i, j) for i in range(8) for j in range(8)]
        sizes = [len(shard_pairs(8, 8, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ppst_amd.evaluation import shard_pairs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
mine = shard_pairs(4, 6, rank, world)
# stand-in for the per-pair work: a deterministic function of the pair
res = torch.tensor([[i, j, i * 100 + j] for i, j in mine], dtype=torch.int64)
pad = torch.full((24, 3), -1, dtype=torch.int64); pad[: len(res)] = res
out = [torch.empty_like(pad) for _ in range(world)]
dist.all_gather(out, pad)                      # result collection only: no data-path collective
t = torch.tensor([float(len(mine))]); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks
if rank == 0:
    rows = torch.cat(out); rows = rows[rows[:, 0] >= 0]
    got = sorted((int(a), int(b), int(c)) for a, b, c in rows)
    assert got == sorted((i, j, i * 100 + j) for i in range(4) for j in range(6)), got
    assert t.item() == 12.0
    print("OK")
dist.destroy_process_group()
'''


WORKER_GRID = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ppst_amd.evaluation import grid_exchange, shard_images, shard_pairs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
N, M, h = 5, 3, 64                     # real code-grid shapes: sp (.,256,64,64), fea||Rselfcorr (.,512,64,64)
ci, si = shard_images(N, M, rank, world)
def fab(tag, idx, C):                  # stand-in for the per-image passes: a deterministic function of the image
    if not idx:
        return None
    t = torch.zeros(len(idx), C, h, h)
    for n, i in enumerate(idx):
        t[n] = tag * 1000 + i + torch.arange(C).view(C, 1, 1) * 1e-3
    return t
local_c = (ci, fab(1, ci, 256), fab(2, ci, 512))
local_s = (si, fab(3, si, 512))
tc, ts = grid_exchange(local_c, local_s, N, M, world)
assert sorted(tc) == list(range(N)) and sorted(ts) == list(range(M)), (sorted(tc), sorted(ts))
for i in range(N):
    assert tc[i][0].shape == (1, 256, h, h) and tc[i][1].shape == (1, 512, h, h)
    assert torch.equal(tc[i][0], fab(1, [i], 256)) and torch.equal(tc[i][1], fab(2, [i], 512)), i
for j in range(M):
    assert torch.equal(ts[j], fab(3, [j], 512)), j
mine = shard_pairs(N, M, rank, world)
cnt = torch.tensor([float(len(mine))]); dist.all_reduce(cnt)
assert cnt.item() == N * M
# a rank that owns NO image (world > N + M; here one image for two ranks) still joins both gathers, with zero rows
ci1, si1 = shard_images(1, 0, rank, world)
assert (ci1, si1) == (([0], []) if rank == 0 else ([], []))
tc1, ts1 = grid_exchange((ci1, fab(1, ci1, 256), fab(2, ci1, 512)), (si1, None), 1, 0, world, like=(torch.device("cpu"), h, h))
assert sorted(tc1) == [0] and not ts1 and torch.equal(tc1[0][0], fab(1, [0], 256))
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


WORKER_DDP = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ppst_amd.train import ddp_average_
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
torch.manual_seed(rank)
g = torch.randn(1000)
ref = (torch.manual_seed(0), torch.randn(1000))[1] * 0.5 + (torch.manual_seed(1), torch.randn(1000))[1] * 0.5
out = ddp_average_(g.clone(), world)
assert torch.allclose(out, ref, atol=1e-6)
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_gradient_average(tmp_path):
    """The train step's only collective (flat gradient all-reduce + divide), world_size 2 on gloo."""
    script = tmp_path / "w2.py"
    script.write_text(WORKER_DDP)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29614")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


def test_two_rank_gloo_sharding(tmp_path):
    """world_size-2 rehearsal (gloo, CPU) of the image-parallel pair sharding + the timing
    reduction bench.py performs."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


def test_one_trainer_per_discriminator():
    """A second DiscriminatorTrainer on the same D rebinds D's parameters and would orphan the first trainer's
    flat / m / v (its Adam step would update memory D no longer reads).  for_network() keeps ONE trainer per D,
    compute_R1_loss goes through it, and adam() refuses to run once the aliasing is broken."""
    from ppst_amd.networks.discriminator import StyleGAN2Discriminator
    from ppst_amd.train import DiscriminatorTrainer
    D = StyleGAN2Discriminator(None, size=32)
    tr = DiscriminatorTrainer.for_network(D)
    assert DiscriminatorTrainer.for_network(D) is tr and tr.owns_parameters()
    w = next(iter(D.parameters()))
    tr.flat[0] = 123.0                       # the flat buffer IS the parameter storage
    assert float(w.view(-1)[0]) == 123.0
    other = DiscriminatorTrainer(D)          # what compute_R1_loss used to do
    assert other.owns_parameters() and not tr.owns_parameters()
    with pytest.raises(RuntimeError, match="no longer alias"):
        tr.adam()
    assert DiscriminatorTrainer.for_network(D) is other
    D.double()                               # _apply() moves the storage: the trainer must notice
    assert not other.owns_parameters()


def test_conv_plan_shape_checks_need_no_gpu():
    """ConvPlan.__call__ validates what the C ABI cannot see (the step table lives on the device)."""
    from ppst_amd import ops
    plan = ops.ConvPlan.__new__(ops.ConvPlan)
    plan.kind, plan.max_chan, plan.cout, plan.cin, plan.k = "conv", 32, 64, 64, 3

    class FakeCuda(torch.Tensor):
        is_cuda = True
    def fake(*shape):
        return torch.zeros(*shape).as_subclass(FakeCuda)
    with pytest.raises(RuntimeError, match="plan reads 64"):
        plan(fake(1, 16, 16, 32))
    with pytest.raises(RuntimeError, match="in_ss"):
        plan(fake(1, 16, 16, 64), in_ss=fake(1, 32, 2))
    with pytest.raises(RuntimeError, match="noise"):
        plan(fake(2, 16, 16, 64), noise=fake(1, 1, 16, 16), out=fake(2, 16, 16, 64))
    with pytest.raises(RuntimeError, match="residual"):
        plan(fake(1, 16, 16, 64), residual=fake(1, 16, 16, 32), out=fake(1, 16, 16, 64))
    with pytest.raises(RuntimeError, match="conv output must be"):
        plan(fake(1, 16, 16, 64), out=fake(1, 16, 16, 32))


def test_two_rank_gloo_grid_image_sharding_and_gather(tmp_path):
    """swapping_grid's exchange step (ONE padded all_gather_into_tensor each for the content rows sp | fea||Rselfcorr
    and the style rows) with the real code-grid shapes, world_size 2 on gloo: every rank ends with the full tables."""
    script = tmp_path / "wg.py"
    script.write_text(WORKER_GRID)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29615")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


def test_grid_sharding_work_per_rank():
    """8 x 8 grid on 8 ranks: 2 image passes + 8 pair passes per rank, every image / pair owned once."""
    from ppst_amd.evaluation import shard_images, shard_pairs
    for world in (1, 2, 3, 8):
        cs, ss, ps = [], [], []
        for r in range(world):
            ci, si = shard_images(8, 8, r, world)
            cs += ci; ss += si; ps += shard_pairs(8, 8, r, world)
            if world == 8:
                assert len(ci) + len(si) == 2 and len(shard_pairs(8, 8, r, world)) == 8
        assert sorted(cs) == list(range(8)) and sorted(ss) == list(range(8)) and len(set(ps)) == 64


def test_conv_kernel_choice_is_a_function_of_one_image():
    """ConvPlan.choose_kernel: which kernel family runs a launch.  All families give bit-identical outputs, but their tile
    statistics differ in the last bit, so the choice must depend on the plan and ONE image's geometry only (a shard of a batch
    reproduces the whole batch bit for bit).  Structural: the batch size enters only inside ``ops.batch_aware()`` (the train
    step: under-filled grids at batch 2), and is ignored outside it.  Behavioural: the production table of the swap path --
    which layers go where -- and the contract that experiments stay off."""
    from ppst_amd import ops
    assert not ops.BATCH_AWARE["value"]

    def plan(kind, cin, cout, k, nsteps, halo, n_groups=1, precision=0, early_a=1):
        p = ops.ConvPlan.__new__(ops.ConvPlan)
        p.kind, p.cin, p.cout, p.k, p.nsteps, p.halo, p.n_groups = kind, cin, cout, k, nsteps, halo, n_groups
        p.precision, p.early_a, p.bn = precision, early_a, (128 if cout >= 128 else 64)
        return p
    ck = lambda p, s, osy=1, out=None: p.choose_kernel(s, s, *(out or (s, s)), s, s, osy)
    # round 4: plain 3x3 stride-1 layers with Cout >= 128 run the Winograd kernel (variant 10) when one image gives >= 16 blocks,
    # whatever the batch; the fused upscale with Cout = 128 runs the N-256 kernel as two phase pairs ("dual")
    assert ops.WINO["value"] and ops.DUAL_CONVT["value"]
    assert ck(plan("conv", 256, 256, 3, 72, 1), 256) == (10, 128, 16) and ck(plan("conv", 128, 128, 3, 36, 1), 512) == (10, 128, 16)
    assert ck(plan("conv", 256, 256, 3, 72, 1), 64) == (10, 128, 16)             # 16 tiles x 2 N tiles per image
    assert ck(plan("conv", 128, 128, 3, 36, 1), 32) == (0, 128, 16)              # 4 blocks per image: stays direct
    assert ck(plan("dgrad", 256, 128, 3, 72, 1), 128) == (10, 128, 16)           # the input gradient of a 3x3 conv is one too
    assert ck(plan("conv", 128, 64, 3, 36, 1), 512)[0] != 10 and ck(plan("conv", 256, 256, 3, 72, 1, precision=3), 256)[0] != 10
    assert ck(plan("s2d", 64, 128, 3, 20, 1), 128)[0] != 10 and ck(plan("convT", 512, 256, 3, 64, 1, n_groups=4), 128, osy=2, out=(256, 256))[0] != 10
    for B in (1, 2, 16):
        assert plan("conv", 256, 256, 3, 72, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, B) == (10, 128, 16)
    pd = plan("convT", 256, 128, 3, 32, 1, n_groups=4)
    pd.steps_dual = object()
    assert ck(pd, 256, osy=2, out=(512, 512)) == ("dual", 256, 16)
    assert ck(pd, 32, osy=2, out=(64, 64)) == (0, 128, 16)                        # 8 blocks per image: the four-group tile kernel
    # the fused upscale as nine products per input pixel (variant 11) where one image gives >= 64 blocks of 15 x 15 positions
    assert ops.UP9["value"]
    for cout, ng in ((128, 4), (256, 4), (512, 4)):
        pu = plan("convT", 256, cout, 3, 32, 1, n_groups=ng)
        pu.steps_dual, pu.steps_up9 = object(), object()
        assert ck(pu, 256, osy=2, out=(512, 512)) == ("up9", 256, 15)
        assert ck(pu, 30, osy=2, out=(60, 60))[0] != "up9"                        # 4 x cout / 64 blocks per image
        assert ck(pu, 64, osy=2, out=(128, 128))[0] != "up9" and ck(pu, 128, osy=2, out=(256, 256))[0] == "up9"   # 64 = 4.27 tiles of 15
        assert ck(plan("convT", 256, cout, 3, 32, 1, n_groups=ng, precision=3), 256, osy=2, out=(512, 512))[0] != "up9"
    # the DIRECT kernels' table (what runs with the three switches off, and what the bit-identity tests compare)
    prev = ops.WINO["value"], ops.DUAL_CONVT["value"], ops.UP9["value"]
    ops.WINO["value"] = ops.DUAL_CONVT["value"] = ops.UP9["value"] = False
    try:
        _direct_kernel_table(ops, plan, ck)
    finally:
        ops.WINO["value"], ops.DUAL_CONVT["value"], ops.UP9["value"] = prev


def _direct_kernel_table(ops, plan, ck):
    # the generator's wide layers: N-256 tile where one image still gives >= 32 blocks
    assert ck(plan("conv", 256, 256, 3, 72, 1), 256) == (2, 256, 16)
    assert ck(plan("conv", 512, 512, 3, 144, 1), 128) == (2, 256, 16)
    assert ck(plan("conv", 256, 256, 3, 72, 1), 64) == (0, 128, 16)              # 16 tiles per image: stays on the tile kernel
    assert ck(plan("convT", 512, 256, 3, 64, 1, n_groups=4), 128, osy=2, out=(256, 256)) == (2, 256, 16)
    # Cout = 128 layers: the tile kernel (the measured experiments are off)
    assert ck(plan("conv", 128, 128, 3, 36, 1), 512) == (0, 128, 16)
    # 1x1 -> streaming kernel; thin 3x3 -> register-reuse form; thin stride-2 -> direct form
    assert ck(plan("conv", 128, 64, 1, 4, 0, early_a=0), 512) == (4, 64, 16)
    assert ck(plan("conv", 256, 128, 1, 8, 0, early_a=0), 256) == (4, 64, 16)
    assert ck(plan("conv", 32, 32, 3, 9, 1), 512) == (6, 64, 16)
    assert ck(plan("s2d", 32, 64, 3, 10, 1), 256) == (5, 64, 16)
    assert ck(plan("s2d", 64, 128, 3, 20, 1), 128) == (0, 128, 16)               # Cout 128: tile kernel
    # the batch changes nothing outside the train step ...
    for B in (1, 2, 16):
        assert plan("conv", 256, 256, 3, 72, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, B) == (0, 128, 16)
        assert plan("conv", 512, 512, 3, 144, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, B) == (2, 256, 16)
    # ... inside it an under-filled launch takes twice the N tiles (tile kernel instead of N-256) and -- round 5 -- keeps its 16-row
    # tiles: the launch splits K across blocks instead (ops.KSPLIT, ppst_conv_args.ksplit); with the split off, twice the M tiles (8 rows)
    prev_ks = ops.KSPLIT["value"]
    try:
        with ops.batch_aware():
            ops.KSPLIT["value"] = True
            assert plan("conv", 256, 256, 3, 72, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, 2) == (0, 128, 16)
            assert plan("conv", 512, 512, 3, 144, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, 2) == (0, 128, 16)
            ops.KSPLIT["value"] = False
            assert plan("conv", 256, 256, 3, 72, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, 2) == (0, 128, 8)
            assert plan("conv", 512, 512, 3, 144, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, 2) == (0, 128, 8)
            assert plan("conv", 512, 512, 3, 144, 1).choose_kernel(64, 64, 64, 64, 64, 64, 1, 16) == (2, 256, 16)
            assert plan("conv", 128, 128, 3, 36, 1).choose_kernel(512, 512, 512, 512, 512, 512, 1, 2) == (0, 128, 16)
    finally:
        ops.KSPLIT["value"] = prev_ks
    # the split itself: the largest S of 8 / 4 / 2 that fills at most 256 blocks, cut at chunk starts, >= min_steps per share
    assert ops._ksplit_choice(64, list(range(0, 73, 9)), 256, 16) == (4, [0, 18, 36, 54, 72])
    assert ops._ksplit_choice(4, list(range(0, 145, 9)), 256, 16) == (8, [0, 18, 36, 54, 72, 90, 108, 126, 144])
    assert ops._ksplit_choice(4, list(range(0, 145, 9)), 256, 16, 4) == (4, [0, 36, 72, 108, 144])
    assert ops._ksplit_choice(256, list(range(0, 73, 9)), 256, 16) == (0, None)          # a full grid is left alone
    assert ops._ksplit_choice(64, [0, 9, 18], 256, 16) == (0, None)                      # 9 steps per share: too short
    s2d = [4 * i for i in range(16)] + [64 + 2 * i for i in range(48)] + [160]           # unequal chunks: 4 / 2 / 2 / 2 taps per phase
    S_, cuts_ = ops._ksplit_choice(4, s2d, 256, 16)
    assert S_ == 8 and cuts_ == [0, 20, 40, 60, 80, 100, 120, 140, 160] and all(c in s2d for c in cuts_)
    assert not ops.BATCH_AWARE["value"]
    # single-pass modes: the N-256 and streaming kernels are built for them, the experiments are not
    assert ck(plan("conv", 256, 256, 3, 72, 1, precision=3), 256) == (2, 256, 16)
    assert ck(plan("conv", 32, 32, 3, 9, 1, precision=1), 512) == (6, 64, 16)
    # fp16x2 experiment and the exact-fp32 verification mode: tile kernel only
    assert ck(plan("conv", 256, 256, 3, 72, 1, precision=4), 256) == (0, 128, 16)
    assert ck(plan("conv", 32, 32, 3, 9, 1, precision=2), 512) == (0, 64, 16)
    assert not ops.TWO_BLOCK_128["value"] and not ops.TALL_TILE_128["value"] and ops.DIRECT_MAX["cout3x3"] == 64


def test_gemm_mode_selection():
    """ops.gemm_nt / gemm_nn pick the kernel from the call site's mode, the conv precision and K (no launch here):
    cosine logits -> six passes over three bf16 planes (fp32-class), products with the softmax matrix -> three passes,
    exact-conv mode (precision 2) and K the split kernels do not take -> the fp32 MFMA kernel."""
    from ppst_amd import ops
    assert ops._gemm_passes(None, 512) == 6 and ops._gemm_passes("x6", 4608) == 6
    assert ops._gemm_passes("x3", 4096) == 3
    assert ops._gemm_passes("x3", 48) == 6            # K % 32 != 0: the six-pass form takes K % 16
    assert ops._gemm_passes("x3", 24) == 0 and ops._gemm_passes(None, 8) == 0
    assert ops._gemm_passes("f32", 512) == 0
    prev = ops.PRECISION["value"]
    try:
        ops.PRECISION["value"] = 2
        assert ops._gemm_passes("x3", 4096) == 0 and ops._gemm_passes(None, 512) == 0
    finally:
        ops.PRECISION["value"] = prev
    ops.GEMM_MODE["value"] = "f32"
    try:
        assert ops._gemm_passes("x3", 4096) == 0
    finally:
        ops.GEMM_MODE["value"] = None
    with pytest.raises(ValueError):
        ops._gemm_passes("bf16", 64)
