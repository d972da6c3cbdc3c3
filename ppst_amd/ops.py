"""Torch-facing wrappers over the C ABI (include/ppst_hip.h).

PyTorch is used here for device memory and streams only: every arithmetic
operation of the path is a HIP kernel in libppst_hip.so.  Activations of the
fused path are NHWC fp32 tensors of shape (B, H, W, C) (or channel-slice views
of such tensors: pixel stride ``ld`` = stride(2)).
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import lib, check

PAD_ZERO, PAD_REFLECT, PAD_REPLICATE = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_PRELU = 0, 1, 2
RES_BEFORE_ACT = 1 << 8

PRECISION = {"value": 0}  # 0 = bf16x3 (fp32-class), 1 = single-pass bf16
TILE_ROWS = {"value": 16}  # conv tile rows: 8 (2 blocks per CU) or 16 (1 block per CU)
# 1: layers whose step table allows it run on the fat-wave kernel (conv_mfma2.hip: 4 waves, 128-px x 64/128-ch wave
# tiles, N tile 128 or 256); 0 (default): everything on the 8-wave kernel.  Measured on MI355X (round 2, same process
# class of runs, tests/conv_table.py): the fat-wave kernel is bit-identical but SLOWER -- 256->256@256^2 395 vs 417 TF/s
# with the 256-channel tile, 128->128@512^2 261 vs 356 with the 128-channel tile; whole step 284-294 vs 330 TF/s: one
# in-order wave per SIMD exposes its own LDS / wait latency, which two co-resident waves hide for each other.
# 2 (default since round 2): layers with Cout % 256 == 0 whose grid still covers the chip run on the 8-wave kernel with
# 128 px x 64 ch wave tiles and a 256-channel N tile (conv_mfma2.hip, WNW = 4): bit-identical to the 64x64-wave-tile kernel
# and 10-16 % faster on those layers (256->256 @256^2 459 vs 417 TFLOP/s, 512->512 @128^2 494 vs 440).
CONV_VARIANT = {"value": 2}
# the fused upscale with Cout % 128 == 0 (and not % 256: those run the N-256 kernel as four phases) as two phase pairs on the N-256
# kernel (ppst_conv_args.dual_b): 256 -> 128 up 2 was the slowest StyledConv layer on the tile kernel (0.42 of the ceiling)
DUAL_CONVT = {"value": True, "min_blocks": 32}
# the fused upscale (Cout % 64 == 0) as the un-blurred 3x3 transposed conv + 2x2 box sum in the epilogue (ppst_conv_args.variant 11,
# conv_mfma2.hip UP9): nine products per input pixel instead of the sixteen of the 4x4 kernel; fp32-class, not bit-identical to the
# four-phase forms (the bit-identity tests of those run with it off)
# (blocks cover 15 x 15 input positions: an extent that fills its last tile badly -- 64 = 4.27 tiles -- stays on the four-phase form)
UP9 = {"value": True, "min_blocks": 64, "min_fill": 0.85}
DIRECT_MAX = {"cout": 64, "nsteps": 40, "cout3x3": 64}   # cout3x3 = 128 was measured: 128->128 @512^2 230 vs 357 TFLOP/s
# thin layers (few channels in and out) on the direct form of that kernel
WGRAD_SPLIT = {"blocks": 1024, "min_tiles": 4}   # conv_wgrad: target block count of a launch, fewest pixel tiles per block
WGRAD_X3 = {"value": True}         # conv weight gradients on the bf16 matrix pipe (hi/lo split), exact fp32 with precision 2
STREAM_1X1 = {"value": True}       # 1x1 convs (halo 0) on the streaming kernel of conv1x1.hip
# variant 8 (conv_ksplit.hip: the block's waves split K in two, 128 px x 64 ch wave tiles, two activation slots) for the
# Cout = 128-class layers when one image gives >= min_blocks: measured 3-7 % SLOWER than the tile kernel (128->128 @512^2
# 342-348 vs 358-368 TFLOP/s; in-kernel trace: step pair 5 750 cycles against an issue floor of 3 072) and not bit-identical -- off
# round 5: across-block K split (ppst_conv_args.ksplit) of launches whose grid fills a fraction of the chip while every block runs
# one long serial chain of steps -- the 64^2 ... 4^2 layers of a train step at batch 2 (4-128 blocks of 72-160 steps: ~0.9 us per
# step whatever the grid).  S = the largest of 8 / 4 / 2 with S x blocks <= max_blocks, chunk count divisible by S and >= min_steps
# steps left per block.  Results equal the unsplit launch's up to fp32 summation order: batch-aware passes (the train step) only.
# PPST_KSPLIT=0 turns it off.
KSPLIT = {"value": os.environ.get("PPST_KSPLIT", "1") != "0", "max_blocks": 256, "min_steps": 16, "variants": (0, 2, 10)}
KSPLIT_128 = {"value": False, "min_blocks": 32}
# variant 7 (32 x 16 px x 128 ch blocks, conv_mfma2.hip WMW = 4) when one image gives >= min_blocks: measured 3-5 % SLOWER than
# the tile kernel on the Cout = 128 layers (one activation slot: the chunk store sits between two barriers; 33-44 spills) -- off
TALL_TILE_128 = {"value": False, "min_blocks": 32}
# the same geometry in the single-pass modes (two activation slots fit there): the Cout = 128-class 3x3 layers, which the tile
# kernel runs at 0.22 of the single-pass ceiling (128 -> 128 @1024^2 fp16)
TALL_TILE_SINGLE = {"value": True, "min_blocks": 64}
# single-pass modes on half-stored activations: 64 input channels per step on the N-256 kernel family (ppst_conv_args.k64) -- half the
# steps (barriers, fragment waits, DMA issues) per MFMA; the Cout = 128-class layers then take 24 x 16 px tiles (variant 9)
K64 = {"value": True}
# variant 9 (24 x 16 px x 128 ch blocks, wave tile 96 px x 64 ch, TWO activation slots: conv_mfma2.hip MT_ = 6) for the Cout = 128-class
# layers whose tile height wastes <= max_waste of the rows (512 -> 528, 256 -> 264: 3.1 %; 128 -> 144 would be 12.5 %)
TILE24_128 = {"value": False, "min_blocks": 32, "max_waste": 0.04}
TWO_BLOCK_128 = {"value": False}   # experiment: variant 3 (see __call__) for the Cout = 128-class layers
# round 3: the tile kernel itself as two 4-wave blocks per CU (8 x 16 px x 128 ch tiles, 64 px x 64 ch waves, TWO activation slots,
# 77 KB of LDS each) for the Cout = 128-class layers: one block's prologue / epilogue / barrier waits overlap the other's MFMAs
TWO_BLOCK_8ROW = {"value": False, "min_blocks": 64}
# Batch-aware kernel choice (the TRAINING path only: the trainers switch it on around their passes).  The inference recipes keep
# the batch out of the choice on purpose (a shard of a batch reproduces the whole batch bit for bit); the train step runs 64 x 64
# layers at batch 1-4, where one launch has 32-128 blocks for 256 CUs: with fewer than ``fill`` blocks a Cout % 256 layer leaves
# the N-256 kernel for the tile kernel (twice the N tiles) and the tile kernel takes its 8-row form (twice the M tiles, two blocks
# per CU).  Outputs are bit-identical across these forms; tile statistics differ in the last bit.
BATCH_AWARE = {"value": False, "fill": 192}


class batch_aware:
    """``with ops.batch_aware():`` -- conv kernel choice may look at the batch size inside the block (the train step)."""

    def __enter__(self):
        self.prev = BATCH_AWARE["value"]
        BATCH_AWARE["value"] = True

    def __exit__(self, *exc):
        BATCH_AWARE["value"] = self.prev
# variant 10 (conv_wino.hip): Winograd F(2,3) along x for the plain 3x3 stride-1 layers with Cout >= 128 -- 1.5x fewer MFMAs per
# output; fp32-class (<= 3e-5 against float64) but not bit-identical to the direct kernels.  ``min_blocks``: blocks ONE image gives
# (the choice stays a function of the plan and one image's geometry).
# ``fill`` (batch-aware passes only, i.e. the train step): with fewer than this many blocks in the LAUNCH the choice falls through to
# the direct kernels' under-filled forms (8-row two-block tiles: twice the blocks)
# ``ksplit_fill`` (batch-aware passes with ops.KSPLIT on): a Cout <= 256 layer whose launch has at most this many Winograd blocks runs
# on the tile kernel with the across-block K split instead (256 -> 256 @64^2 x 2: 38 us with S = 4 against 48 on the Winograd kernel
# with S = 2 -- its 128 accumulator registers per thread make the hand-over twice as large; tests/conv_ksplit_time.py)
WINO = {"value": True, "min_blocks": 16, "fill": 0, "ksplit_fill": 64}
# round 5: apply passes whose only consumer is a 1x1 conv are applied by that conv while it loads -- layert1's resnet merge by
# layert1.1 (ConvPlan in_res), the last upsampling block's merge by ToRGB in the image pass (torgb_apply).  Off: the round-4 passes.
# "torgb": measured SLOWER (rocprofv3, batch-8 swap step: 1.09 ms against 0.56 + 0.23 for the pass + the conv -- the 32-lanes-per-pixel ToRGB
# kernel becomes instruction-bound with the bilinear skip sampled per element) -- built, tested (t_fuse_tail), off
FUSE_TAIL = {"value": os.environ.get("PPST_FUSE_TAIL", "1") != "0", "torgb": os.environ.get("PPST_FUSE_TORGB", "0") == "1"}
FAT_MIN_BLOCKS = 32         # take the 256-channel tile only when ONE image still gives >= this many blocks (B = 8: one per CU)


# the measured-and-off conv forms exist only in a PPST_EXPERIMENTS=1 build of the library (ppst_amd/build.py)
EXPERIMENTS = bool(lib.ppst_has_experiments())


def _need_experiments(what):
    if not EXPERIMENTS:
        raise RuntimeError("%s is an experiment kernel: rebuild the library with PPST_EXPERIMENTS=1 (python -m ppst_amd.build)" % what)


def set_precision(p):
    """0: bf16x3 (fp32-class, the measured path); 1: single-pass bf16; 2: exact fp32 MFMA (verification only, slow);
    3: single-pass fp16 (the "fp16 generator" of BASELINE configs[4]; fp32 accumulate / statistics / StyleMod);
    4: two-pass fp16 (activation hi + lo, weight rounded once to fp16): a measured experiment, 2/3 of the MFMAs of mode 0."""
    assert p in (0, 1, 2, 3, 4)
    if p == 4:
        _need_experiments("precision 4 (two-pass fp16)")
    PRECISION["value"] = p


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current stream on the current device as a raw hipStream_t.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers: ~9 us per call, 2000 calls per training step.)"""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name="tensor"):
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32, got %s" % (name, t.dtype))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_ST = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}       # PPST_ST_* of include/ppst_hip.h
# Half-precision activation storage (round 4, ppst_conv_args.io_st and the `_st` entry points): in the single-pass precision modes
# the NHWC activations of the inference networks (E1, E2, the generator without feature extraction) live in HBM in the operand
# type of the mode -- IEEE half in mode 3, bfloat16 in mode 1 -- instead of fp32: every elementwise pass, blur and HBM-bound conv
# of those paths moves half the bytes.  Accumulation, statistics and the (scale, shift) pairs stay fp32; a kernel rounds once, at
# its store.  Off: fp32 storage in every mode (the round-3 behaviour).  Training keeps fp32 storage either way.
HALF_STORE = {"value": os.environ.get("PPST_HALF_STORE", "1") != "0"}     # (the env switch: A/B runs of bench.py / the tests)


def act_dtype():
    """storage type of the inference networks' activations in the current precision mode"""
    if HALF_STORE["value"] and not torch.is_grad_enabled():
        return {1: torch.bfloat16, 3: torch.float16}.get(PRECISION["value"], torch.float32)
    return torch.float32


# Round 5: half-precision storage of the TRAINING activations and their gradients in precision mode 1 (BASELINE configs[3]'s "bf16": the
# accuracy class of bf16 autocast, whose conv outputs and activation gradients are bfloat16 tensors).  The differentiable networks
# (train_g.py) and the discriminator tape (train.py) start their trunks in train_dtype(); every kernel downstream keeps its input's
# type, gradients take the type of the tensor they belong to.  Parameters, their gradients, Adam, statistics, pooled vectors, style
# codes and the whole correspondence branch stay fp32.
TRAIN_HALF = {"value": os.environ.get("PPST_TRAIN_HALF", "1") != "0"}


def train_dtype():
    return torch.bfloat16 if (TRAIN_HALF["value"] and PRECISION["value"] == 1) else torch.float32


def _chk_act(t, name="activation"):
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor" % name)
    if t.dtype not in _ST:
        raise RuntimeError("%s must be float32, float16 or bfloat16, got %s" % (name, t.dtype))


def _nhwc_ld(t, name="activation", half_ok=False):
    """validate a (B,H,W,C) tensor or channel-slice view; return pixel stride (elements).  half_ok: the caller's kernel takes a
    storage type (fp16 / bf16 tensors allowed); everything else insists on float32."""
    if half_ok:
        _chk_act(t, name)
    else:
        _chk(t, name)
    B, H, W, C = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != W * ld or (B > 1 and t.stride(0) != H * W * ld):
        raise RuntimeError("%s must be NHWC-contiguous (or a channel slice of one)" % name)
    return ld


def empty_nhwc(B, H, W, C, like):
    return torch.empty((B, H, W, C), device=like.device, dtype=torch.float32)


# ---------------------------------------------------------------- layout ----
def nchw_to_nhwc(x):
    _chk(x, "x")
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    check(lib.ppst_nchw_to_nhwc(_p(x), _p(y), B, C, H, W, _stream()), "ppst_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x):
    ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    if ld != C:
        raise RuntimeError("nhwc_to_nchw needs a dense tensor")
    y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    check(lib.ppst_nhwc_to_nchw(_p(x), _p(y), B, C, H, W, _stream()), "ppst_nhwc_to_nchw")
    return y


# ------------------------------------------------------ upfirdn2d / blur ----
_DT_F64 = 3          # PPST_F64: the two native ops only (every tensor of the call double, arithmetic in double)


def _chk_op(t, name):
    """tensors of the two native ops: float32 / float16 / bfloat16 / float64 (AT_DISPATCH_FLOATING_TYPES_AND_HALF)"""
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor" % name)
    if t.dtype not in _ST and t.dtype != torch.float64:
        raise RuntimeError("%s must be float32, float16, bfloat16 or float64, got %s" % (name, t.dtype))


def upfirdn2d_raw(x4, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """x4: [major, H, W, minor] contiguous (upfirdn2d.cpp:4-23). Returns [major, oh, ow, minor].  float32, float16, bfloat16
    or float64 like the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (upfirdn2d_kernel.cu:225); for the 16- and 32-bit types
    the taps are used as fp32 and the products accumulate in fp32 (the reference accumulates in the tensor's type), rounded once
    to the input's type; float64: taps and accumulation in double."""
    _chk_op(x4, "input")
    _chk_op(kernel, "kernel")
    x4 = x4.contiguous()
    f64 = x4.dtype == torch.float64
    kernel = (kernel.double() if f64 else kernel.float()).contiguous()
    major, in_h, in_w, minor = x4.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    y = torch.empty((major, out_h, out_w, minor), device=x4.device, dtype=x4.dtype)
    check(lib.ppst_upfirdn2d(_p(x4), _p(kernel), _p(y), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
                             px0, px1, py0, py1, _DT_F64 if f64 else _ST[x4.dtype], _stream()), "ppst_upfirdn2d")
    return y


def blur_nhwc(x, kernel, pad0, pad1, pad_mode=PAD_ZERO, down=1, s2d=False, in_ss=None, in_act=ACT_NONE):
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    if ld != C:
        raise RuntimeError("blur_nhwc needs a dense tensor")
    _chk(kernel, "kernel")
    ks = kernel.shape[0]
    oh = (H + pad0 + pad1 - ks + down) // down
    ow = (W + pad0 + pad1 - ks + down) // down
    if s2d:
        y = torch.empty((B, (oh + 1) // 2, (ow + 1) // 2, 4 * C), device=x.device, dtype=x.dtype)
    else:
        y = torch.empty((B, oh, ow, C), device=x.device, dtype=x.dtype)
    _chk(in_ss, "in_ss")
    check(lib.ppst_blur_nhwc_st(_p(x), _p(kernel.contiguous()), _p(y), B, H, W, C, ks, pad0, pad1, pad_mode, down,
                                1 if s2d else 0, _p(in_ss), in_act, _ST[x.dtype], _stream()), "ppst_blur_nhwc")
    return y, (oh, ow)


def fused_bias_act_raw(x, b, ref, act, grad, alpha, scale):
    """fused.fused_bias_act (fused_bias_act.cpp:4-20): empty tensor == absent.  float32, float16 or bfloat16 (the reference:
    AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79): bias and refer take the input's type, fp32 arithmetic,
    one rounding.  float64 (round 5): arithmetic in double."""
    _chk_op(x, "input")
    x = x.contiguous()
    b = None if (b is None or b.numel() == 0) else b.to(x.dtype).contiguous()
    ref = None if (ref is None or ref.numel() == 0) else ref.to(x.dtype).contiguous()
    _chk_op(b, "bias")
    _chk_op(ref, "refer")
    y = torch.empty_like(x)
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    size_b = b.numel() if b is not None else 1
    check(lib.ppst_fused_bias_act(_p(x), _p(b), _p(ref), _p(y), x.numel(), step_b, size_b, act, grad, float(alpha),
                                  float(scale), _DT_F64 if x.dtype == torch.float64 else _ST[x.dtype], _stream()), "ppst_fused_bias_act")
    return y


# ------------------------------------------------------------- fused conv ----
class ConvPlan:
    """Packed weights + step table for ppst_conv2d_mfma.

    kind: 'conv' (k in {1,3}, stride 1), 's2d' (3x3 stride 2 over a space-to-depth
    input), 'convT' (fused 4x4 stride-2 transposed conv = 4 output phases of 2x2 taps).
    weight: (Cout, Cin, k, k) fp32 CUDA tensor; scale multiplies the weights.
    """

    # Step tables are a function of (kind, weight shape, chan_base, device) only; building one is a Python loop over up to a few
    # hundred steps plus three host-to-device copies from pageable memory.  A training step rebuilds every plan (the weights
    # changed), ~150 per D + G iteration: the tables are shared between plans of one geometry, only the weights are new.
    _GEOMETRY = {}
    _GEOM_ATTRS = ("kind", "cout", "cin", "k", "bn", "n_groups", "halo", "src", "max_chan", "wstrides", "nsteps", "flop_steps",
                   "early_a", "ksplit_ok", "chunk_starts0", "chunk_starts0_k64", "steps", "src_dev", "chunk_start", "chunks_per_group", "w4_shape", "max_chunk_steps",
                   "min_chunk_steps", "full_cover", "steps_dual", "src_dual", "steps_up9", "steps_k64", "src_k64", "steps_dual_k64",
                   "src_dual_k64")

    def __init__(self, weight, kind="conv", scale=1.0, chan_base=0, precision=None):
        _chk(weight, "weight")
        w = weight.detach().contiguous()
        gkey = (kind, tuple(w.shape), int(chan_base), str(w.device))
        geom = ConvPlan._GEOMETRY.get(gkey)
        if geom is None:
            self._build(w, kind, scale, chan_base, precision)
            ConvPlan._GEOMETRY[gkey] = {a: getattr(self, a) for a in ConvPlan._GEOM_ATTRS if hasattr(self, a)}
            return
        self.__dict__.update(geom)
        self.precision = PRECISION["value"] if precision is None else precision
        self.wparam, self.up_scale = w, float(scale)       # (repack_plans: the fp32 parameter storage this plan was packed from)
        cout, cin, k, _ = w.shape
        if kind in ("convT", "dgradT"):      # the fused upscale's 4x4 kernel (Cin, Cout, 4, 4), scale folded in
            wsrc = torch.empty((cin, cout, 4, 4), device=w.device, dtype=torch.float32)
            check(lib.ppst_upscale_weight(_p(w), _p(wsrc), cout, cin, float(scale), _stream()), "ppst_upscale_weight")
            if kind == "dgradT":
                self.fwd_scale = float(scale)
            scale = 1.0
        elif kind == "dgrad":
            wsrc = w.view(-1)[k * k - 1:]
        elif kind == "dgrad_s2ds":
            wsrc = torch.empty((4 * cin, cout, 2, 2), device=w.device, dtype=torch.float32)
            check(lib.ppst_dgrad_s2d_stack_weight(_p(w), _p(wsrc), cout, cin, _stream()), "ppst_dgrad_s2d_stack_weight")
        else:
            wsrc = w
        self.scale = float(scale)
        self.wsrc = wsrc
        if self.precision == 2:
            self.wpack = None
            return
        self._packs = {}
        self.wpack = self.pack_for(self.bn)

    def _build(self, w, kind, scale, chan_base, precision):
        self.kind = kind
        self.wparam, self.up_scale = w, float(scale)
        self.precision = PRECISION["value"] if precision is None else precision
        cout, cin, k, _ = w.shape
        assert cin % 32 == 0, "fused conv needs Cin % 32 == 0 (got %d)" % cin
        self.cout, self.cin, self.k = cout, cin, k
        self.bn = 128 if cout >= 128 else 64
        steps, src = [], []
        nchunk = cin // 32
        if kind == "conv":
            assert k in (1, 3)
            self.n_groups = 1
            self.halo = 0 if k == 1 else 1
            for c in range(nchunk):
                first = True
                for ky in range(k):
                    for kx in range(k):
                        steps.append((chan_base + 32 * c, ky - k // 2, kx - k // 2, 1 if first else 0))
                        src.append((32 * c, ky, kx))
                        first = False
            sn, sc, sy, sx = cin * k * k, k * k, k, 1
            wsrc = w
        elif kind == "s2d":
            assert k == 3
            self.n_groups = 1
            self.halo = 1
            for py in range(2):
                for px in range(2):
                    for c in range(nchunk):
                        first = True
                        for ey in range(2):
                            for ex in range(2):
                                ky, kx = 2 * ey + py, 2 * ex + px
                                if ky > 2 or kx > 2:
                                    continue
                                steps.append(((py * 2 + px) * cin + 32 * c, ey, ex, 1 if first else 0))
                                src.append((32 * c, ky, kx))
                                first = False
                        if py == 1 and px == 1:
                            # the (1,1) phase has a single tap: pad the chunk with a zero-weight step
                            # (8-row tile kernels need >= 2 steps per chunk, ppst_hip.h tile_rows)
                            steps.append(((py * 2 + px) * cin + 32 * c, 0, 0, 0))
                            src.append((-1, 0, 0))
            sn, sc, sy, sx = cin * 9, 9, 3, 1
            wsrc = w
        elif kind == "convT":
            assert k == 3
            self.n_groups = 4
            self.halo = 1
            # F.conv_transpose2d(x, w4, stride=2, padding=1): oy = 2*iy - 1 + ky
            wsrc = torch.empty((cin, cout, 4, 4), device=w.device, dtype=torch.float32)
            check(lib.ppst_upscale_weight(_p(w), _p(wsrc), cout, cin, float(scale), _stream()), "ppst_upscale_weight")
            scale = 1.0
            taps = {0: [(-1, 3), (0, 1)], 1: [(0, 2), (1, 0)]}
            for a in range(2):
                for b in range(2):
                    for c in range(nchunk):
                        first = True
                        for dy, ky in taps[a]:
                            for dx, kx in taps[b]:
                                steps.append((chan_base + 32 * c, dy, dx, 1 if first else 0))
                                src.append((32 * c, ky, kx))
                                first = False
            sn, sc, sy, sx = 16, cout * 16, 4, 1
            if cout % 128 == 0 and chan_base % 1 == 0:
                # the same conv as TWO row phases whose N tile holds both column phases (ppst_conv_args.dual_b): a step is a tap row
                # dy with one tap column per column phase -- the per-element tap order stays (dy major), outputs bit-identical
                dsteps, dsrc = [], []
                for a in range(2):
                    for c in range(nchunk):
                        first = True
                        for dy, ky in taps[a]:
                            for j in range(2):
                                (dx0, kx0), (dx1, kx1) = taps[0][j], taps[1][j]
                                dsteps.append((chan_base + 32 * c, dy, (dx0 + 1) | ((dx1 + 1) << 8), 1 if first else 0))
                                dsrc.append((32 * c, ky, kx0 | (kx1 << 8)))
                                first = False
                self._dual_tmp = (dsteps, dsrc)
            if cout % 64 == 0:
                # variant 11: per chunk the four input shifts; the u types each shift feeds are the kernel's (ppst_hip.h)
                self._up9_tmp = [(chan_base + 32 * c, dy, dx, 1 if (dy, dx) == (0, 0) else 0)
                                 for c in range(nchunk) for dy, dx in ((0, 0), (-1, 0), (0, -1), (-1, -1))]
        elif kind == "dgrad":
            # input gradient of a stride-1 conv (zero padding): a conv of dY with the transposed,
            # flipped weights  Wd[c][n][ky][kx] = W[n][c][k-1-ky][k-1-kx]  -- same memory, other strides
            assert k in (1, 3)
            self.n_groups = 1
            self.halo = 0 if k == 1 else 1
            nchunk = cout // 32                      # the reduction now runs over the forward's Cout
            assert cout % 32 == 0
            for c in range(nchunk):
                first = True
                for ky in range(k):
                    for kx in range(k):
                        steps.append((32 * c, ky - k // 2, kx - k // 2, 1 if first else 0))
                        src.append((32 * c, ky, kx))
                        first = False
            wsrc = w.view(-1)[k * k - 1:]            # element (ky', kx') = (0,0) is W[..][k-1][k-1]
            sn, sc, sy, sx = k * k, cin * k * k, -k, -1
            self.cout, self.cin = cin, cout          # roles swap
            cout, cin = cin, cout
            self.bn = 128 if cout >= 128 else 64
        elif kind == "dgrad_s2d":
            # input gradient of the stride-2 3x3 conv: element i = 2q+p of the (blurred) input grid
            # receives  sum_{ky = p (mod 2)} W[.,.,ky,.]^T dY[q - ky//2]  -> 4 output phases scattered
            # with stride 2, like the transposed conv; groups are padded to 4 steps per chunk.
            assert k == 3 and cout % 32 == 0
            self.n_groups = 4
            self.halo = 1
            taps = {0: [(0, 0), (-1, 2)], 1: [(0, 1)]}
            for py in range(2):
                for px in range(2):
                    for c in range(cout // 32):
                        tl = [(dy, dx, ky, kx) for dy, ky in taps[py] for dx, kx in taps[px]]
                        for i in range(4):
                            if i < len(tl):
                                dy, dx, ky, kx = tl[i]
                                steps.append((32 * c, dy, dx, 1 if i == 0 else 0))
                                src.append((32 * c, ky, kx))
                            else:
                                steps.append((32 * c, 0, 0, 0))
                                src.append((-1, 0, 0))
            wsrc = w
            sn, sc, sy, sx = 9, cin * 9, 3, 1     # n' = c (stride 9), c' = n (stride cin*9), no flip
            self.cout, self.cin = cin, cout
            cout, cin = cin, cout
            self.bn = 128 if cout >= 128 else 64
        elif kind == "dgrad_s2ds":
            # round 5: the same input gradient with the four output phases STACKED as 4 x Cin output channels of ONE stride-1 conv
            # with 2 x 2 taps (offsets 0 / -1 per axis), followed by ops.depth_to_space: one group, every step real (the four-group
            # table pads 7 of its 16 steps per chunk), the tile staged once instead of once per phase, and 4 x Cin >= 128 output
            # channels put a thin layer (Cin 32 / 64) on the 8-wave kernels -- its four-group launches ran 2 312-5 780 FOUR-wave
            # blocks at 21-65 TFLOP/s.  Weights: ppst_dgrad_s2d_stack_weight (out (4 Cin, Cout, 2, 2) from the forward parameter).
            assert k == 3 and cout % 32 == 0
            self.n_groups = 1
            self.halo = 1
            for c in range(cout // 32):
                for i, (ty, tx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                    steps.append((32 * c, -ty, -tx, 1 if i == 0 else 0))
                    src.append((32 * c, ty, tx))
            wsrc = torch.empty((4 * cin, cout, 2, 2), device=w.device, dtype=torch.float32)
            check(lib.ppst_dgrad_s2d_stack_weight(_p(w), _p(wsrc), cout, cin, _stream()), "ppst_dgrad_s2d_stack_weight")
            sn, sc, sy, sx = cout * 4, 4, 2, 1
            self.cout, self.cin = 4 * cin, cout
            cout, cin = 4 * cin, cout
            self.bn = 128 if cout >= 128 else 64
        elif kind == "dgradT":
            # input gradient of the fused 4x4 stride-2 transposed conv (kind 'convT'): a stride-2 4x4 conv (pad 1) of
            # dY, run over the space-to-depth copy of dY:  oy = 2*iy - 1 + ky  ->  (dq, phase, ky) per axis in
            # {(-1,1,0), (0,0,1), (0,1,2), (+1,0,3)}.  ``weight`` is the FORWARD (Cout,Cin,3,3) parameter.
            assert k == 3 and cout % 32 == 0
            self.n_groups = 1
            self.halo = 1
            wsrc = torch.empty((cin, cout, 4, 4), device=w.device, dtype=torch.float32)
            check(lib.ppst_upscale_weight(_p(w), _p(wsrc), cout, cin, float(scale), _stream()), "ppst_upscale_weight")
            self.fwd_scale = float(scale)
            scale = 1.0
            taps = {0: [(0, 1), (1, 3)], 1: [(-1, 0), (0, 2)]}
            for py in range(2):
                for px in range(2):
                    for c in range(cout // 32):
                        first = True
                        for dqy, ky in taps[py]:
                            for dqx, kx in taps[px]:
                                steps.append(((py * 2 + px) * cout + 32 * c, dqy, dqx, 1 if first else 0))
                                src.append((32 * c, ky, kx))
                                first = False
            sn, sc, sy, sx = cout * 16, 16, 4, 1      # w4[c][n][ky][kx]: output channel = c, reduction = n
            self.w4_shape = (cin, cout, 4, 4)
            self.cout, self.cin = cin, cout
            cout, cin = cin, cout
            self.bn = 128 if cout >= 128 else 64
        else:
            raise ValueError(kind)
        self.src = src
        self.max_chan = max(t[0] for t in steps)      # highest first-channel of any step (input needs max_chan + 32)
        self.wstrides = (sn, sc, sy, sx)
        self.nsteps = len(steps) // self.n_groups
        self.flop_steps = sum(1 for t in src if t[0] >= 0) // self.n_groups
        # every element of the weight tensor is the target of exactly one (step, output channel, k) of the table: the weight
        # gradient's split reduction then WRITES all of dW and no zero fill has to run in front of it
        live = [tuple(t[:3]) for t in src if t[0] >= 0]
        target = math.prod(self.w4_shape) if kind == "dgradT" else w.numel()     # what conv_wgrad(plan, ...) returns
        self.full_cover = bool(self.n_groups == 1 and len(set(live)) == len(live) and len(live) * 32 * cout == target)
        dev = w.device
        # flags: bit 0 = this step opens a chunk; bit 1 = the NEXT step of the group opens one, bits 8.. = its channel
        # offset (lets the kernel request a chunk's activations a step early when every chunk spans >= 2 steps)
        ns_ = self.nsteps

        def encode(steps_):
            enc, chunk_idx = [], -1
            for i, (c_, dy_, dx_, f_) in enumerate(steps_):
                if i % ns_ == 0:
                    chunk_idx = -1                   # the chunk count restarts with every group
                chunk_idx += 1 if f_ else 0
                nxt = steps_[i + 1] if (i + 1) % ns_ != 0 else None
                # bit 2 = parity of this step's chunk index within its group (the activation-ring slot of conv_ksplit.hip)
                w_ = f_ | ((2 | (nxt[0] << 8)) if (nxt is not None and nxt[3]) else 0) | ((chunk_idx & 1) << 2)
                enc.append((c_, dy_, dx_, w_))
            return enc
        enc = encode(steps)
        starts = [i for i, t in enumerate(steps) if t[3] == 1] + [len(steps)]
        lens = [b_ - a_ for a_, b_ in zip(starts[:-1], starts[1:])]
        self.chunk_starts0 = [i for i in starts if i < ns_] + [ns_]          # chunk starts of ONE group (every group has the same)
        self.chunk_starts0_k64 = None
        self.early_a = 1 if (min(lens) >= 2 and ns_ >= 3) else 0
        self.max_chunk_steps = max(lens)
        self.min_chunk_steps = min(lens)
        # conv_ksplit.hip stores a chunk one step PAIR before its first use: a 2-step chunk must not straddle two pairs
        self.ksplit_ok = bool(self.early_a and all(l_ >= 3 or (a_ % ns_) % 2 == 0 for a_, l_ in zip(starts[:-1], lens)))
        # 4 padding rows: the kernel prefetches the descriptor of step s+3 without a bounds test
        self.steps = torch.tensor(enc + [(0, 0, 0, 0)] * 4, dtype=torch.int32, device=dev).contiguous()
        self.steps_dual = self.src_dual = None
        dual = self.__dict__.pop("_dual_tmp", None)
        if dual is not None:
            self.steps_dual = torch.tensor(encode(dual[0]) + [(0, 0, 0, 0)] * 4, dtype=torch.int32, device=dev).contiguous()
            sd_ = torch.tensor(dual[1], dtype=torch.int32, device=dev)
            self.src_dual = (sd_[:, 0].contiguous(), sd_[:, 1].contiguous(), sd_[:, 2].contiguous())
        # K64 tables: every second 32-channel chunk of a group opens a 64-channel step group with the same taps (needs an even chunk
        # count per group and plain 32-channel chunk order: conv k = 3, s2d, convT and its phase-pair form)
        self.steps_k64 = self.src_k64 = self.steps_dual_k64 = self.src_dual_k64 = None

        def k64_tables(steps_, src_, ngroups):
            per = len(steps_) // ngroups
            keep, chunk = [], -1
            for i, t in enumerate(steps_):
                if i % per == 0:
                    chunk = -1
                chunk += 1 if t[3] else 0
                if chunk % 2 == 0:
                    keep.append(i)
            st = [steps_[i] for i in keep]
            per2 = len(st) // ngroups
            enc2, ci = [], -1
            for i, (c_, dy_, dx_, f_) in enumerate(st):
                if i % per2 == 0:
                    ci = -1
                ci += 1 if f_ else 0
                nxt = st[i + 1] if (i + 1) % per2 != 0 else None
                enc2.append((c_, dy_, dx_, f_ | ((2 | (nxt[0] << 8)) if (nxt is not None and nxt[3]) else 0) | ((ci & 1) << 2)))
            if ngroups == self.n_groups:
                self.chunk_starts0_k64 = [i for i, t in enumerate(st[:per2]) if t[3]] + [per2]
            sr = torch.tensor([src_[i] for i in keep], dtype=torch.int32, device=dev)
            return (torch.tensor(enc2 + [(0, 0, 0, 0)] * 4, dtype=torch.int32, device=dev).contiguous(),
                    (sr[:, 0].contiguous(), sr[:, 1].contiguous(), sr[:, 2].contiguous()))
        if kind in ("conv", "s2d", "convT") and self.halo == 1 and cin % 64 == 0 and cout >= 128 and self.early_a:
            self.steps_k64, self.src_k64 = k64_tables(steps, src, self.n_groups)
            if dual is not None:
                self.steps_dual_k64, self.src_dual_k64 = k64_tables(dual[0], dual[1], 2)
        up9 = self.__dict__.pop("_up9_tmp", None)
        self.steps_up9 = None
        if up9 is not None:
            self.steps_up9 = torch.tensor(encode(up9) + [(0, 0, 0, 0)] * 4, dtype=torch.int32, device=dev).contiguous()
        s = torch.tensor(src, dtype=torch.int32, device=dev)
        src_c, src_ky, src_kx = s[:, 0].contiguous(), s[:, 1].contiguous(), s[:, 2].contiguous()
        self.src_dev = (src_c, src_ky, src_kx)
        cs = [i for i, t in enumerate(steps) if t[3] == 1] + [len(steps)]
        self.chunk_start = torch.tensor(cs, dtype=torch.int32, device=dev)
        # chunks per group (every group has the same count): ring depth hint for the kernel
        self.chunks_per_group = (len(cs) - 1) // self.n_groups
        self.scale = float(scale)
        self.wsrc = wsrc
        if self.precision == 2:      # exact-fp32 verification kernel reads the fp32 weights directly: nothing to pack
            self.wpack = None
            return
        self._packs = {}
        self.wpack = self.pack_for(self.bn)

    def pack_for(self, bn):
        """packed weight blob for N tile ``bn`` (built on first use: the fat-wave kernel wants 128 / 256)."""
        hit = self._packs.get(bn)
        if hit is not None:
            return hit
        cout = self.cout
        sn, sc, sy, sx = self.wstrides
        src_c, src_ky, src_kx = self.src_dev
        n_tiles = (cout + bn - 1) // bn
        npl = 8 if self.precision == 0 else 4
        wpack = torch.empty(self.n_groups * n_tiles * self.nsteps * npl * bn * 8, dtype=torch.int16, device=self.steps.device)
        check(lib.ppst_conv_pack(_p(self.wsrc), sn, sc, sy, sx, float(self.scale), cout, bn, _p(src_c), _p(src_ky), _p(src_kx),
                                 self.nsteps, self.n_groups, self.precision, _p(wpack), _stream()), "ppst_conv_pack")
        self._packs[bn] = wpack
        return wpack

    def pack_dual(self):
        """weights for ppst_conv_args.dual_b (ppst_conv_pack_dual), built on first use."""
        hit = self._packs.get("dual")
        if hit is not None:
            return hit
        sn, sc, sy, sx = self.wstrides
        c_, ky_, kx_ = self.src_dual
        n_tiles = (self.cout + 127) // 128
        npl = 8 if self.precision == 0 else 4                # hi + lo planes, or the hi planes of a single-pass mode
        wpack = torch.empty(2 * n_tiles * self.nsteps * npl * 256 * 8, dtype=torch.int16, device=self.steps.device)
        check(lib.ppst_conv_pack_dual(_p(self.wsrc), sn, sc, sy, sx, float(self.scale), self.cout, _p(c_), _p(ky_), _p(kx_),
                                      self.nsteps, 2, self.precision, _p(wpack), _stream()), "ppst_conv_pack_dual")
        self._packs["dual"] = wpack
        return wpack

    def pack_k64(self, bn, dual=False):
        """weights for ppst_conv_args.k64 (ppst_conv_pack_k64), built on first use."""
        key = "k64_dual" if dual else "k64_%d" % bn
        hit = self._packs.get(key)
        if hit is not None:
            return hit
        sn, sc, sy, sx = self.wstrides
        c_, ky_, kx_ = self.src_dual_k64 if dual else self.src_k64
        ng = 2 if dual else self.n_groups
        nst = c_.numel() // ng
        n_tiles = (self.cout + (bn // 2 if dual else bn) - 1) // (bn // 2 if dual else bn)
        wpack = torch.empty(ng * n_tiles * nst * 8 * bn * 8, dtype=torch.int16, device=self.steps.device)
        check(lib.ppst_conv_pack_k64(_p(self.wsrc), sn, sc, sy, sx, float(self.scale), self.cout, bn, _p(c_), _p(ky_), _p(kx_), nst, ng,
                                     self.precision, 1 if dual else 0, _p(wpack), _stream()), "ppst_conv_pack_k64")
        self._packs[key] = wpack
        return wpack

    def pack_up9(self):
        """weights for variant 11 (ppst_conv_pack_up9: the un-blurred 3x3 kernel), built on first use."""
        hit = self._packs.get("up9")
        if hit is not None:
            return hit
        w = self.wparam                                          # (Cout, Cin, 3, 3) fp32, the layer's own parameter
        nbytes = lib.ppst_conv_pack_up9_bytes(self.cout, self.cin)
        wpack = torch.empty(nbytes // 2, dtype=torch.int16, device=self.steps.device)
        check(lib.ppst_conv_pack_up9(_p(w), self.cin * 9, 9, 3, 1, float(self.up_scale), self.cout, self.cin, _p(wpack), _stream()),
              "ppst_conv_pack_up9")
        self._packs["up9"] = wpack
        return wpack

    def pack_wino(self):
        """transformed weights for variant 10 (ppst_conv_pack_wino), built on first use."""
        hit = self._packs.get("wino")
        if hit is not None:
            return hit
        sn, sc, sy, sx = self.wstrides
        nbytes = lib.ppst_conv_pack_wino_bytes(self.cout, self.cin)
        wpack = torch.empty(nbytes // 2, dtype=torch.int16, device=self.steps.device)
        check(lib.ppst_conv_pack_wino(_p(self.wsrc), sn, sc, sy, sx, float(self.scale), self.cout, self.cin, _p(wpack), _stream()),
              "ppst_conv_pack_wino")
        self._packs["wino"] = wpack
        return wpack

    def takes_in_res(self, H, W):
        """True when a call of this plan on an (H, W) input may carry ``in_res`` (the 1x1 streaming kernel in the fp32-class mode)."""
        return (self.precision == 0 and self.kind == "conv" and self.k == 1 and
                self.choose_kernel(H, W, H, W, H, W, 1)[0] == 4)

    def wino_ok(self, th, tw, oh, ow, H, W, osy):
        # (a call with normalise-on-load and more than 1024 input channels is kept off this kernel in __call__: its LDS copy of
        # the (a, s) table is sized for 32 chunks)
        return (self.precision == 0 and self.kind in ("conv", "dgrad") and self.k == 3 and self.cout >= 128 and osy == 1
                and (th, tw) == (oh, ow) == (H, W)
                and ((th + 15) // 16) * ((tw + 15) // 16) * ((self.cout + 127) // 128) >= WINO["min_blocks"])

    def choose_kernel(self, th, tw, oh, ow, H, W, osy, B=None, allow_up9=True):
        """(variant, N tile, tile rows) of ppst_conv_args for one launch of this plan -- a function of the plan and of ONE
        image's geometry only.  The batch size is deliberately not an argument: every variant gives bit-identical outputs,
        but their tile statistics differ in the last bit (other summation tree), and a shard of a batch has to reproduce the
        whole batch bit for bit (evaluation.grid_exchange; "2 simulated ranks == 1 rank"; tests/test_abi_cpu.py).
          0  tile kernel (64 px x 64 ch waves), bn 64 / 128;      2  its N-256 form (128 px x 64 ch waves) for Cout % 256 == 0
          4  1x1 streaming kernel;   5  direct form for thin stride-2 tables;   6  3x3 register-reuse form (thin layers)
          1 / 3 / 7 and variant 6 with bn 128: measured experiments, off unless a test switches them on."""
        rows = TILE_ROWS["value"]
        variant, bn = 0, self.bn
        single = self.precision in (1, 3)
        if self.precision not in (0, 1, 3):
            return variant, bn, rows                     # fp16x2 experiment / exact-fp32 verification: the tile kernel only
        if WINO["value"] and self.wino_ok(th, tw, oh, ow, H, W, osy):
            wblocks = ((th + 15) // 16) * ((tw + 15) // 16) * ((self.cout + 127) // 128) * (B or 1)
            aware_b = BATCH_AWARE["value"] and B is not None
            if not (aware_b and (wblocks < WINO["fill"] or (KSPLIT["value"] and 0 in KSPLIT["variants"] and self.cout <= 256
                                                             and wblocks <= WINO["ksplit_fill"]))):
                return 10, 128, 16
        tiles16 = ((th + 15) // 16) * ((tw + 15) // 16) * self.n_groups          # blocks PER IMAGE per N tile
        cv = CONV_VARIANT["value"]
        if cv in (1, 3) or TWO_BLOCK_128["value"] or KSPLIT_128["value"] or TILE24_128["value"] or TALL_TILE_128["value"]:
            _need_experiments("the requested conv variant")
        if (UP9["value"] and allow_up9 and self.kind == "convT" and self.precision == 0 and getattr(self, "steps_up9", None) is not None
                and self.early_a and ((th + 14) // 15) * ((tw + 14) // 15) * (self.cout // 64) >= UP9["min_blocks"]
                and th * tw >= UP9["min_fill"] * (((th + 14) // 15) * 15) * (((tw + 14) // 15) * 15)):
            return "up9", 256, 15
        if (DUAL_CONVT["value"] and self.kind == "convT" and getattr(self, "steps_dual", None) is not None
                and self.cout % 256 != 0 and CONV_VARIANT["value"] == 2 and self.early_a
                and ((th + 15) // 16) * ((tw + 15) // 16) * 2 * (self.cout // 128) >= DUAL_CONVT["min_blocks"]):
            return "dual", 256, 16
        n256_ok = self.cout % 256 == 0 and tiles16 * (self.cout // 256) >= FAT_MIN_BLOCKS
        aware = BATCH_AWARE["value"] and B is not None and not single and self.bn == 128 and self.early_a and self.halo == 1
        if aware and n256_ok and tiles16 * (self.cout // 256) * B < BATCH_AWARE["fill"]:
            n256_ok = False                              # under-filled: the tile kernel has twice the N tiles
        # (with the across-block K split on, an under-filled launch keeps its 16-row tiles and splits K instead: 256 -> 256 @64^2 x 2
        #  38 us with S = 4 against 69 us on 8-row tiles with S = 2)
        small = (aware and tiles16 * ((self.cout + 127) // 128) * B < BATCH_AWARE["fill"]
                 and not (KSPLIT["value"] and 0 in KSPLIT["variants"]))
        if cv == 1 and not single and self.early_a and self.cout >= 128:
            variant, bn = 1, (256 if n256_ok else 128)
        elif cv in (2, 3) and self.early_a:
            force3 = cv == 3                             # (tests) every eligible plan on the two-block kernel
            if n256_ok and not force3:
                variant, bn = 2, 256                     # 8 waves x (128 px x 64 ch), N tile 256 (conv_mfma2.hip, WNW = 4)
            elif single:
                # single-pass modes: the N-256 geometry, and for Cout = 128-class layers with a halo the 32 x 16 px x 128 ch tile
                if (self.bn == 128 and self.halo == 1 and self.n_groups == 1 and TALL_TILE_SINGLE["value"] and
                        ((th + 31) // 32) * ((tw + 15) // 16) * ((self.cout + 127) // 128) >= TALL_TILE_SINGLE["min_blocks"]):
                    variant, rows = 7, 32
            elif self.bn == 128 and (force3 or (TWO_BLOCK_128["value"] and
                                                tiles16 * ((self.cout + 127) // 128) >= 2 * FAT_MIN_BLOCKS)):
                variant = 3                              # two 4-wave blocks per CU, N tile 128, one activation slot
            elif (self.bn == 128 and self.halo == 1 and (small or (TWO_BLOCK_8ROW["value"] and
                  ((th + 7) // 8) * ((tw + 15) // 16) * self.n_groups * ((self.cout + 127) // 128) >= TWO_BLOCK_8ROW["min_blocks"]))):
                rows = 8                                 # tile kernel, two 4-wave blocks per CU
            elif (self.bn == 128 and KSPLIT_128["value"] and getattr(self, "ksplit_ok", False)
                  and tiles16 * ((self.cout + 127) // 128) >= KSPLIT_128["min_blocks"]):
                variant = 8                              # two K-groups of 128 px x 64 ch waves, N tile 128 (conv_ksplit.hip)
            elif (self.bn == 128 and TILE24_128["value"] and ((th + 23) // 24) * 24 <= th * (1.0 + TILE24_128["max_waste"]) and
                  ((th + 23) // 24) * ((tw + 15) // 16) * self.n_groups * ((self.cout + 127) // 128) >= TILE24_128["min_blocks"]):
                variant, rows = 9, 24                    # block tile 24 x 16 px x 128 ch, two activation slots (WMW = 4, MT_ = 6)
            elif (self.bn == 128 and TALL_TILE_128["value"] and
                  ((th + 31) // 32) * ((tw + 15) // 16) * self.n_groups * ((self.cout + 127) // 128) >= TALL_TILE_128["min_blocks"]):
                variant, rows = 7, 32                    # block tile 32 x 16 px x 128 ch, one activation slot (WMW = 4)
        same = (th, tw) == (oh, ow)
        if STREAM_1X1["value"] and self.halo == 0 and self.n_groups == 1 and osy == 1 and same and (oh, ow) == (H, W):
            variant, bn, rows = 4, 64, TILE_ROWS["value"]            # 1x1 convs: streaming kernel, no activation staging
        elif (self.halo == 1 and self.n_groups == 1 and osy == 1 and same and self.cout <= DIRECT_MAX["cout"]
              and self.nsteps <= DIRECT_MAX["nsteps"]):
            # thin layers: direct form of the streaming kernel; plain 3x3 stride-1 tables (order (chunk, dy, dx)) on its
            # register-reuse form
            variant, bn, rows = (6 if (self.kind in ("conv", "dgrad") and self.k == 3) else 5), 64, TILE_ROWS["value"]
        elif (not single and self.kind in ("conv", "dgrad") and self.k == 3 and 64 < self.cout <= DIRECT_MAX["cout3x3"] and same):
            variant, bn, rows = 6, 128, TILE_ROWS["value"]           # (experiment) 32 px x 128 ch waves for Cout in 65..128
        return variant, bn, rows

    def _ksplit(self, variant, a, skip):
        """ppst_conv_args.ksplit of the launch described by ``a`` (0: none) -- ops.KSPLIT."""
        if (skip or variant not in KSPLIT["variants"] or a.in_presplit
                or not (a.tile_rows == 16 or (variant == 0 and a.tile_rows == 8))):
            return 0, None
        cs = self.chunk_starts0_k64 if a.k64 else self.chunk_starts0       # chunk starts of one group of the launch's table
        if cs is None or cs[-1] != a.nsteps:
            return 0, None
        if a.dual_b or (variant == 2 and ((self.precision == 0) != (a.io_st == 0) or (a.k64 and not a.halo))):
            return 0, None              # (the N-256 kernel's K-split instances: fp32-class on fp32 tensors, single-pass on half-stored ones)
        n_tiles = -(-self.cout // a.bn)
        blocks = a.B * lib.ppst_conv_tiles(a.tile_h, a.tile_w, a.tile_rows) * a.n_groups * n_tiles
        # (the N-256 and Winograd kernels hold 128 accumulator registers per thread: S <= 4, include/ppst_hip.h)
        # -- and what the hand-over of 256 KB per block costs them (tests/conv_ksplit_time.py): the Winograd kernel gains from S = 2 with
        # >= 4 chunks left per block only, the N-256 kernel needs >= 24 steps left
        max_s, min_steps = {0: (8, KSPLIT["min_steps"]), 2: (4, 24), 10: (2, 36)}[variant]
        return _ksplit_choice(blocks, cs, KSPLIT["max_blocks"], min_steps, max_s)

    def __call__(self, x, bias=None, noise=None, noise_weight=0.0, act=ACT_NONE, prelu=None, stats=False,
                 residual=None, out=None, out_scale=1.0, pad_mode=PAD_ZERO, out_hw=None, res_after_act=False,
                 in_ss=None, in_act=ACT_NONE, in_prelu=None, presplit=False, in_res=None):
        in_ld = _nhwc_ld(x, "conv input", half_ok=True)
        B, H, W, xc = x.shape
        if x.dtype != torch.float32 and {torch.float16: 3, torch.bfloat16: 1}[x.dtype] != self.precision:
            raise RuntimeError("conv input stored as %s needs a plan of the matching single-pass precision mode (plan: mode %d)"
                               % (x.dtype, self.precision))
        # the step table lives on the device: the C ABI cannot check these, and the kernel indexes
        # x[.. + chan + 32), in_ss[b][chan][2] and noise[(b*oh+oy)*ow+ox] blindly
        need_c = self.max_chan + 32
        if xc < need_c:
            raise RuntimeError("conv input has %d channels (pixel stride %d); the %s plan reads %d" % (xc, in_ld, self.kind, need_c))
        if in_ss is not None and (in_ss.dim() != 3 or in_ss.shape[0] != B or in_ss.shape[1] < need_c or in_ss.shape[2] != 2
                                  or not in_ss.is_contiguous()):
            raise RuntimeError("in_ss must be a contiguous (B=%d, >=%d, 2) table, got %s" % (B, need_c, tuple(in_ss.shape)))
        if self.kind == "convT":
            th, tw, oh, ow, osy = H, W, 2 * H, 2 * W, 2
        elif self.kind == "dgrad_s2d":
            oh, ow = out_hw                      # extent of the (blurred) tensor the forward conv read
            th, tw, osy = (oh + 1) // 2, (ow + 1) // 2, 2
        elif self.kind in ("s2d", "dgrad_s2ds"):     # (dgrad_s2ds: out_hw = the stacked extents, ceil(input-grid extent / 2))
            oh, ow = out_hw
            th, tw, osy = oh, ow, 1
        else:
            th, tw, oh, ow, osy = H, W, H, W, 1
        if out is None:
            out = torch.empty((B, oh, ow, self.cout), device=x.device, dtype=x.dtype)
        out_ld = _nhwc_ld(out, "conv output", half_ok=True)
        if out.dtype != x.dtype or (residual is not None and residual.dtype != x.dtype):
            raise RuntimeError("conv input, residual and output share one storage type (ppst_conv_args.io_st): got %s / %s / %s"
                               % (x.dtype, None if residual is None else residual.dtype, out.dtype))
        if tuple(out.shape) != (B, oh, ow, self.cout):
            raise RuntimeError("conv output must be %s, got %s" % ((B, oh, ow, self.cout), tuple(out.shape)))
        if noise is not None and (noise.numel() != B * oh * ow or not noise.is_contiguous()):
            raise RuntimeError("noise must be a contiguous (B,1,%d,%d) tensor with B = %d rows, got %s" % (oh, ow, B, tuple(noise.shape)))
        if residual is not None and tuple(residual.shape) != (B, oh, ow, self.cout):
            raise RuntimeError("residual must match the output shape %s, got %s" % ((B, oh, ow, self.cout), tuple(residual.shape)))
        for t, n in ((bias, "bias"), (noise, "noise"), (prelu, "prelu")):
            _chk(t, n)
        # the nine-product upscale takes no normalise-on-load, residual, PReLU or non-zero padding: a call that carries one of them
        # falls back to the phase-pair / four-phase forms (which take all of them) instead of failing (round-4 ADVICE)
        up9_ok = in_ss is None and residual is None and act != ACT_PRELU and pad_mode == PAD_ZERO
        variant, bn, rows = self.choose_kernel(th, tw, oh, ow, H, W, osy, B, allow_up9=up9_ok)
        # single-pass modes on half-stored activations: 64 input channels per step where the N-256 kernel family runs the layer
        if variant == 10 and in_ss is not None and self.cin > 1024:
            WINO["value"], prev = False, WINO["value"]
            try:
                variant, bn, rows = self.choose_kernel(th, tw, oh, ow, H, W, osy, B, allow_up9=up9_ok)
            finally:
                WINO["value"] = prev
        k64 = (K64["value"] and x.dtype != torch.float32 and self.precision in (1, 3) and variant in (2, 7, "dual")
               and (self.steps_dual_k64 if variant == "dual" else self.steps_k64) is not None)
        if k64 and variant == 7:
            variant, rows = 9, 24                # the Cout = 128-class layers: 24 x 16 px x 128 ch tiles (two activation slots fit)
        st = None
        if stats:
            tiles = lib.ppst_conv_tiles(th, tw, rows)
            st = torch.empty((B, (1 if variant == "up9" else self.n_groups) * tiles, self.cout, 2), device=x.device, dtype=torch.float32)
            # (a block of variant 11 writes its row only for channels it owns and every block of the grid writes: no zero fill needed)
        a = _lib.ConvArgs()
        dual, up9 = variant == "dual", variant == "up9"
        if dual:
            variant = 2
        if up9:
            variant = 11
        if k64:
            wp = self.pack_k64(bn, dual)
            steps_t = self.steps_dual_k64 if dual else self.steps_k64
        else:
            wp = None if self.precision == 2 else (self.pack_wino() if variant == 10 else self.pack_up9() if up9 else self.pack_dual() if dual
                                                   else self.pack_for(bn))
            steps_t = self.steps_up9 if up9 else self.steps_dual if dual else self.steps
        a.x, a.wpack, a.steps, a.y = _p(x), _p(wp), _p(steps_t), _p(out)
        a.k64 = 1 if k64 else 0
        a.variant = variant
        a.dual_b = 1 if dual else 0
        a.bias, a.noise, a.prelu, a.stats = _p(bias), _p(noise), _p(prelu), _p(st)
        a.residual = _p(residual)
        a.res_ld = _nhwc_ld(residual, "residual", half_ok=True) if residual is not None else 0
        a.io_st = _ST[x.dtype]
        a.noise_weight, a.out_scale = float(noise_weight), float(out_scale)
        a.B, a.in_h, a.in_w, a.in_ld = B, H, W, in_ld
        a.out_h, a.out_w, a.out_ld, a.cout = oh, ow, out_ld, self.cout
        a.nsteps, a.n_groups, a.pad_mode = (self.nsteps // 2 if k64 else self.nsteps), (1 if up9 else 2 if dual else self.n_groups), pad_mode
        a.in_off_y = a.in_off_x = 0
        a.out_sy = a.out_sx = osy
        a.act, a.precision = act | (0x100 if res_after_act else 0), self.precision
        a.tile_h, a.tile_w, a.halo, a.bn, a.tile_rows = th, tw, self.halo, bn, rows
        _chk(in_ss, "in_ss"); _chk(in_prelu, "in_prelu")
        a.in_scale_shift, a.in_prelu, a.in_act = _p(in_ss), _p(in_prelu), in_act
        a.in_c = in_ss.shape[1] if in_ss is not None else 0
        if in_res is not None:       # the producer's resnet merge applied on load (ppst_conv_args.in_res): the 1x1 streaming kernel only
            if variant != 4 or in_ss is None or self.precision != 0 or x.dtype != torch.float32:
                raise RuntimeError("in_res needs a 1x1 plan on the streaming kernel (ops.STREAM_1X1), in_ss, precision 0 and fp32 storage")
            if tuple(in_res.shape) != tuple(x.shape):
                raise RuntimeError("in_res must have the conv input's shape %s, got %s" % (tuple(x.shape), tuple(in_res.shape)))
            a.in_res, a.in_res_ld = _p(in_res), _nhwc_ld(in_res, "in_res")
        a.flop_steps = self.flop_steps
        a.a_slots = min(3, self.chunks_per_group)
        a.early_a = self.early_a
        if presplit:         # experiment: x holds the pre-split layout (ops.presplit); the tile kernel stages it by LDS-DMA
            _need_experiments("pre-split conv input")
            if variant != 0 or bn != 128 or in_ss is not None or self.nsteps // max(1, self.chunks_per_group) < 4:
                raise RuntimeError("pre-split input: tile kernel (bn 128) plans with chunks of >= 4 steps and no in_ss only")
            a.in_presplit = 1
        # (batch-aware passes only: the split depends on the batch in the launch and changes the summation order -- a shard of an
        #  inference batch has to reproduce the whole batch bit for bit)
        ks_S, ks_cuts = self._ksplit(variant, a, self.precision == 2) if (KSPLIT["value"] and BATCH_AWARE["value"]) else (0, None)
        a.ksplit = ks_S
        ks_arr = (ctypes.c_int32 * (ks_S + 1))(*ks_cuts) if ks_S else None         # (alive until the call returns)
        a.ksplit_starts = ctypes.cast(ks_arr, ctypes.c_void_p) if ks_S else None
        if self.precision == 2:
            sn, sc, sy, sx = self.wstrides
            c_, ky_, kx_ = self.src_dev
            check(lib.ppst_conv2d_f32(ctypes.byref(a), _p(self.wsrc), sn, sc, sy, sx, self.scale, _p(c_), _p(ky_), _p(kx_), _stream()),
                  "ppst_conv2d_f32")
        else:
            check(lib.ppst_conv2d_mfma(ctypes.byref(a), _stream()), "ppst_conv2d_mfma")
        if stats:
            return out, st
        return out


def ksplit_check():
    """ppst_conv_ksplit_check on the current stream: True if every flag wait of the K-split launches so far ended on its flag (or none
    ran), False if a block gave up (that launch's output is wrong).  Synchronises the stream."""
    return lib.ppst_conv_ksplit_check(_stream()) <= 0


def _ksplit_choice(blocks, chunk_starts, max_blocks, min_steps, max_s=8):
    """(S, starts) of ppst_conv_args.ksplit / ksplit_starts for a launch of ``blocks`` blocks over a table whose chunks open at
    ``chunk_starts`` (ascending, from 0; the last entry = nsteps): the largest S of 8 / 4 / 2 that keeps S x blocks <= max_blocks with
    every share -- cut at the chunk starts nearest to equal shares -- at least ``min_steps`` long.  (0, None): no split."""
    nsteps = chunk_starts[-1]
    for S in (8, 4, 2):
        if S > max_s or blocks * S > max_blocks or (S - 1) * blocks > 256 or len(chunk_starts) - 1 < S:
            continue
        cuts = [0]
        for i in range(1, S):
            t = i * nsteps / S
            cuts.append(min(chunk_starts, key=lambda c: (abs(c - t), c)))
        cuts.append(nsteps)
        if all(b_ - a_ >= min_steps for a_, b_ in zip(cuts[:-1], cuts[1:])):
            return S, cuts
    return 0, None


def repack_plans(plans):
    """Refresh the packed weights of ``plans`` (ConvPlan objects whose fp32 weights changed in place -- an Adam step) with TWO
    launches: the fused-upscale 4x4 kernels of the 'convT' / 'dgradT' plans (ppst_upscale_weight_batch), then every pack of every
    plan (ppst_conv_pack_batch).  Returns the job tables; pass them back as ``tables`` while the set of plans / packs is unchanged
    (they hold device pointers only: nothing is rebuilt on the host)."""
    import numpy as np
    up, pk, wino = [], [], []
    for pl in plans:
        if pl.precision == 2:
            continue
        if pl.kind == "dgrad_s2ds":              # the phase-stacked 2 x 2 kernel: refreshed from the parameter before the packs read it
            wino.append(("s2ds", pl.wparam, pl.wsrc, pl.wparam.shape[0], pl.wparam.shape[1], None))
        if pl.kind in ("convT", "dgradT"):
            cin4, cout4 = pl.wsrc.shape[0], pl.wsrc.shape[1]         # wsrc is the (Cin, Cout, 4, 4) kernel of the FORWARD conv
            up.append((pl.wparam.data_ptr(), pl.wsrc.data_ptr(), cin4 * cout4 * 16, float(pl.up_scale), cout4, cin4))
        sn, sc, sy, sx = pl.wstrides
        c_, ky_, kx_ = pl.src_dev
        for bn, wpack in pl._packs.items():
            if bn == "wino":                     # variant-10 pack: its own transform kernel, one launch per plan
                wino.append((pl.wsrc, pl.wstrides, float(pl.scale), pl.cout, pl.cin, wpack))
                continue
            if isinstance(bn, str) and bn.startswith("k64_"):     # 64-channel-step packs: jobs of the batched pack kernel (x3 = 2)
                kd = bn == "k64_dual"
                kc, kky, kkx = pl.src_dual_k64 if kd else pl.src_k64
                kbn = 256 if kd else int(bn[4:])
                kng = 2 if kd else pl.n_groups
                knt = (pl.cout + (kbn // 2 if kd else kbn) - 1) // (kbn // 2 if kd else kbn)
                pk.append((pl.wsrc.data_ptr(), sn, sc, sy, sx, kc.data_ptr(), kky.data_ptr(), kkx.data_ptr(), wpack.data_ptr(),
                           kng * knt * (kc.numel() // kng) * 4 * kbn, float(pl.scale), pl.cout, kbn, kc.numel() // kng, kng,
                           2, 1 if pl.precision == 3 else 0, 1 if kd else 0))
                continue
            if bn == "up9":                      # variant-11 pack: from the 3x3 parameter itself, one launch per plan
                wino.append(("up9", pl.wparam, float(pl.up_scale), pl.cout, pl.cin, wpack))
                continue
            if bn == "dual":                     # two-phase-pair pack of the fused upscale: a job of the batched pack kernel
                dc, dky, dkx = pl.src_dual
                pk.append((pl.wsrc.data_ptr(), sn, sc, sy, sx, dc.data_ptr(), dky.data_ptr(), dkx.data_ptr(), wpack.data_ptr(),
                           2 * ((pl.cout + 127) // 128) * pl.nsteps * 4 * 256, float(pl.scale), pl.cout, 256, pl.nsteps, 2,
                           1 if pl.precision == 0 else 0, 1 if pl.precision == 3 else 0, 1))
                continue
            n_tiles = (pl.cout + bn - 1) // bn
            pk.append((pl.wsrc.data_ptr(), sn, sc, sy, sx, c_.data_ptr(), ky_.data_ptr(), kx_.data_ptr(), wpack.data_ptr(),
                       pl.n_groups * n_tiles * pl.nsteps * 4 * bn, float(pl.scale), pl.cout, bn, pl.nsteps, pl.n_groups,
                       1 if pl.precision == 0 else 0, 1 if pl.precision in (3, 4) else 0, 0))
    dev = plans[0].steps.device if plans else None

    def table(cls, rows, fill):
        if not rows:
            return None, 0
        arr = (cls * len(rows))()
        b0 = 0
        for j, r in zip(arr, rows):
            nb = lib.ppst_pack_job_blocks(fill(j, r))
            j.block0, j.nblocks = b0, nb
            b0 += nb
        buf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        return buf, b0

    def fill_up(j, r):
        j.w, j.out, j.total, j.scale, j.cout, j.cin = r
        return j.total

    def fill_pk(j, r):
        (j.w, j.sn, j.sc, j.sy, j.sx, j.src_c, j.src_ky, j.src_kx, j.out, j.total, j.scale, j.cout, j.bn, j.nsteps, j.n_groups,
         j.x3, j.f16, j.dual) = r
        return j.total
    tu, nbu = table(_lib.UpscaleJob, up, fill_up)
    tp, nbp = table(_lib.PackJob, pk, fill_pk)
    tables = (tu, len(up), nbu, tp, len(pk), nbp, wino)
    run_repack(tables)
    return tables


def run_repack(tables):
    tu, nu, nbu, tp, npk, nbp, wino = tables
    if nu:
        check(lib.ppst_upscale_weight_batch(_p(tu), nu, nbu, _stream()), "ppst_upscale_weight_batch")
    for e in wino:
        if isinstance(e[0], str) and e[0] == "s2ds":
            check(lib.ppst_dgrad_s2d_stack_weight(_p(e[1]), _p(e[2]), e[3], e[4], _stream()), "ppst_dgrad_s2d_stack_weight")
    if npk:
        check(lib.ppst_conv_pack_batch(_p(tp), npk, nbp, _stream()), "ppst_conv_pack_batch")
    for wsrc, strides, scale, cout, cin, wpack in wino:
        if isinstance(wsrc, str) and wsrc == "s2ds":
            continue
        if isinstance(wsrc, str):                # ("up9", the 3x3 parameter, ...): variant 11
            check(lib.ppst_conv_pack_up9(_p(strides), cin * 9, 9, 3, 1, scale, cout, cin, _p(wpack), _stream()), "ppst_conv_pack_up9")
            continue
        sn, sc, sy, sx = strides
        check(lib.ppst_conv_pack_wino(_p(wsrc), sn, sc, sy, sx, scale, cout, cin, _p(wpack), _stream()), "ppst_conv_pack_wino")


def _grad_out(out, shape, like):
    """destination of a parameter gradient: ``out`` (a contiguous view of the trainer's flat gradient, written or added into
    by the kernel itself) or a fresh tensor."""
    if out is None:
        return torch.empty(shape, device=like.device, dtype=torch.float32)
    n = 1
    for d in shape:
        n *= d
    if out.numel() != n or not out.is_contiguous() or out.dtype != torch.float32 or not out.is_cuda:
        raise RuntimeError("gradient destination must be a contiguous float32 CUDA tensor of %d elements, got %s" % (n, tuple(out.shape)))
    return out


# the LDS-DMA / transposed-read weight-gradient kernel (round 3): one 8-wave block per CU -> ONE round of <= 256 blocks (every
# further split is 2 x 147 KB of partial sums written and re-read by the scatter), >= min_tiles pixel tiles per block
# form 2: two 256-thread blocks per CU; bf16_single_pass: in precision mode 1 (bf16 compute, fp32 master weights -- BASELINE
# configs[3]) the weight gradient multiplies the hi halves only (one MFMA pass instead of three)
WGRAD_TR = {"value": True, "blocks": 256, "min_tiles": 8, "form": 2, "blocks2": 512, "pair": True, "bf16_single_pass": True, "exact": True, "nohalo": True}


def presplit(x):
    """fp32 NHWC -> the pre-split layout (experiment, ppst_conv_args.in_presplit): same shape / dtype container, per pixel and
    8-channel group 32 bytes [hi x 8 | lo x 8] bf16."""
    ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    y = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    check(lib.ppst_presplit(_p(x), _p(y), B * H * W, C, ld, C, _stream()), "ppst_presplit")
    return y


def conv_wgrad(plan, x, dy, splits=None, out=None, accumulate=False, bias_out=None, bias_accumulate=False, want_bias=False, dy_scale=1.0):
    """Weight gradient of the conv described by forward ``plan`` ('conv' or 's2d'):
    x = the tensor the forward conv read (NHWC, or the space-to-depth tensor for 's2d'),
    dy = gradient w.r.t. the conv output (before bias/activation).  Returns dW shaped
    (Cout, Cin, k, k), already multiplied by the plan's weight scale (EqualConv2d).
    ``out`` / ``accumulate``: write (or add) into a given destination -- the reduction over the pixel splits adds straight
    into the flat gradient buffer, so no separate accumulation pass (and no zero fill) runs.
    ``want_bias`` / ``bias_out``: also the column sums of dy (the bias gradient), from the fp32 values the weight-gradient kernel
    stages anyway (no second pass over dy): returns (dW, db).
    ``dy_scale``: the gradient is that of ``dy_scale * dy`` (a constant upstream factor rides on the split reduction instead of a
    pass that scales dy; not with the bias outputs)."""
    assert plan.kind in ("conv", "s2d", "dgradT")
    assert dy_scale == 1.0 or not (want_bias or bias_out is not None)
    in_ld = _nhwc_ld(x, "x", half_ok=True)
    dy_ld = _nhwc_ld(dy, "dy", half_ok=True)
    half = x.dtype != torch.float32 or dy.dtype != torch.float32
    if half and not (x.dtype == dy.dtype == torch.bfloat16 and PRECISION["value"] == 1 and WGRAD_X3["value"] and WGRAD_TR["value"]
                     and WGRAD_TR["form"] == 2 and WGRAD_TR["bf16_single_pass"]):
        raise RuntimeError("conv_wgrad on half-stored operands: both bfloat16, precision mode 1, the single-pass transposed-read kernel "
                           "(got %s / %s, mode %d)" % (x.dtype, dy.dtype, PRECISION["value"]))
    B, H, W, _ = x.shape
    _, oh, ow, cout = dy.shape
    assert cout == plan.cout
    nchunks = plan.chunk_start.numel() - 1
    want_bias = want_bias or bias_out is not None
    aligned = cout % 4 == 0 and dy_ld % 4 == 0 and in_ld % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
    if half and not (aligned and cout % 8 == 0 and dy_ld % 8 == 0 and in_ld % 8 == 0):
        raise RuntimeError("conv_wgrad on bfloat16 operands needs channel counts / strides divisible by 8 and 16-byte aligned tensors")
    x3 = WGRAD_X3["value"] and plan.precision != 2 and aligned
    use_tr = x3 and WGRAD_TR["value"]
    per = nchunks * ((cout + 127) // 128)
    tiles_total = B * ((oh + 1) // 2) * ((ow + 31) // 32)
    csum = None
    if use_tr:
        # one 8-wave block per CU (256 slots): ~2 rounds of blocks, >= min_tiles pixel tiles (2 x 32 px) each; a block writes two
        # partial slots (one per tile row)
        form2 = WGRAD_TR["form"] == 2
        if form2 and WGRAD_TR["pair"] and plan.max_chunk_steps <= 4 and nchunks % 2 == 0:
            per = (nchunks // 2) * ((cout + 127) // 128)       # two chunks per block
            if (WGRAD_TR["nohalo"] and WGRAD_TR["exact"] and plan.kind == "conv" and plan.k == 1 and plan.halo == 0 and nchunks % 4 == 0
                    and (H, W) == (oh, ow)):
                per = (nchunks // 4) * ((cout + 127) // 128)   # 1x1 tables: four chunks per block (no halo image)
        if form2:       # one partial slot per block, two blocks per CU: one round of <= 512 blocks
            gz = splits if splits is not None else max(1, min(max(1, WGRAD_TR["blocks2"] // per), max(1, tiles_total // WGRAD_TR["min_tiles"]), 2048))
            splits = gz
        else:
            gz = splits // 2 if splits is not None else max(1, min(max(1, WGRAD_TR["blocks"] // per), max(1, tiles_total // WGRAD_TR["min_tiles"]), 1024))
            splits = 2 * gz
        if want_bias:
            csum = torch.empty((gz, cout), device=x.device, dtype=torch.float32)
    elif splits is None:
        # blocks = n-tiles x chunks x splits; the LDS-staged kernel runs 3 blocks per CU (768 slots): aim for ~2 rounds, keep
        # >= 2 pixel tiles (2 x 32 px) per block.  (Round 1 capped splits at 64: the 512x512 layers with few channels ran on
        # 64-256 blocks, a third of the chip or less.)
        splits = max(1, min((WGRAD_SPLIT["blocks"] + per - 1) // per, max(1, tiles_total // WGRAD_SPLIT["min_tiles"]), 2048))
    partial = torch.empty((splits, plan.nsteps, cout, 32), device=x.device, dtype=torch.float32)
    if PROF_ON["value"]:
        lib.ppst_wgrad_flop_steps(int(plan.flop_steps))
    if use_tr:
        if WGRAD_TR["form"] == 2:
            check(lib.ppst_conv_wgrad_tr2_st(_p(x), _p(dy), _p(plan.steps), _p(plan.chunk_start), _p(partial), _p(csum), B, H, W, in_ld, oh, ow,
                                             dy_ld, cout, plan.nsteps, nchunks, splits, plan.max_chunk_steps if WGRAD_TR["pair"] else 0,
                                             plan.min_chunk_steps if WGRAD_TR["exact"] else 0,
                                             0 if (WGRAD_TR["nohalo"] and plan.kind == "conv" and plan.k == 1 and plan.halo == 0) else 1,
                                             1 if (PRECISION["value"] == 1 and WGRAD_TR["bf16_single_pass"]) else 3,
                                             _ST[x.dtype], _stream()), "ppst_conv_wgrad_tr2")
        else:
            check(lib.ppst_conv_wgrad_tr(_p(x), _p(dy), _p(plan.steps), _p(plan.chunk_start), _p(partial), _p(csum), B, H, W, in_ld, oh, ow,
                                         dy_ld, cout, plan.nsteps, nchunks, splits, _stream()), "ppst_conv_wgrad_tr")
    else:
        # bf16x3 with register staging (round 2), or precision 2 (verification): the exact fp32 MFMA
        fn, name = ((lib.ppst_conv_wgrad_bf16x3, "ppst_conv_wgrad_bf16x3") if x3 else (lib.ppst_conv_wgrad_f32, "ppst_conv_wgrad_f32"))
        check(fn(_p(x), _p(dy), _p(plan.steps), _p(plan.chunk_start), _p(partial), B, H, W, in_ld, oh, ow, dy_ld,
                 cout, plan.nsteps, nchunks, splits, _stream()), name)
    # 'dgradT': the plan's "weights" are the blurred 4x4 kernel (Cin,Cout,4,4) of the transposed conv
    shape = plan.w4_shape if plan.kind == "dgradT" else (plan.cout, plan.cin, plan.k, plan.k)
    cover = getattr(plan, "full_cover", False)
    if out is None:
        dw = (torch.empty if cover else torch.zeros)(shape, device=x.device, dtype=torch.float32)
    else:
        dw = _grad_out(out, shape, x)
        if not accumulate and not cover:
            dw.zero_()
    sn, sc, sy, sx = plan.wstrides
    c_, ky_, kx_ = plan.src_dev
    db = None
    if csum is not None:         # the split reduction also sums the bias partials: no launch of its own
        db = _grad_out(bias_out, (cout,), x)
    check(lib.ppst_wgrad_scatter(_p(partial), _p(c_), _p(ky_), _p(kx_), _p(dw), sn, sc, sy, sx, cout, plan.nsteps, splits,
                                 plan.scale * float(dy_scale), 1 if (accumulate and out is not None) else 0, _p(csum), _p(db),
                                 csum.shape[0] if csum is not None else 0, 1 if (bias_accumulate and bias_out is not None) else 0,
                                 _stream()), "ppst_wgrad_scatter")
    if not want_bias:
        return dw
    if csum is None:
        db = colsum(dy.as_strided((dy.shape[0] * oh * ow, cout), (dy_ld, 1)), out=bias_out, accumulate=bias_accumulate)
    return dw, db


def wgrad_small_cin(x, dy, scale, out=None, accumulate=False):
    """FromRGB: x (B,H,W,cin<=4), dy (B,H,W,cout) -> dW (cout, cin, 1, 1)."""
    in_ld = _nhwc_ld(x)
    B, H, W, cin = x.shape
    cout = dy.shape[3]
    assert _nhwc_ld(dy, "dy", half_ok=True) == cout
    ws = torch.empty(lib.ppst_wgrad_small_cin_ws(B * H * W, cin, cout) // 4, device=x.device, dtype=torch.float32)
    dw = _grad_out(out, (cout, cin, 1, 1), x)
    check(lib.ppst_wgrad_small_cin_st(_p(x), _p(dy), _p(dw), _p(ws), B * H * W, cin, in_ld, cout, float(scale),
                                      1 if (accumulate and out is not None) else 0, _ST[dy.dtype], _stream()), "ppst_wgrad_small_cin")
    return dw


def colsum(x2d, scale=1.0, out=None, accumulate=False):
    """x2d (rows, C) [row stride ld] -> (C,) column sums (bias gradients); x2d fp32 or half-stored, the sums fp32."""
    _chk_act(x2d)
    rows, C = x2d.shape
    ld = x2d.stride(0)
    assert x2d.stride(1) == 1
    ws = torch.empty(lib.ppst_colsum_ws(rows, C) // 4, device=x2d.device, dtype=torch.float32)
    dst = _grad_out(out, (C,), x2d)
    check(lib.ppst_colsum_st(_p(x2d), _p(dst), _p(ws), rows, C, ld, float(scale), 1 if (accumulate and out is not None) else 0,
                             _ST[x2d.dtype], _stream()), "ppst_colsum")
    return dst


def linear_wgrad(dy, x, scale=1.0, out=None, accumulate=False):
    _chk(dy); _chk(x)
    dy, x = dy.contiguous(), x.contiguous()
    B, N = dy.shape
    K = x.shape[1]
    dw = _grad_out(out, (N, K), x)
    check(lib.ppst_linear_wgrad(_p(dy), _p(x), _p(dw), B, N, K, float(scale), 1 if (accumulate and out is not None) else 0, _stream()),
          "ppst_linear_wgrad")
    return dw


def linear_wgrad_fused(dy, x, scale=1.0, out=None, accumulate=False, relu_in=False, bias_out=None, bias_scale=1.0, bias_accumulate=False,
                       want_bias=False):
    """linear_wgrad with the passes around it folded in (ppst_linear_wgrad_fused): ``relu_in`` reads x as max(x, 0); the bias
    gradient bias_scale * sum_b dy[b] comes out of the same launch (``bias_out`` / ``want_bias``).  Returns (dW, db or None)."""
    _chk(dy); _chk(x)
    dy, x = dy.contiguous(), x.contiguous()
    B, N = dy.shape
    K = x.shape[1]
    dw = _grad_out(out, (N, K), x)
    db = _grad_out(bias_out, (N,), x) if (want_bias or bias_out is not None) else None
    check(lib.ppst_linear_wgrad_fused(_p(dy), _p(x), _p(dw), _p(db), B, N, K, float(scale), float(bias_scale),
                                      1 if (accumulate and out is not None) else 0, 1 if (bias_accumulate and bias_out is not None) else 0,
                                      1 if relu_in else 0, _stream()), "ppst_linear_wgrad_fused")
    return dw, db


def linear_dgrad_gate(dy, w, gate, scale=1.0):
    """linear_dgrad whose result is multiplied by [gate > 0] in the slice reduction (the nn.ReLU in front of the linear)."""
    _chk(dy); _chk(w); _chk(gate)
    dy, w, gate = dy.contiguous(), w.detach().contiguous(), gate.contiguous()
    B, N = dy.shape
    K = w.shape[1]
    assert tuple(gate.shape) == (B, K)
    dx = torch.empty((B, K), device=dy.device, dtype=torch.float32)
    ws = torch.empty(lib.ppst_linear_dgrad_ws(B, N, K) // 4, device=dy.device, dtype=torch.float32)
    check(lib.ppst_linear_dgrad_gate(_p(dy), _p(w), _p(dx), _p(ws), _p(gate), B, N, K, float(scale), _stream()), "ppst_linear_dgrad_gate")
    return dx


def linear_dgrad(dy, w, scale=1.0):
    _chk(dy); _chk(w)
    dy, w = dy.contiguous(), w.detach().contiguous()
    B, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((B, K), device=dy.device, dtype=torch.float32)
    ws = torch.empty(lib.ppst_linear_dgrad_ws(B, N, K) // 4, device=dy.device, dtype=torch.float32)
    check(lib.ppst_linear_dgrad(_p(dy), _p(w), _p(dx), _p(ws), B, N, K, float(scale), _stream()), "ppst_linear_dgrad")
    return dx


def lsgan(pred, target, weight):
    """(loss (1,), dloss/dpred) of weight*mean((pred-target)^2)."""
    _chk(pred)
    pred = pred.contiguous()
    loss = torch.empty((1,), device=pred.device, dtype=torch.float32)
    grad = torch.empty_like(pred)
    check(lib.ppst_lsgan(_p(pred), _p(loss), _p(grad), pred.numel(), float(target), float(weight), _stream()), "ppst_lsgan")
    return loss, grad


def l1_mean(a, b, weight=1.0):
    """weight * mean|a - b| -> (1,) tensor (torch.nn.L1Loss)."""
    _chk(a); _chk(b)
    a, b = a.contiguous(), b.contiguous()
    assert a.shape == b.shape
    n = a.numel()
    ws = torch.empty(lib.ppst_l1_mean_ws(n) // 4, device=a.device, dtype=torch.float32)
    out = torch.empty((1,), device=a.device, dtype=torch.float32)
    check(lib.ppst_l1_mean(_p(a), _p(b), _p(out), _p(ws), n, float(weight), _stream()), "ppst_l1_mean")
    return out


def rscl_loss(q, k, k0, queue, nce_T=0.07):
    """rsclLoss.forward: q, k (n, C), k0 (n0, C), queue (C, K) -> (1,) mean NCE loss."""
    for t in (q, k, k0, queue):
        _chk(t)
    q, k, k0, queue = q.contiguous(), k.contiguous(), k0.contiguous(), queue.contiguous()
    n, C = q.shape
    ws = torch.empty((n,), device=q.device, dtype=torch.float32)
    out = torch.empty((1,), device=q.device, dtype=torch.float32)
    check(lib.ppst_rscl_loss(_p(q), _p(k), _p(k0), _p(queue), _p(out), _p(ws), n, k0.shape[0], C, queue.shape[1], float(nce_T), _stream()),
          "ppst_rscl_loss")
    return out


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step):
    for t in (p, g, m, v):
        _chk(t)
        assert t.is_contiguous()
    check(lib.ppst_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step),
                             _stream()), "ppst_adam_step")


def conv1x1_small_cin(x, w, bias, wscale, act, out_dtype=torch.float32):
    ld = _nhwc_ld(x)
    B, H, W, cin = x.shape
    cout = w.shape[0]
    y = torch.empty((B, H, W, cout), device=x.device, dtype=out_dtype)
    w2 = w.detach().reshape(cout, cin).contiguous()
    check(lib.ppst_conv1x1_small_cin_st(_p(x), _p(w2), _p(bias), _p(y), B * H * W, cin, ld, cout, float(wscale), act,
                                        _ST[out_dtype], _stream()), "ppst_conv1x1_small_cin")
    return y


def conv1x1_small_cout(x, w, bias, wscale):
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, cin = x.shape
    assert ld == cin
    cout = w.shape[0]
    y = torch.empty((B, H, W, cout), device=x.device, dtype=torch.float32)
    w2 = w.detach().reshape(cout, cin).contiguous()
    check(lib.ppst_conv1x1_small_cout_st(_p(x), _p(w2), _p(bias), _p(y), B * H * W, cin, cout, float(wscale), _ST[x.dtype], _stream()),
          "ppst_conv1x1_small_cout")
    return y


def torgb_apply(x, scale_shift, res, out_scale, w, bias, wscale):
    """ToRGB's 1x1 conv reading (a*x + s + bilinear_x2(res)) * out_scale (ppst_torgb_apply_st): x (B,H,W,C) dense, res (B,H/2,W/2,C)
    or None (x's storage type) -> (B,H,W,3) fp32."""
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, cin = x.shape
    if ld != cin or w.shape[0] != 3:
        raise RuntimeError("torgb_apply needs a dense input and a 3-channel weight")
    _chk(scale_shift, "scale_shift"); _chk(bias, "bias")
    res_ld = 0
    if res is not None:
        res_ld = _nhwc_ld(res, "res", half_ok=True)
        if res.dtype != x.dtype or tuple(res.shape) != (B, H // 2, W // 2, cin):
            raise RuntimeError("torgb_apply: res must be the half-resolution tensor of x's type")
    if tuple(scale_shift.shape) != (B, cin, 2) or not scale_shift.is_contiguous():
        raise RuntimeError("torgb_apply: scale_shift must be a contiguous (B, C, 2) table")
    y = torch.empty((B, H, W, 3), device=x.device, dtype=torch.float32)
    w2 = w.detach().reshape(3, cin).contiguous()
    check(lib.ppst_torgb_apply_st(_p(x), _p(scale_shift), _p(res), res_ld, float(out_scale), _p(w2), _p(bias), _p(y), B, H, W, cin,
                                  float(wscale), _ST[x.dtype], _stream()), "ppst_torgb_apply")
    return y


def upscale_weight_bwd(dw4, cout, cin, scale=1.0, out=None, accumulate=False):
    """adjoint of the fused-upscale weight blur (stylegan2_layers.py:314-319): dw4 (Cin,Cout,4,4) -> (Cout,Cin,3,3)."""
    _chk(dw4)
    dw = _grad_out(out, (cout, cin, 3, 3), dw4)
    check(lib.ppst_upscale_weight_bwd(_p(dw4.contiguous()), _p(dw), cout, cin, float(scale), 1 if (accumulate and out is not None) else 0,
                                      _stream()), "ppst_upscale_weight_bwd")
    return dw


# round 5: thin stride-2 layers (forward Cin <= max_cin) take their input gradient as the phase-stacked stride-1 conv + depth_to_space
# (plan kind "dgrad_s2ds") instead of the four-group scattered form ("dgrad_s2d").  PPST_DGRAD_S2D_STACK=0: the four-group form.
DGRAD_S2D_STACK = {"value": os.environ.get("PPST_DGRAD_S2D_STACK", "1") != "0", "max_cin": int(os.environ.get("PPST_DGRAD_S2D_MAXCIN", "64"))}


def depth_to_space(x, out_hw):
    """x (B, th, tw, 4 C) with channel block py*2+px = output phase -> (B, oh, ow, C), y[b, 2q+py, 2p+px] = x[b, q, p, phase block]."""
    ld = _nhwc_ld(x, "depth_to_space input", half_ok=True)
    B, th, tw, c4 = x.shape
    if ld != c4 or c4 % 4:
        raise RuntimeError("depth_to_space needs a dense tensor with 4 C channels")
    oh, ow = out_hw
    y = torch.empty((B, oh, ow, c4 // 4), device=x.device, dtype=x.dtype)
    check(lib.ppst_depth_to_space_st(_p(x), _p(y), B, th, tw, oh, ow, c4 // 4, _ST[x.dtype], _stream()), "ppst_depth_to_space_st")
    return y


def dgrad_s2d(net, wname, scale, g, out_hw):
    """Input gradient of the stride-2 3x3 conv ``wname`` of ``net`` (on the blurred tensor of extent ``out_hw``) from the gradient g at
    its output: the phase-stacked form for thin layers (ops.DGRAD_S2D_STACK), else the four-group scattered form."""
    cin = net.p(wname).shape[1]
    per = 8 if g.dtype != torch.float32 else 4
    if DGRAD_S2D_STACK["value"] and cin <= DGRAD_S2D_STACK["max_cin"] and cin % per == 0:
        oh, ow = out_hw
        ys = net.plan(wname, "dgrad_s2ds", scale)(g, out_hw=((oh + 1) // 2, (ow + 1) // 2))
        return depth_to_space(ys, out_hw)
    return net.plan(wname, "dgrad_s2d", scale)(g, out_hw=out_hw)


def space_to_depth(x):
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), device=x.device, dtype=x.dtype)
    check(lib.ppst_space_to_depth_st(_p(x), _p(y), B, H, W, C, ld, _ST[x.dtype], _stream()), "ppst_space_to_depth")
    return y


# ------------------------------------------------- instance norm / affine ----
def in_stats(x, rep_pad=False):
    ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    n = ctypes.c_int(0)
    check(lib.ppst_in_stats(None, None, B, H, W, C, ld, 0, ctypes.byref(n), None), "ppst_in_stats(size)")
    part = torch.empty((B, n.value, C, 2), device=x.device, dtype=torch.float32)
    check(lib.ppst_in_stats(_p(x), _p(part), B, H, W, C, ld, 1 if rep_pad else 0, ctypes.byref(n), _stream()), "ppst_in_stats")
    return part


def in_finalize(partial, count, style=None, post_bias=None, eps=1e-5):
    """partial (B, n, C, 2) -> scale_shift (B, C, 2); style (B, 2C) optional (StyleMod);
    post_bias (C,) optional, added to the shift (activation bias after the norm)."""
    _chk(partial)
    _chk(style, "style")
    _chk(post_bias, "post_bias")
    B, n, C, _ = partial.shape
    style_ld = 0
    if style is not None:  # (B, 2C) rows, possibly a column slice of a wider batched GEMV output
        assert style.shape == (B, 2 * C) and style.stride(1) == 1
        style_ld = style.stride(0) if B > 1 else max(style.stride(0), 2 * C)
    ss = torch.empty((B, C, 2), device=partial.device, dtype=torch.float32)
    check(lib.ppst_in_finalize(_p(partial), n, _p(style), style_ld, _p(post_bias), _p(ss), B, C, float(count), float(eps),
                               _stream()), "ppst_in_finalize")
    return ss


def in_finalize_train(partial, count, style=None, post_bias=None, eps=1e-5):
    """in_finalize that also returns mean_rstd (B, C, 2) for the backward of the norm."""
    _chk(partial); _chk(style, "style"); _chk(post_bias, "post_bias")
    B, n, C, _ = partial.shape
    style_ld = 0
    if style is not None:
        assert style.shape == (B, 2 * C) and style.stride(1) == 1
        style_ld = style.stride(0) if B > 1 else max(style.stride(0), 2 * C)
    ss = torch.empty((B, C, 2), device=partial.device, dtype=torch.float32)
    mr = torch.empty((B, C, 2), device=partial.device, dtype=torch.float32)
    check(lib.ppst_in_finalize_train(_p(partial), n, _p(style), style_ld, _p(post_bias), _p(ss), _p(mr), B, C, float(count), float(eps),
                                     _stream()), "ppst_in_finalize_train")
    return ss, mr


def dual_stats(g, y, gate=None):
    """per-(b, c) partial sums (sum g', sum g'*y), g' = g * lrelu'(gate) when gate is given."""
    g_ld, y_ld = _nhwc_ld(g, "g", half_ok=True), _nhwc_ld(y, "y", half_ok=True)
    gate_ld = _nhwc_ld(gate, "gate", half_ok=True) if gate is not None else 0
    B, H, W, C = g.shape
    assert y.shape == g.shape
    if y.dtype != g.dtype or (gate is not None and gate.dtype != g.dtype):
        raise RuntimeError("dual_stats: g, y and gate share one storage type (got %s / %s / %s)" % (g.dtype, y.dtype, None if gate is None else gate.dtype))
    n = ctypes.c_int(0)
    check(lib.ppst_dual_stats(None, None, None, None, B, H * W, C, g_ld, y_ld, gate_ld, ctypes.byref(n), None), "ppst_dual_stats(size)")
    part = torch.empty((B, n.value, C, 2), device=g.device, dtype=torch.float32)
    check(lib.ppst_dual_stats_st(_p(g), _p(y), _p(gate), _p(part), B, H * W, C, g_ld, y_ld, gate_ld, ctypes.byref(n), _ST[g.dtype], _stream()),
          "ppst_dual_stats")
    return part


def in_bwd_finalize(partial, count, mean_rstd=None, style=None, want_dstyle=False):
    """-> (coef (B,C,4) or None, dstyle (B,2C) or None); see ppst_in_bwd_finalize."""
    _chk(partial); _chk(mean_rstd); _chk(style)
    B, n, C, _ = partial.shape
    style_ld = 0
    if style is not None:
        assert style.shape[0] == B and style.shape[1] >= C and style.stride(1) == 1
        style_ld = style.stride(0) if B > 1 else max(style.stride(0), C)
    coef = torch.empty((B, C, 4), device=partial.device, dtype=torch.float32) if mean_rstd is not None else None
    dstyle = torch.empty((B, 2 * C), device=partial.device, dtype=torch.float32) if (want_dstyle or mean_rstd is None) else None
    check(lib.ppst_in_bwd_finalize(_p(partial), n, _p(mean_rstd), _p(style), style_ld, _p(coef), _p(dstyle), B, C, float(count),
                                   _stream()), "ppst_in_bwd_finalize")
    return coef, dstyle


def in_bwd_apply(g, y, coef, gate=None, post_gate=False):
    g_ld, y_ld = _nhwc_ld(g, "g", half_ok=True), _nhwc_ld(y, "y", half_ok=True)
    gate_ld = _nhwc_ld(gate, "gate", half_ok=True) if gate is not None else 0
    B, H, W, C = g.shape
    if y.dtype != g.dtype or (gate is not None and gate.dtype != g.dtype):
        raise RuntimeError("in_bwd_apply: g, y and gate share one storage type")
    dx = torch.empty((B, H, W, C), device=g.device, dtype=g.dtype)
    check(lib.ppst_in_bwd_apply_st(_p(g), _p(y), _p(gate), _p(coef), _p(dx), B, H * W, C, g_ld, y_ld, gate_ld, C, 1 if post_gate else 0,
                                   _ST[g.dtype], _stream()), "ppst_in_bwd_apply")
    return dx


def prelu_bwd(g, y, prelu, scale_shift=None, res=None):
    """z = a*y + s [+ res]; returns (g * prelu'(z) dense, dslope (1,))."""
    g_ld, y_ld = _nhwc_ld(g, "g"), _nhwc_ld(y, "y")
    res_ld = _nhwc_ld(res, "res") if res is not None else 0
    _chk(prelu); _chk(scale_shift)
    B, H, W, C = g.shape
    total = B * H * W * C
    ws = torch.empty(lib.ppst_prelu_bwd_ws(total) // 4, device=g.device, dtype=torch.float32)
    gpre = torch.empty((B, H, W, C), device=g.device, dtype=torch.float32)
    check(lib.ppst_prelu_bwd(_p(g), _p(y), _p(scale_shift), _p(res), _p(prelu), _p(gpre), _p(ws), B, H * W, C, g_ld, y_ld, res_ld,
                             _stream()), "ppst_prelu_bwd")
    ds = torch.empty((1,), device=g.device, dtype=torch.float32)
    check(lib.ppst_sum_partials(_p(ws), _p(ds), ws.numel(), 1.0, _stream()), "ppst_sum_partials")
    return gpre, ds


def pad2d(x, py0, py1, px0, px1, mode):
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    y = torch.empty((B, H + py0 + py1, W + px0 + px1, C), device=x.device, dtype=x.dtype)
    check(lib.ppst_pad2d_st(_p(x), _p(y), B, H, W, C, ld, py0, py1, px0, px1, mode, _ST[x.dtype], _stream()), "ppst_pad2d")
    return y


def pad2d_bwd(dy, py0, py1, px0, px1, mode):
    _chk_act(dy)
    dy = dy.contiguous()
    B, OH, OW, C = dy.shape
    H, W = OH - py0 - py1, OW - px0 - px1
    dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
    check(lib.ppst_pad2d_bwd_st(_p(dy), _p(dx), B, H, W, C, py0, py1, px0, px1, mode, _ST[dy.dtype], _stream()), "ppst_pad2d_bwd")
    return dx


def bilinear_bwd(dy, H, W):
    dy_ld = _nhwc_ld(dy, half_ok=True)
    B, OH, OW, C = dy.shape
    dx = torch.zeros((B, H, W, C), device=dy.device, dtype=dy.dtype)
    check(lib.ppst_bilinear_bwd_st(_p(dy), _p(dx), B, H, W, C, C, OH, OW, dy_ld, _ST[dy.dtype], _stream()), "ppst_bilinear_bwd")
    return dx


def avgpool_bwd(dy, f):
    dy_ld = _nhwc_ld(dy)
    B, oh, ow, C = dy.shape
    dx = torch.empty((B, oh * f, ow * f, C), device=dy.device, dtype=torch.float32)
    check(lib.ppst_avgpool_bwd(_p(dy), _p(dx), B, oh * f, ow * f, C, C, f, dy_ld, _stream()), "ppst_avgpool_bwd")
    return dx


def gap_gmp_bwd(x, mask, v, g, out=None):
    """adjoint of gap_gmp; out given -> accumulate into it.  The gradient takes x's storage type."""
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    _chk(mask); _chk(v); _chk(g)
    acc = out is not None
    if out is None:
        out = torch.empty((B, H, W, C), device=x.device, dtype=x.dtype)
    arg = torch.empty((B, C), device=x.device, dtype=torch.int32)
    check(lib.ppst_gap_gmp_bwd_st(_p(x), _p(mask), _p(v.contiguous()), _p(g.contiguous()), _p(out), ctypes.c_void_p(arg.data_ptr()), B, H * W, C,
                                  ld, 1 if acc else 0, _ST[x.dtype], _stream()), "ppst_gap_gmp_bwd")
    return out


def gap_gmp_multi(x, masks, with_plain=True):
    """GAP || GMP of x * mask for every channel of ``masks`` (B,H,W,nm) -- and unmasked first when with_plain -- in one read of x:
    -> ((nm + with_plain) * B, 2C), head-major (ppst_gap_gmp_multi)."""
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    _chk(masks, "masks")
    masks = masks.contiguous()
    nm = masks.shape[3]
    assert tuple(masks.shape[:3]) == (B, H, W) and 1 <= nm <= 3
    heads = nm + (1 if with_plain else 0)
    ws = torch.empty(lib.ppst_gap_gmp_multi_ws(B, H * W, C, heads) // 4, device=x.device, dtype=torch.float32)
    out = torch.empty((heads * B, 2 * C), device=x.device, dtype=torch.float32)
    check(lib.ppst_gap_gmp_multi(_p(x), _p(masks), _p(out), _p(ws), B, H, W, C, ld, nm, 1 if with_plain else 0, _ST[x.dtype], _stream()),
          "ppst_gap_gmp_multi")
    return out


def gap_gmp_multi_bwd(x, masks, v, g, with_plain=True, out=None):
    """adjoint of gap_gmp_multi: the sum over the heads, one pass (out given -> accumulate into it)."""
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    _chk(masks); _chk(v); _chk(g)
    masks = masks.contiguous()
    nm = masks.shape[3]
    heads = nm + (1 if with_plain else 0)
    acc = out is not None
    if out is None:
        out = torch.empty((B, H, W, C), device=x.device, dtype=x.dtype)
    arg = torch.empty((heads * B, C), device=x.device, dtype=torch.int32)
    check(lib.ppst_gap_gmp_multi_bwd(_p(x), _p(masks), _p(v.contiguous()), _p(g.contiguous()), _p(out), ctypes.c_void_p(arg.data_ptr()), B,
                                     H * W, C, ld, nm, 1 if with_plain else 0, 1 if acc else 0, _ST[x.dtype], _stream()),
          "ppst_gap_gmp_multi_bwd")
    return out


def l2norm_rows_bwd(g, x, eps, mode):
    _chk(g); _chk(x)
    g, x = g.contiguous(), x.contiguous()
    B, K = x.shape
    dx = torch.empty_like(x)
    check(lib.ppst_l2norm_rows_bwd(_p(g), _p(x), _p(dx), B, K, float(eps), mode, _stream()), "ppst_l2norm_rows_bwd")
    return dx


def softmax_rows_bwd_(p, g, div=1.0):
    """in place on g: g <- p * (g - sum(g*p)) / div."""
    _chk(p); _chk(g)
    assert p.is_contiguous() and g.is_contiguous() and p.shape == g.shape
    cols = p.shape[-1]
    check(lib.ppst_softmax_rows_bwd(_p(p), _p(g), p.numel() // cols, cols, float(div), _stream()), "ppst_softmax_rows_bwd")
    return g


def corr_prep_bwd(g, x, ncenter=256):
    _chk(g); _chk(x)
    g, x = g.contiguous(), x.contiguous()
    B, P, C = x.shape
    dx = torch.empty_like(x)
    check(lib.ppst_corr_prep_bwd(_p(g), _p(x), _p(dx), B * P, C, ncenter, 2.220446049250313e-16, _stream()), "ppst_corr_prep_bwd")
    return dx


def l1_grad(a, b, weight=1.0):
    _chk(a); _chk(b)
    a, b = a.contiguous(), b.contiguous()
    da = torch.empty_like(a)
    check(lib.ppst_l1_grad(_p(a), _p(b), _p(da), a.numel(), float(weight), _stream()), "ppst_l1_grad")
    return da


def rscl_loss_bwd(q, k, k0, queue, gout, nce_T=0.07):
    for t in (q, k, k0, queue, gout):
        _chk(t)
    q, k, k0, queue = q.contiguous(), k.contiguous(), k0.contiguous(), queue.contiguous()
    n, C = q.shape
    dq = torch.empty_like(q)
    check(lib.ppst_rscl_loss_bwd(_p(q), _p(k), _p(k0), _p(queue), _p(gout.contiguous()), _p(dq), n, k0.shape[0], C, queue.shape[1],
                                 float(nce_T), _stream()), "ppst_rscl_loss_bwd")
    return dq


def rselfcorr_bwd(fea, dout):
    ld = _nhwc_ld(fea)
    B, H, W, C = fea.shape
    assert ld == C
    dout_ld = _nhwc_ld(dout)
    dfea = torch.empty_like(fea)
    check(lib.ppst_rselfcorr_bwd(_p(fea), _p(dout), _p(dfea), B, H, W, C, dout_ld, _stream()), "ppst_rselfcorr_bwd")
    return dfea


def transpose_last2(x):
    """(b, M, N) -> (b, N, M) contiguous (the layout-transpose kernel)."""
    _chk(x)
    x = x.contiguous()
    b, M, N = x.shape
    y = torch.empty((b, N, M), device=x.device, dtype=torch.float32)
    check(lib.ppst_nchw_to_nhwc(_p(x), _p(y), b, M, 1, N, _stream()), "ppst_nchw_to_nhwc")
    return y


def scale_by(x, s):
    """x * s[0] with s a (1,) device tensor."""
    _chk(x); _chk(s)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.ppst_scale_by(_p(x), _p(s.contiguous()), _p(y), x.numel(), _stream()), "ppst_scale_by")
    return y


def noise_wgrad(dpre, noise, out=None, accumulate=False):
    ld = _nhwc_ld(dpre, half_ok=True)
    B, H, W, C = dpre.shape
    _chk(noise)
    noise = noise.contiguous()
    assert noise.numel() == B * H * W
    ws = torch.empty(lib.ppst_noise_wgrad_ws(B * H * W) // 4, device=dpre.device, dtype=torch.float32)
    dst = _grad_out(out, (1,), dpre)
    check(lib.ppst_noise_wgrad_st(_p(dpre), _p(noise), _p(dst), _p(ws), B * H * W, C, ld, 1 if (accumulate and out is not None) else 0,
                                  _ST[dpre.dtype], _stream()), "ppst_noise_wgrad")
    return dst


def affine_act(x, scale_shift=None, res=None, act=ACT_NONE, prelu=None, out_scale=1.0, res_before_act=False, out=None,
               res_scale_shift=None, res_up2=False, out_dtype=None):
    """res_up2: ``res`` is a half-resolution tensor, bilinearly upsampled x2 on the fly.  x (and res, which must share its type)
    may be half-stored; the output takes out_dtype (default: x's type)."""
    x_ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, H, W, C), device=x.device, dtype=out_dtype or x.dtype)
    y_ld = _nhwc_ld(out, half_ok=True)
    res_ld = _nhwc_ld(res, "res", half_ok=True) if res is not None else 0
    if res is not None and res.dtype != x.dtype:
        raise RuntimeError("affine_act: res must have the storage type of x (%s), got %s" % (x.dtype, res.dtype))
    _chk(scale_shift, "scale_shift")
    _chk(res_scale_shift, "res_scale_shift")
    _chk(prelu, "prelu")
    if res_up2:
        assert res.shape[1] * 2 == H and res.shape[2] * 2 == W
    flag = act | (RES_BEFORE_ACT if res_before_act else 0)
    check(lib.ppst_affine_act_st(_p(x), _p(scale_shift), _p(res), _p(res_scale_shift), _p(out), B, H * W, C, x_ld, res_ld, y_ld,
                                 flag, _p(prelu), float(out_scale), W if res_up2 else 0, _ST[x.dtype], _ST[out.dtype], _stream()),
          "ppst_affine_act")
    return out


def affine_act_stats(x, scale_shift=None, res=None, act=ACT_NONE, prelu=None, out_scale=1.0, res_before_act=False,
                     res_scale_shift=None, rep_pad=False, res_up2=False):
    """affine_act that also returns the instance-norm partials (B, n, C, 2) of its output."""
    x_ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    out = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    res_ld = _nhwc_ld(res, "res") if res is not None else 0
    for t in (scale_shift, res_scale_shift, prelu):
        _chk(t)
    n = ctypes.c_int(0)
    check(lib.ppst_in_stats(None, None, B, H, W, C, x_ld, 0, ctypes.byref(n), None), "ppst_in_stats(size)")
    part = torch.empty((B, n.value, C, 2), device=x.device, dtype=torch.float32)
    flag = act | (RES_BEFORE_ACT if res_before_act else 0)
    check(lib.ppst_affine_act_stats(_p(x), _p(scale_shift), _p(res), _p(res_scale_shift), _p(out), _p(part), B, H, W, C, x_ld,
                                    res_ld, C, flag, _p(prelu), float(out_scale), 1 if rep_pad else 0, 1 if res_up2 else 0,
                                    _stream()),
          "ppst_affine_act_stats")
    return out, part


def upsample_nearest2(x):
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    assert ld == C
    y = torch.empty((B, 2 * H, 2 * W, C), device=x.device, dtype=x.dtype)
    check(lib.ppst_upsample_nearest2_st(_p(x), _p(y), B, H, W, C, _ST[x.dtype], _stream()), "ppst_upsample_nearest2")
    return y


# ------------------------------------------------------- pooling / resize ----
def gap_gmp(x, mask=None):
    """(B,H,W,C) -> (B, 2C) = cat(mean, max) over pixels; mask (B,H,W) optional multiplier."""
    ld = _nhwc_ld(x, half_ok=True)
    B, H, W, C = x.shape
    _chk(mask, "mask")
    ws = torch.empty(lib.ppst_gap_gmp_ws(B, H * W, C) // 4, device=x.device, dtype=torch.float32)
    out = torch.empty((B, 2 * C), device=x.device, dtype=torch.float32)
    check(lib.ppst_gap_gmp_st(_p(x), _p(mask), _p(out), _p(ws), B, H, W, C, ld, _ST[x.dtype], _stream()), "ppst_gap_gmp")
    return out


def avgpool(x, f, out=None):
    x_ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, H // f, W // f, C), device=x.device, dtype=torch.float32)
    y_ld = _nhwc_ld(out)
    check(lib.ppst_avgpool(_p(x), _p(out), B, H, W, C, x_ld, f, y_ld, _stream()), "ppst_avgpool")
    return out


def bilinear(x, OH, OW, out=None):
    x_ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, OH, OW, C), device=x.device, dtype=torch.float32)
    y_ld = _nhwc_ld(out)
    check(lib.ppst_bilinear(_p(x), _p(out), B, H, W, C, x_ld, OH, OW, y_ld, _stream()), "ppst_bilinear")
    return out


def head_tail(x, scale_shift, feat, feat1, act=ACT_NONE, prelu=None):
    """feat <- avgpool(act(a*x+s)) to feat's grid; feat1 <- the same f resized by the exact factor 1 or 2 (see ppst_head_tail)."""
    x_ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    P, D = H // feat.shape[1], H // feat1.shape[1]
    _chk(scale_shift, "scale_shift"); _chk(prelu, "prelu")
    check(lib.ppst_head_tail(_p(x), _p(scale_shift), _p(prelu), _p(feat), _p(feat1), B, H, W, C, x_ld, _nhwc_ld(feat), _nhwc_ld(feat1),
                             P, D, act, _stream()), "ppst_head_tail")


def maxpool2(x):
    ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    assert ld == C
    y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
    check(lib.ppst_maxpool2(_p(x), _p(y), B, H, W, C, _stream()), "ppst_maxpool2")
    return y


# ----------------------------------------------------------------- linear ----
def linear(x, w, bias=None, wscale=1.0, bscale=1.0, relu_in=False, act=ACT_NONE):
    _chk(x, "x"); _chk(w, "weight"); _chk(bias, "bias")
    x = x.contiguous()
    w = w.detach().contiguous()
    B, K = x.shape
    N = w.shape[0]
    assert w.numel() == N * K
    y = torch.empty((B, N), device=x.device, dtype=torch.float32)
    check(lib.ppst_linear(_p(x), _p(w), _p(bias), _p(y), B, K, N, float(wscale), float(bscale), 1 if relu_in else 0, act,
                          _stream()), "ppst_linear")
    return y


def l2norm_rows(x, eps, mode):
    _chk(x)
    x = x.contiguous()
    B, K = x.shape
    y = torch.empty_like(x)
    check(lib.ppst_l2norm_rows(_p(x), _p(y), B, K, float(eps), mode, _stream()), "ppst_l2norm_rows")
    return y


def lerp(a, b, r):
    _chk(a); _chk(b)
    a, b = a.contiguous(), b.contiguous()
    y = torch.empty_like(a)
    check(lib.ppst_lerp(_p(a), _p(b), _p(y), a.numel(), float(r), _stream()), "ppst_lerp")
    return y


def spatial_modulation(x, scale, bias, out_dtype=torch.float32):
    ld = _nhwc_ld(x)
    B, H, W, C = x.shape
    assert ld == C
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    check(lib.ppst_spatial_modulation_st(_p(x), _p(scale.contiguous()), _p(bias.contiguous()), _p(y), B, H * W, C, _ST[out_dtype],
                                         _stream()), "ppst_spatial_modulation")
    return y


# --------------------------------------------------------- correspondence ----
def rselfcorr(fea, out=None):
    ld = _nhwc_ld(fea)
    B, H, W, C = fea.shape
    assert ld == C
    if out is None:
        out = torch.empty((B, H // 4, W // 4, 256), device=fea.device, dtype=torch.float32)
    out_ld = _nhwc_ld(out)
    check(lib.ppst_rselfcorr(_p(fea), _p(out), B, H, W, C, out_ld, _stream()), "ppst_rselfcorr")
    return out


def corr_prep(fea, ncenter=256):
    """fea (B, P, C) dense -> centred / L2-normalised rows."""
    _chk(fea)
    fea = fea.contiguous()
    B, P, C = fea.shape
    y = torch.empty_like(fea)
    check(lib.ppst_corr_prep(_p(fea), _p(y), B, P, C, ncenter, _stream()), "ppst_corr_prep")
    return y


def unfold_rows(x, k):
    """F.unfold(x, k, padding=k // 2) of an NHWC map (B,H,W,C) as rows (B, H*W, C*k*k) (ppst_model.py:345-347; k odd)."""
    _chk(x)
    x = x.contiguous()
    B, H, W, C = x.shape
    out = torch.empty((B, H * W, C * k * k), device=x.device, dtype=torch.float32)
    check(lib.ppst_unfold_rows(_p(x), _p(out), B, H, W, C, int(k), _stream()), "ppst_unfold_rows")
    return out


def unfold_rows_bwd(g, shape, k):
    """gradient of unfold_rows: g (B, H*W, C*k*k) -> (B,H,W,C)."""
    _chk(g)
    g = g.contiguous()
    B, H, W, C = shape
    dx = torch.empty((B, H, W, C), device=g.device, dtype=torch.float32)
    check(lib.ppst_unfold_rows_bwd(_p(g), _p(dx), B, H, W, C, int(k), _stream()), "ppst_unfold_rows_bwd")
    return dx


GEMM_MODE = {"value": None}    # None: per call site ("x6" fp32-class / "x3" bf16x3); "f32": the exact-fp32 MFMA kernels everywhere


def _gemm_passes(mode, K):
    """0 = exact-fp32 MFMA kernel, 6 / 3 = bf16 planes (ppst_gemm_*_split).  Exact-conv mode (precision 2) and shapes the
    split kernels do not take (K % 16 / K % 32) use the fp32 kernel."""
    mode = GEMM_MODE["value"] or mode or "x6"
    if mode not in ("f32", "x6", "x3"):
        raise ValueError("gemm mode %r" % (mode,))
    if mode == "f32" or PRECISION["value"] == 2:
        return 0
    if mode == "x3" and K % 32 == 0:
        return 3
    return 6 if K % 16 == 0 else 0


def gemm_nt(A, Bm, alpha=1.0, mode=None):
    """A (b,M,K), Bm (b,N,K) -> (b,M,N) = alpha * A @ Bm^T.  mode: "x6" (default: three bf16 planes per operand, fp32-class),
    "x3" (bf16x3, the convs' accuracy class), "f32" (fp32 MFMA)."""
    _chk(A); _chk(Bm)
    A, Bm = A.contiguous(), Bm.contiguous()
    b, M, K = A.shape
    N = Bm.shape[1]
    C = torch.empty((b, M, N), device=A.device, dtype=torch.float32)
    passes = _gemm_passes(mode, K)
    if passes:
        check(lib.ppst_gemm_nt_split(_p(A), _p(Bm), _p(C), b, M, N, K, float(alpha), passes, _stream()), "ppst_gemm_nt_split")
    else:
        check(lib.ppst_gemm_nt_f32(_p(A), _p(Bm), _p(C), b, M, N, K, float(alpha), _stream()), "ppst_gemm_nt_f32")
    return C


def gemm_nn(A, Bm, mode=None):
    """A (b,M,K), Bm (b,K,N) -> (b,M,N); mode as gemm_nt."""
    _chk(A); _chk(Bm)
    A, Bm = A.contiguous(), Bm.contiguous()
    b, M, K = A.shape
    N = Bm.shape[2]
    C = torch.empty((b, M, N), device=A.device, dtype=torch.float32)
    passes = _gemm_passes(mode, K) if N % 4 == 0 else 0
    if passes:
        check(lib.ppst_gemm_nn_split(_p(A), _p(Bm), _p(C), b, M, N, K, N, N, passes, _stream()), "ppst_gemm_nn_split")
    else:
        check(lib.ppst_gemm_nn_f32(_p(A), _p(Bm), _p(C), b, M, N, K, N, N, _stream()), "ppst_gemm_nn_f32")
    return C


def softmax_rows_(x, div=1.0):
    _chk(x)
    assert x.is_contiguous()
    cols = x.shape[-1]
    check(lib.ppst_softmax_rows(_p(x), x.numel() // cols, cols, float(div), _stream()), "ppst_softmax_rows")
    return x


def unfold_patches(x, s):
    _chk(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, (H // s) * (W // s), C * s * s), device=x.device, dtype=torch.float32)
    check(lib.ppst_unfold_patches(_p(x), _p(y), B, C, H, W, s, _stream()), "ppst_unfold_patches")
    return y


def fold_patches(x, C, H, W, s):
    _chk(x)
    x = x.contiguous()
    B = x.shape[0]
    y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    check(lib.ppst_fold_patches(_p(x), _p(y), B, C, H, W, s, _stream()), "ppst_fold_patches")
    return y


# ----------------------------------------------------------- post-process ----
def tensor2im_u8(x):
    """NCHW fp32 -> (B,H,W,C) uint8 (util.tensor2im truncation)."""
    _chk(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), device=x.device, dtype=torch.uint8)
    check(lib.ppst_tensor2im_u8(_p(x), _p(y), B, C, H, W, _stream()), "ppst_tensor2im_u8")
    return y


def guided_filter(guide_u8, src_u8, r=30, eps=(0.02 * 255) ** 2, want_u8=False):
    """guide/src (B,H,W,3) uint8 -> fp32 NCHW in [-1,1] (and the uint8 HWC result)."""
    for t in (guide_u8, src_u8):
        if not t.is_cuda or t.dtype != torch.uint8:
            raise RuntimeError("guided_filter needs CUDA uint8 tensors")
    guide_u8, src_u8 = guide_u8.contiguous(), src_u8.contiguous()
    B, H, W, _ = guide_u8.shape
    ws = torch.empty(lib.ppst_guided_filter_ws(B, H, W), device=guide_u8.device, dtype=torch.uint8)
    out = torch.empty((B, 3, H, W), device=guide_u8.device, dtype=torch.float32)
    out_u8 = torch.empty((B, H, W, 3), device=guide_u8.device, dtype=torch.uint8) if want_u8 else None
    check(lib.ppst_guided_filter(_p(guide_u8), _p(src_u8), _p(out), _p(out_u8), B, H, W, r, float(eps), _p(ws), _stream()),
          "ppst_guided_filter")
    return (out, out_u8) if want_u8 else out


# --------------------------------------------------------------- profiling ----
PROF_ON = {"value": False}


def prof_enable(on):
    PROF_ON["value"] = bool(on)
    check(lib.ppst_prof_enable(1 if on else 0), "ppst_prof_enable")


def prof_collect():
    """(ms, launches, flop) of the bracketed conv launches.  Raises if the event pool overflowed (the totals
    would silently miss launches)."""
    ms, n, fl = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
    dropped = lib.ppst_prof_dropped()
    check(lib.ppst_prof_collect(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "ppst_prof_collect")
    if dropped:
        raise RuntimeError("conv profiling pool overflowed: %d launches were not timed (use fewer --steps)" % dropped)
    return ms.value, n.value, fl.value


def prof_detail():
    """list of (ms, flop, (B, tile_h, tile_w, nsteps, cout, n_groups, halo, bn)) per bracketed conv launch."""
    out = []
    i = 0
    while True:
        ms, fl = ctypes.c_double(0), ctypes.c_double(0)
        info = (ctypes.c_int * 8)()
        if lib.ppst_prof_detail(i, ctypes.byref(ms), ctypes.byref(fl), info) != 0:
            break
        out.append((ms.value, fl.value, tuple(info)))
        i += 1
    return out
