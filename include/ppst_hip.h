/*
 * libppst_hip.so -- C ABI of the MI355X (gfx950) PPST hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes and an
 * explicit HIP stream, allocates nothing, keeps no global state (except the
 * opt-in profiling event pool of ppst_prof_*), never throws and returns
 *      0            success
 *      > 0          a hipError_t from the launch
 *      < 0          PPST_E* argument error (nothing was launched)
 * so it can be bound from ctypes / cgo / JNI alike.  Each declaration cites the
 * reference interface (file:line under wangxb29/PPST) it replaces.
 *
 * Tensor layouts: "NCHW" = the reference's contiguous torch layout;
 * "NHWC" = channels-last [B][H][W][C], the internal layout of the fused path.
 * All tensors are fp32 unless stated.  Activation tensors of the single-pass precision
 * modes may be stored as IEEE half / bfloat16: the `*_st` entry points and
 * ppst_conv_args.io_st take a storage type (PPST_ST_* below, the values of the PPST_F32 /
 * PPST_F16 / PPST_BF16 dtype enum); the `dtype` argument of ppst_upfirdn2d and
 * ppst_fused_bias_act takes the same three types (the reference's
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF: upfirdn2d_kernel.cu:225, fused_bias_act_kernel.cu:79;
 * fp32 arithmetic, one rounding; the FIR taps stay fp32) and PPST_F64 (round 5: every tensor
 * of the call -- the FIR taps too -- is double and the arithmetic runs in double, scalar_t = double
 * in the reference's dispatch: what a gradcheck-style caller of the two ops hands over).
 */
#ifndef PPST_HIP_H
#define PPST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPST_OK 0
#define PPST_EINVAL (-1)      /* bad size / flag combination */
#define PPST_EUNSUPPORTED (-2) /* valid but not implemented (e.g. dtype) */
#define PPST_ENULL (-3)       /* null pointer where data is required */

enum { PPST_F32 = 0, PPST_F16 = 1, PPST_BF16 = 2, PPST_F64 = 3 /* ppst_upfirdn2d / ppst_fused_bias_act only */ };

/* padding modes of the fused conv (nn.ReflectionPad2d / ReplicationPad2d /
 * zero padding, stylegan2_layers.py:528-531, generator.py:13-17) */
enum { PPST_PAD_ZERO = 0, PPST_PAD_REFLECT = 1, PPST_PAD_REPLICATE = 2 };

/* epilogue activation of the fused conv */
enum { PPST_ACT_NONE = 0, PPST_ACT_LRELU = 1 /* lrelu(0.2)*sqrt2 */, PPST_ACT_PRELU = 2 };

/* ABI revision: bumped whenever a struct of this header changes size or an entry point changes its arguments.
 *   1: rounds 1-3;  2: round 4-5 (ppst_pack_job gained `dual` -- it is the element of the job ARRAY ppst_conv_pack_batch walks, so its
 *   stride changed --, ppst_conv_args gained dual_b / io_st / k64, PPST_F64 for the two native ops).  A binding built against
 *   another revision must refuse to run (ppst_amd/_lib.py does). */
#define PPST_ABI_VERSION 2
int ppst_version(void);

/* Storage type of an NHWC activation tensor at the entry points that take one (`*_st` twins and ppst_conv_args.io_st; round 4):
 * fp32, IEEE half or bfloat16.  A kernel given a half type computes in fp32 exactly as its fp32 form and rounds once, to nearest
 * even, when it stores: f_st(x_half) == round(f(float(x_half))).  Leading dimensions stay in ELEMENTS; half tensors must be 8-byte
 * aligned and have a channel count (and leading dimensions) divisible by 4.  Each `_st` twin with every type 0 IS its plain form. */
#define PPST_ST_F32 0   /* == PPST_F32 */
#define PPST_ST_F16 1   /* == PPST_F16 */
#define PPST_ST_BF16 2  /* == PPST_BF16 */

/* ---------------------------------------------------------------- ops ----
 * upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel[kh,kw], up_x, up_y,
 * down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 *   -- models/networks/stylegan2_op/upfirdn2d.cpp:4-23,
 *      upfirdn2d_kernel.cu:52-137 (kernel), :140-272 (dispatch).
 * y must hold major*out_h*out_w*minor floats with
 *   out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y  (.cu:166).
 * minor=1 is the reference's NCHW use; minor=C, major=B is NHWC.
 * Any kh,kw <= 8 is supported (the reference silently launches nothing for
 * combinations outside its 6 modes, .cu:172-223).  The backward of the op is
 * the same entry with the flipped kernel and g_pad (upfirdn2d.py:116-121). */
int ppst_upfirdn2d(const void* x, const void* k, void* y,
                   int major, int in_h, int in_w, int minor, int kh, int kw,
                   int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                   int dtype, void* stream);

/* Fused-path Blur on NHWC activations (Blur inside ConvLayer(downsample=True),
 * stylegan2_layers.py:142-164,513-520): square FIR (ksize 3 or 4), pad (pad0 before,
 * pad1 after) with zero or reflection padding folded in (the reference runs
 * nn.ReflectionPad2d first, :151-159); down=2 keeps every second sample (all that a
 * following stride-2 1x1 conv reads); s2d=1 writes the output space-to-depth
 * y[B][ceil(oh/2)][ceil(ow/2)][4*C] (phase (oy&1)*2+(ox&1) major) for the stride-2
 * 3x3 conv that ppst_conv2d_mfma runs as a stride-1 conv over that layout (the padded
 * extent is written in full: zeros beyond (oh, ow), so y needs no initialisation).
 * in_scale_shift (optional, [B][C][2] (a, s)) + in_act (PPST_ACT_NONE / PPST_ACT_LRELU):
 * the input is read as in_act(a*x + s) -- the InstanceNorm + FusedLeakyReLU that
 * ConvLayer(norm='in') puts in front of the Blur (:542-549) without a pass of its own;
 * zero-padding positions stay 0. */
int ppst_blur_nhwc(const void* x, const void* k, void* y, int B, int in_h, int in_w, int C,
                   int ksize, int pad0, int pad1, int pad_mode, int down, int s2d,
                   const void* in_scale_shift, int in_act, void* stream);
int ppst_blur_nhwc_st(const void* x, const void* k, void* y, int B, int in_h, int in_w, int C,
                      int ksize, int pad0, int pad1, int pad_mode, int down, int s2d,
                      const void* in_scale_shift, int in_act, int st /* of x and y */, void* stream);

/* fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *   -- stylegan2_op/fused_bias_act.cpp:4-20, fused_bias_act_kernel.cu:19-49.
 * y[i] = f(x[i] + b[(i/step_b) % size_b]) * scale; b==NULL => no bias,
 * ref==NULL => no reference.  act: 1 linear, 3 leaky-relu; grad: 0 forward,
 * 1 first derivative gated by sign(ref), 2 second derivative (= 0). */
int ppst_fused_bias_act(const void* x, const void* b, const void* ref, void* y,
                        int64_t n, int step_b, int size_b, int act, int grad,
                        float alpha, float scale, int dtype, void* stream);

/* -------------------------------------------------------------- layout --- */
int ppst_nchw_to_nhwc(const void* x, void* y, int B, int C, int H, int W, void* stream);
int ppst_nhwc_to_nchw(const void* x, void* y, int B, int C, int H, int W, void* stream);

/* ---------------------------------------------------- fused conv (MFMA) ---
 * Implicit-GEMM convolution on NHWC fp32 activations with bf16 MFMA
 * (v_mfma_f32_16x16x32_bf16).  precision 0: every fp32 operand is split into
 * hi+lo bf16 and 3 MFMAs (hi*hi, hi*lo, lo*hi) accumulate in fp32 ("bf16x3",
 * fp32-class, relative error ~1e-5); precision 1: single bf16 pass.
 * Replaces F.conv2d / F.conv_transpose2d inside EqualConv2d
 * (stylegan2_layers.py:184-193), EqualizedConv2d (:305-347, incl. the fused
 * 4x4 stride-2 transposed-conv upscale :312-321) and nn.Conv2d of the
 * generator heads (generator.py:174-238), with the StyledConv epilogue
 * (stylegan2_layers.py:467-475) fused in.
 *
 * The convolution is described by a packed-weight blob + step table built by
 * ppst_conv_pack (one-time per weight tensor). */

typedef struct ppst_conv_step {
  int32_t chan_off;   /* first input channel (in the in_ld-strided pixel) of this 32-ch chunk */
  int32_t dy, dx;     /* tap offset relative to the output pixel, each in [-1,1] */
  int32_t new_chunk;  /* 1 if this step starts a new input chunk (LDS restage) */
} ppst_conv_step;

/* Pack one weight tensor for ppst_conv2d_mfma (one-time per weight tensor).
 *  w        fp32 weights; element (n, c, ky, kx) at w[n*sn + c*sc + ky*sy + kx*sx]
 *  scale    multiplied in (EqualConv2d runtime scale, stylegan2_layers.py:177)
 *  src_*    DEVICE int32 arrays [n_groups*nsteps]: the K-slice of step s of group g is
 *           w[n][src_c .. src_c+31][src_ky][src_kx]
 *  bn       64, 128 or 256 = N tile of the kernel variant that will consume the blob
 *  out      n_groups * ceil(cout/bn) * nsteps * (precision==0 ? 8 : 4) * bn * 8 bf16 */
/* The same for ppst_conv_args.dual_b: bn = 256 = two column phases x 128 channels of N tile t (channels 128 t ..); src_kx[i] holds
 * (kx of phase 0) | (kx of phase 1) << 8.  out: n_groups * ceil(cout/128) * nsteps * (precision==0 ? 8 : 4) * 256 * 8 bf16. */
/* Weights for ppst_conv_args.variant 11 (the fused upscale of stylegan2_layers.py:312-321 as the UN-BLURRED 3x3 stride-2
 * transposed conv -- nine products per input pixel instead of the sixteen of the 4x4 kernel -- with the 2x2 box sum of the weight
 * blur applied to the OUTPUT in the kernel's epilogue): w = the layer's 3x3 weight (Cout, Cin, 3, 3) with element strides
 * sn / sc / sy / sx, scaled by ``scale``; cout % 64 == 0, cin % 32 == 0; out: ppst_conv_pack_up9_bytes(cout, cin) bytes. */
int64_t ppst_conv_pack_up9_bytes(int cout, int cin);
int ppst_conv_pack_up9(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int cin,
                       void* out, void* stream);
/* Weights for ppst_conv_args.k64: as ppst_conv_pack (dual = 0) / ppst_conv_pack_dual (dual = 1) with precision 1 / 3, but step i of
 * the table covers channels src_c[i] .. src_c[i] + 63; out: n_groups * n_tiles * nsteps * 8 * bn * 8 elements of the operand type. */
int ppst_conv_pack_k64(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int bn,
                       const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                       int precision, int dual, void* out, void* stream);
int ppst_conv_pack_dual(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout,
                        const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                        int precision, void* out, void* stream);
int ppst_conv_pack(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx,
                   float scale, int cout, int bn,
                   const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx,
                   int nsteps, int n_groups, int precision, void* out, void* stream);
/* Batched forms (training: every plan of a network is re-packed after each Adam step -- one launch instead of one per plan).
 * ``jobs`` is a DEVICE array; the caller fills block0 / nblocks: consecutive block ranges, nblocks = ppst_pack_job_blocks(total),
 * total_blocks = their sum.  Field meaning as the arguments of ppst_conv_pack / ppst_upscale_weight. */
typedef struct ppst_pack_job {
  const void* w;
  int64_t sn, sc, sy, sx;
  const int32_t* src_c;
  const int32_t* src_ky;
  const int32_t* src_kx;
  void* out;
  int64_t total;              /* n_groups * ceil(cout / bn) * nsteps * 4 * bn */
  int64_t block0;
  float scale;
  int32_t cout, bn, nsteps, n_groups, x3, f16, nblocks;   /* x3: precision 0 (hi + lo planes); f16: precision 3 / 4 */
  int32_t dual;               /* 1: ppst_conv_pack_dual semantics (the N tile holds two output column phases) */
                              /* (x3 = 2 with f16 = 0 / 1: ppst_conv_pack_k64 semantics -- the lo planes hold channels 32-63 of a 64-channel step) */
} ppst_pack_job;
typedef struct ppst_upscale_job {
  const void* w;
  void* out;
  int64_t total, block0;      /* total = cin * cout * 16 */
  float scale;
  int32_t cout, cin, nblocks;
} ppst_upscale_job;
int ppst_pack_job_blocks(int64_t total);
int ppst_conv_pack_batch(const void* jobs, int njobs, int total_blocks, void* stream);
int ppst_upscale_weight_batch(const void* jobs, int njobs, int total_blocks, void* stream);

/* EqualizedConv2d fused-upscale weight (stylegan2_layers.py:314-319):
 * w (Cout,Cin,3,3)*scale -> out (Cin,Cout,4,4), the F.conv_transpose2d operand */
int ppst_upscale_weight(const void* w, void* out, int cout, int cin, float scale, void* stream);
/* Round 5: the input gradient of the stride-2 3x3 conv (stylegan2_layers.py:497-555, ConvLayer(downsample=True)) as ONE stride-1 conv
 * with 2 x 2 taps whose 4 x cin output channels stack the four output phases -- out (4*cin, cout, 2, 2) from the forward weight w
 * (cout, cin, 3, 3): out[(py*2+px)*cin + n][c][ty][tx] = w[c][n][ky][kx], a phase p using tap offset 0 with k = (p == 0 ? 0 : 1) and
 * offset 1 (one position back) with k = 2 when p == 0; zero elsewhere.  The stacked output goes through ppst_depth_to_space_st.  For
 * thin layers (cin 32 / 64) the four-group scattered form ran 2 312-5 780 four-wave blocks that each staged the same tile. */
int ppst_dgrad_s2d_stack_weight(const void* w, void* out, int cout, int cin, void* stream);
/* x [B][th][tw][4 C] (channel block py*2+px = output phase) -> y [B][oh][ow][C], y[b][2q+py][2p+px][c] = x[b][q][p][(py*2+px)*C + c];
 * oh <= 2 th, ow <= 2 tw; st = PPST_ST_* of both tensors (C % 4 == 0 fp32, % 8 half; 16-byte aligned). */
int ppst_depth_to_space_st(const void* x, void* y, int B, int th, int tw, int oh, int ow, int C, int st, void* stream);

typedef struct ppst_conv_args {
  const void* x;        /* NHWC fp32 input, pixel stride in_ld floats */
  const void* wpack;    /* from ppst_conv_pack */
  const void* steps;    /* device array ppst_conv_step[n_groups*nsteps + 4]: 4 padding entries at the end (the kernel
                           prefetches the descriptor of step s+3 without a bounds test; their content is ignored) */
  void* y;              /* NHWC fp32 output, pixel stride out_ld floats; one image (out_h * out_w * out_ld, and the same with
                           res_ld) must stay below 2^31 elements: the epilogues address inside an image with 32-bit offsets */
  const void* bias;     /* [cout] or NULL (sum of all per-channel biases) */
  const void* noise;    /* [B][out_h][out_w] or NULL (NoiseInjection, stylegan2_layers.py:376-399) */
  const void* prelu;    /* [1] PReLU slope (device) when act == PPST_ACT_PRELU */
  void* stats;          /* [B][tiles_per_image][cout][2] per-tile (sum, sumsq) of the stored output, or NULL */
  const void* residual; /* NHWC fp32 [B][out_h][out_w][res_ld] added before act, or NULL */
  float noise_weight;
  float out_scale;      /* multiplies the activated output (1.0 default) */
  int32_t B, in_h, in_w, in_ld;
  int32_t out_h, out_w, out_ld, cout;
  int32_t nsteps, n_groups;      /* n_groups: 1, or 4 output phases (transposed conv) */
  int32_t pad_mode;              /* PPST_PAD_* applied to out-of-image taps */
  int32_t in_off_y, in_off_x;    /* input pixel = tile pixel + tap + in_off (e.g. -pad) */
  int32_t out_sy, out_sx;        /* output pixel stride (2 for the transposed conv) ... */
  int32_t act;                   /* PPST_ACT_* | 0x100: residual joins AFTER the activation */
  int32_t precision;             /* 0 bf16x3 (fp32-class), 1 bf16 single pass, 3 fp16 single pass, 4 fp16 two-pass
                                    (activation hi + lo, weight fp16); ppst_conv_pack with the same value */
  int32_t res_ld;
  int32_t tile_h, tile_w;        /* logical (pre-scatter) output extent tiled by 16x16 */
  int32_t halo;                  /* 0: every tap is (0,0) (1x1 conv); 1: taps in [-1,1]^2 */
  int32_t bn;                    /* N tile the weights were packed for (64 or 128) */
  const void* in_scale_shift;    /* optional [B][in_c][2] (a, s): the input is read as in_act(a*x + s) --
                                    "normalise on load" of the producer's InstanceNorm/StyleMod/activation; padding
                                    positions stay 0 (they pad the normalised tensor).  NULL: input used as is */
  const void* in_prelu;          /* [1] slope when in_act == PPST_ACT_PRELU */
  int32_t in_c, in_act;          /* channel count of the in_scale_shift table; PPST_ACT_* */
  int32_t flop_steps;            /* steps that carry real weights (profiling only; 0 = nsteps) */
  int32_t tile_rows;             /* (32 only with variant 7, 24 only with variant 9) 16: 16x16-pixel tiles, 512- (bn 128) / 256-thread blocks, 1 block per CU;
                                    8:  8x16-pixel tiles, 256-thread blocks, 2 blocks per CU, a two-slot activation ring: variant 0,
                                    bn 128, halo 1, precision 0 and the early_a promise (chunks of >= 2 steps).  Bit-identical
                                    outputs; for launches whose 16-row grid would under-fill the chip */
  int32_t a_slots;               /* depth of the activation-tile ring in LDS: 0 = default (3).  1 or 2 may be given when
                                    NO group of the step table has more chunks (steps with new_chunk = 1) than that:
                                    with bn = 64 the smaller footprint lets two blocks share a CU (small-K layers are
                                    latency/HBM-bound with one).  The caller owns this promise: the table is on the
                                    device and is not re-read by the host. */
  int32_t early_a;               /* 1: every chunk of the step table spans >= 2 steps AND steps[i].w carries, besides bit 0
                                    (step i opens a chunk), bit 1 = step i+1 opens a chunk and bits 8.. = that chunk's
                                    channel offset: the kernel then requests a chunk's activations one step earlier
                                    (HBM latency no longer stalls the staging store).  0: bits 1.. are ignored. */
  int32_t variant;               /* 0: the 8-wave kernel (512 threads, wave tile 64 px x 64 ch; bn 64 / 128).
                                    1: the fat-wave kernel (conv_mfma2.hip: 256 threads = one wave per SIMD, wave tile
                                    128 px x 64 / 128 ch; bn 128 / 256; precision 0 only).  Its activation ring has two
                                    slots: every chunk of the step table must span >= 2 steps (the early_a promise).
                                    2: 8 waves x (128 px x 64 ch), bn = 256 (the production kernel for Cout % 256 == 0);
                                    3: two 4-wave blocks per CU, bn = 128, one activation slot (experiment) -- both with
                                    the early_a promise, precision 0.
                                    4 / 5 / 6 (conv1x1.hip, precision 0, one group, unit output stride, no in_off):
                                    4: all taps (0,0) (halo 0), bn = 64, in and out extents equal;
                                    5: any taps in [-1,1]^2, bn = 64;
                                    6: plain 3x3 stride-1 tables only -- nsteps = 9 * chunks and step 9c + 3(dy+1) + (dx+1)
                                       is tap (dy, dx) of chunk c; bn = 64, or 128 for Cout in 65..128.  The table lives on
                                       the device and is not re-read: the CALLER owns this promise (as with a_slots).
                                    7: conv_mfma2.hip with 8 waves as 4 (M) x 2 (N): block tile 32 x 16 px x 128 ch, one
                                       activation slot; bn = 128, tile_rows = 32, the early_a promise (precision 0: experiment, measured
                                       slower; precision 1 / 3: a production form with TWO activation slots -- the Cout = 128-class
                                       layers of the single-pass modes).
                                    8: conv_ksplit.hip -- bn = 128, the block's 8 waves as 2 (K) x 2 (M) x 2 (N): wave group k
                                       takes the steps of parity k (wave tile 128 px x 64 ch), the two partial sums meet in
                                       LDS before the epilogue.  Needs the early_a promise, precision 0, unit output stride
                                       and steps[i].w bit 2 = parity of the chunk step i belongs to (the activation slot).
                                       NOT bit-identical to the others (the K sum is split in two: <= 1.1e-6 relative);
                                       experiment, measured 3-7 % slower than variant 0.
                                    10: conv_wino.hip -- plain 3x3 stride-1 tables only (the variant-6 promise: nsteps = 9 * chunks,
                                       steps[9c].x = first channel of chunk c), wpack from ppst_conv_pack_wino, bn = 128,
                                       tile_rows = 16, one group, unit strides, precision 0.  Winograd F(2,3) along x: fp32-class
                                       like the others (<= 3e-5 against float64) but NOT bit-identical to them.
                                   11: conv_mfma2.hip "UP9" -- the fused 4x4 stride-2 upscale computed as the un-blurred 3x3 transposed
                                       conv (u types ee / eo / oe / oo of 4 / 2 / 2 / 1 taps: 9 products per input pixel instead of
                                       16) whose 2x2 box sum runs in the epilogue; bn = 256 = 4 N-waves x 4 types x 16 channels,
                                       n_groups 1, tile_rows 15 (blocks of 15 x 15 input positions: the 16 x 16 grid of a block
                                       feeds its neighbours' row / column), halo 1, zero padding, out_sy = out_sx = 2, precision
                                       0, no normalise-on-load / residual / PReLU; wpack from ppst_conv_pack_up9, steps = per
                                       32-channel chunk the shifts (0,0), (-1,0), (0,-1), (-1,-1).  fp32-class (<= 3e-5 against
                                       float64), NOT bit-identical to the 4x4 forms (other summation order).
                                    9: conv_mfma2.hip with 6 m-tiles per wave: block tile 24 x 16 px x 128 ch (wave tile 96 px x 64 ch),
                                       TWO activation slots; bn = 128, tile_rows = 24, the early_a promise, precision 0
                                       (experiment: bit-identical, +-1.5 % of variant 0 -- the 36-step tiles of the Cout = 128
                                       layers are prologue / epilogue bound, not wave-tile bound).
                                       With k64 (precision 1 / 3, io_st != 0) a production form: the Cout = 128-class layers of the
                                       single-pass modes on half-stored activations.
                                    Variants 0-7 and 9 give bit-identical outputs; the per-tile statistics differ in the last
                                    bit between variants (other summation tree).  The library returns PPST_EINVAL for a
                                    variant whose shape conditions do not hold. */
  int32_t in_presplit;           /* experiment (PPST_EXPERIMENTS builds): x is pre-split -- per pixel and 8-channel group 32 bytes
                                    [hi x 8 | lo x 8] bf16 (ppst_presplit), same pixel stride in_ld -- and the activation tile is
                                    staged by LDS-DMA.  variant 0, bn 128, halo 1, precision 0, no in_scale_shift, every chunk of
                                    the step table >= 4 steps (the caller's promise, like early_a). */
  int32_t dual_b;                /* 1 (variant 2, bn 256, n_groups 2, halo 1, precision 0 / 1 / 3, out_sy = out_sx = 2): the fused 4x4 stride-2
                                    upscale (stylegan2_layers.py:312-321) with Cout % 128 == 0 as TWO row phases whose N tile of 256
                                    is [column phase 0: 128 channels | column phase 1: 128 channels] -- wpack from
                                    ppst_conv_pack_dual, steps[i].dx = (dx of phase 0 + 1) | (dx of phase 1 + 1) << 8.  The
                                    activation tile is staged once for two phases and the layer runs on the 128 x 64 wave
                                    tiles; outputs bit-identical to the four-group form, stats [B][4 * tiles][cout][2] as there. */
  int32_t io_st;                 /* storage type of x, residual and y (PPST_ST_*): 0 fp32; PPST_ST_F16 with precision 3 / PPST_ST_BF16
                                    with precision 1 only -- "half-precision activation storage": the generator / encoder activations
                                    of the fp16 and bf16 modes (BASELINE configs[4] / [3]) live in HBM in the operand type of the mode.
                                    Accumulation, bias / noise / activation, the instance-norm statistics and (a, s) of
                                    normalise-on-load stay fp32; the output is rounded once, to nearest even, at the store.
                                    variant 0 (tile_rows 16), 2, 4, 5, 6; pointers 8-byte aligned; in_ld / out_ld / res_ld in ELEMENTS. */
  int32_t k64;                   /* 1 (io_st != 0, precision 1 / 3, halo 1, variant 2 [also with dual_b] or variant 9 with tile_rows 24):
                                    a step of the table covers SIXTY-FOUR input channels (steps[i].chan .. + 63; wpack from
                                    ppst_conv_pack_k64: channels 0-31 of the chunk where the fp32-class blob keeps its hi planes,
                                    32-63 where it keeps its lo planes) -- half the steps per MFMA of the 32-channel single-pass
                                    form.  Outputs equal that form's up to fp32 summation order (the two halves of a chunk are
                                    accumulated alternately instead of chunk after chunk). */
  const void* in_res;            /* round 5, variant 4 with in_scale_shift only: a second input tensor of the conv's own extent and
                                    channel count, added BEFORE in_act -- the input is read as in_act(a*x + s + in_res): the resnet
                                    merge of generator.py:28-31 (prelu(IN(conv2) + x)) applied while the 1x1 conv that is its only
                                    consumer loads its fragments, so the merged tensor is never written (layert1 -> layert1.1).
                                    Same fp32 operations in the same order as ppst_affine_act(res_before_act) followed by the plain
                                    conv (the test holds the pair to 2e-6 of each other).  fp32 storage, precision 0; NULL: none */
  int32_t in_res_ld;             /* pixel stride of in_res in elements */
  int32_t ksplit;                /* round 5: 0 / 1 none; S = 2, 4 or 8 (variant 0, 2 or 10; nsteps % S == 0 and -- the CALLER's promise --
                                    step i * nsteps / S opens a chunk for every i): the reduction is split over S blocks per output
                                    tile (grid y), each running nsteps / S steps of the table; the first S - 1 leave their raw
                                    accumulators in a scratch buffer of the library and raise a flag, the last block of the tile
                                    (dispatched behind them) adds them in a fixed order and runs the epilogue.  For launches whose
                                    grid fills a fraction of the chip and whose blocks are one long serial chain of steps (the
                                    64^2 ... 4^2 layers of a train step at batch 2: 4-128 blocks, 72-160 steps): S x the blocks,
                                    1 / S of the chain.  Results equal the unsplit launch's up to fp32 summation order.  At most
                                    256 (S - 1) x tiles per launch; launches that use it are serialised per stream by the library
                                    (one scratch buffer per stream that uses it, sixteen (device, stream) pairs per process). */
  const int32_t* ksplit_starts;  /* HOST pointer to ksplit + 1 ascending step indices, starts[0] = 0, starts[ksplit] = nsteps, every one of
                                    them a step that opens a chunk (the caller's promise): block row i runs steps
                                    [starts[i], starts[i + 1]) -- tables whose chunks differ in length (the stride-2 conv on a
                                    space-to-depth tensor: 4 / 2 / 2 / 2 taps per phase).  NULL: equal shares of nsteps / ksplit
                                    steps.  Read during the call, not kept. */
} ppst_conv_args;

int ppst_conv2d_mfma(const ppst_conv_args* a, void* stream);
/* diagnostic of ppst_conv_args.ksplit: 1 if a block of a K-split launch on `stream` ever gave up waiting for its partner blocks (that
 * launch's output is wrong; cannot happen while blocks are dispatched in grid order), 0 if none did, < 0 if the stream never ran a
 * K-split launch.  Synchronises the stream and resets the marker. */
int ppst_conv_ksplit_check(void* stream);
/* Weights of a plain 3x3 stride-1 conv for ppst_conv_args.variant 10 (conv_wino.hip: Winograd F(2,3) along x, direct along y --
 * 12 K-steps per 32-channel chunk and pixel PAIR instead of 9 per pixel, 1.5x fewer MFMAs; EqualConv2d / StyledConv conv,
 * stylegan2_layers.py:184-193, 275-348, 439-475).  Element (n, c, ky, kx) of the kernel at w[n*sn + c*sc + ky*sy + kx*sx] (any
 * strides: the input-gradient plan passes the transposed, flipped view of the same tensor), scaled by ``scale``; the row
 * transform U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) is formed in double and stored as bf16 hi + lo in the register-fragment
 * order of the kernel's waves: [Cout/128][4 positions][2 channel halves][Cin/32][3 ky][4 n-tiles][hi|lo][64 lanes][8 k].
 * cin % 32 == 0; ``out`` holds ppst_conv_pack_wino_bytes(cout, cin) bytes. */
int64_t ppst_conv_pack_wino_bytes(int cout, int cin);
int ppst_conv_pack_wino(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int cin,
                        void* out, void* stream);
/* fp32 NHWC [npix][x_ld] -> the pre-split layout of ppst_conv_args.in_presplit in y [npix][y_ld] (C % 8 == 0) */
int ppst_presplit(const void* x, void* y, int64_t npix, int C, int x_ld, int y_ld, void* stream);
/* 1 when the library was built with PPST_EXPERIMENTS=1: the measured-and-off forms (variants 1 / 3 / 7 / 8 / 9, precision 4,
 * in_presplit) are then compiled in; the production build returns PPST_EINVAL for them. */
int ppst_has_experiments(void);
/* Exact-fp32 twin of ppst_conv2d_mfma (v_mfma_f32_32x32x2_f32; a->wpack, bn, precision, a_slots, early_a are ignored):
 * same step table / padding / epilogue semantics, weights read from the fp32 tensor itself -- element (n, c, ky, kx) at
 * w[n*sn + c*sc + ky*sy + kx*sx], step s of group g uses w[n][src_c .. src_c+31][src_ky][src_kx] * wscale (src_c < 0:
 * zero-weight step).  A verification path (10-30x slower): it tells rounding of the bf16 hi+lo split apart from defects. */
int ppst_conv2d_f32(const ppst_conv_args* a, const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float wscale,
                    const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, void* stream);
/* number of tile_rows x 16 tiles per image for (tile_h, tile_w) -- size of the stats buffer */
int ppst_conv_tiles(int tile_h, int tile_w, int tile_rows);

/* FromRGB-type conv: 1x1, Cin <= 4 (HBM-bound), NHWC in (in_ld) -> NHWC out, fused
 * bias + leaky relu (ConvLayer(3, C, 1), stylegan2_layers.py:497-555). */
int ppst_conv1x1_small_cin(const void* x, const void* w, const void* bias, void* y,
                           int64_t npix, int cin, int in_ld, int cout, float wscale,
                           int act, void* stream);
int ppst_conv1x1_small_cin_st(const void* x, const void* w, const void* bias, void* y,
                              int64_t npix, int cin, int in_ld, int cout, float wscale,
                              int act, int y_st /* x stays fp32: the image */, void* stream);
/* ToRGB-type conv: 1x1, Cout <= 4 (stylegan2_layers.py:487-489); y NHWC [npix][cout] */
int ppst_conv1x1_small_cout(const void* x, const void* w, const void* bias, void* y,
                            int64_t npix, int cin, int cout, float wscale, void* stream);
int ppst_conv1x1_small_cout_st(const void* x, const void* w, const void* bias, void* y,
                               int64_t npix, int cin, int cout, float wscale, int x_st /* y stays fp32 */, void* stream);
/* ToRGB's 1x1 conv (stylegan2_layers.py:477-495, Cout = 3) with the merge pass of the block in front of it applied ON LOAD (round 5):
 * the input is read as (a[b][c] * x + s[b][c] + bilinear_x2(res)) * out_scale -- (IN + StyleMod of conv2 + the x2-upsampled skip) /
 * sqrt2 of the last UpsamplingResnetBlock (generator.py:63-78), whose only consumer in the image pass is this conv.
 * x [B][H][W][cin] dense (storage type x_st), scale_shift [B][cin][2], res [B][H/2][W/2][res_ld] (type x_st) or NULL,
 * w [3][cin], bias [3] or NULL -> y [B][H][W][3] fp32.  Equal to ppst_affine_act (res_up2) followed by ppst_conv1x1_small_cout. */
int ppst_torgb_apply_st(const void* x, const void* scale_shift, const void* res, int res_ld, float out_scale, const void* w,
                        const void* bias, void* y, int B, int H, int W, int cin, float wscale, int x_st, void* stream);

/* ------------------------------------------- instance norm / style mod ---
 * nn.InstanceNorm2d (eps 1e-5, biased var) + StyleMod (stylegan2_layers.py:361-374,
 * :414-437).  Statistics come either from the conv epilogue partials or from
 * ppst_in_stats (optionally weighting the border as if the tensor had been
 * ReplicationPad2d(1)-padded first: generator.py:175-176). */
int ppst_in_stats(const void* x, void* partial, int B, int H, int W, int C, int ld,
                  int rep_pad, int* n_partials, void* stream);
/* reduce partials -> per (b,c) scale a and shift s with  y = a*x + s:
 *   a = rstd * (style0+1), s = style1 - mean*a   (style==NULL: a=rstd, s=-mean*rstd)
 * style: [B][style_ld] rows whose first 2*C entries are the StyleMod linear output.
 * count = #elements per (b,c). */
int ppst_in_finalize(const void* partial, int n_partials, const void* style,
                     int style_ld /* row stride of style (>= 2C): slices of a batched StyleMod GEMV */,
                     const void* post_bias /* [C] added to the shift, or NULL */,
                     void* scale_shift /* [B][C][2] */, int B, int C, double count,
                     float eps, void* stream);
/* y = act(a*x + s [+ res]) * out_scale ; (skip + res)/sqrt2 of the resnet blocks
 * (generator.py:47-78) is res + out_scale. act as PPST_ACT_*; PReLU slope ptr. */
int ppst_affine_act(const void* x, const void* scale_shift, const void* res,
                    const void* res_scale_shift /* optional affine of res */, void* y,
                    int B, int64_t hw, int C, int x_ld, int res_ld, int y_ld,
                    int act /* | 0x100: res joins before act */, const void* prelu,
                    float out_scale,
                    int res_up2_w /* >0: res is a HALF-resolution tensor, upsampled x2 bilinearly on the
                                     fly (resnet skip, generator.py:75); value = output width W */,
                    void* stream);
/* x_st: storage of x and res; y_st: of y (either may be fp32 beside a half type; two different half types are refused) */
int ppst_affine_act_st(const void* x, const void* scale_shift, const void* res, const void* res_scale_shift, void* y,
                       int B, int64_t hw, int C, int x_ld, int res_ld, int y_ld, int act, const void* prelu,
                       float out_scale, int res_up2_w, int x_st, int y_st, void* stream);
/* ppst_affine_act that also emits the instance-norm partials of its OUTPUT (the layout
 * ppst_in_stats produces for (H, W); rep_pad as there), so the norm that follows needs no
 * read pass of its own.  C % 4 == 0. */
int ppst_affine_act_stats(const void* x, const void* scale_shift, const void* res,
                          const void* res_scale_shift, void* y, void* partial,
                          int B, int H, int W, int C, int x_ld, int res_ld, int y_ld,
                          int act, const void* prelu, float out_scale, int rep_pad,
                          int res_up2 /* as res_up2_w, boolean */, void* stream);
/* nearest x2 upsample NHWC (Upscale2d, stylegan2_layers.py:86-97) */
int ppst_upsample_nearest2(const void* x, void* y, int B, int H, int W, int C, void* stream);
int ppst_upsample_nearest2_st(const void* x, void* y, int B, int H, int W, int C, int st, void* stream); /* C % 8 == 0 for half */

/* ------------------------------------------------- pooling / resizing ---- */
/* GAP + GMP over HxW per (b,c): out [B][2C] = cat(mean, max)  (encoder_col.py:159-161).
 * mask: optional [B][H][W] multiplier (mask channel i of encoder_col.py:176-178). */
int64_t ppst_gap_gmp_ws(int B, int64_t hw, int C); /* workspace bytes */
int ppst_gap_gmp(const void* x, const void* mask, void* out, void* ws, int B, int H, int W,
                 int C, int ld, void* stream);
int ppst_gap_gmp_st(const void* x, const void* mask, void* out, void* ws, int B, int H, int W,
                    int C, int ld, int x_st, void* stream);
/* integer-factor average pool (adaptive_avg_pool2d to H/f) NHWC -> dst slice */
int ppst_avgpool(const void* x, void* y, int B, int H, int W, int C, int x_ld, int f,
                 int y_ld, void* stream);
/* bilinear resize, align_corners=False (F.interpolate; generator.py:75,274-277,
 * encoder_col.py:129) NHWC -> dst slice with pixel stride y_ld */
int ppst_bilinear(const void* x, void* y, int B, int H, int W, int C, int x_ld,
                  int OH, int OW, int y_ld, void* stream);
/* Tail of a correspondence feature head (generator.py:174-238, the layer128 / layer256 heads): with
 * f = act(a*x + s) the activated output of the head's last conv, writes feat = PxP average pool of f
 * ([B][H/P][W/P], pixel stride feat_ld) and feat1 = F.interpolate(f, (H/D, W/D), bilinear) for the exact
 * factors D = 1 (f itself) and D = 2 (2x2 mean) in one read of x -- f is never stored at full size. */
int ppst_head_tail(const void* x, const void* scale_shift, const void* prelu, void* feat, void* feat1,
                   int B, int H, int W, int C, int x_ld, int feat_ld, int feat1_ld, int P, int D, int act,
                   void* stream);
/* 2x2 max pool on masks (encoder_col.py:218) NHWC */
int ppst_maxpool2(const void* x, void* y, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------- linear ----
 * y[b][n] = act( sum_k f(x[b][k]) * w[n][k] * wscale + bias[n]*bscale )
 * (EqualLinear stylegan2_layers.py:222-242, EqualizedLinear :268-273,
 * nn.Linear of the E2 projectors encoder_col.py:52-88).  relu_in applies ReLU
 * to x first (the projectors' leading nn.ReLU). act: PPST_ACT_NONE/LRELU. */
int ppst_linear(const void* x, const void* w, const void* bias, void* y,
                int B, int K, int N, float wscale, float bscale, int relu_in, int act,
                void* stream);
/* rows: y = x * rsqrt(sum x^2 + eps)  (util.normalize, util/util.py:18-22; mode 0)
 *       y = x / max(||x||, eps)        (F.normalize; mode 1) */
int ppst_l2norm_rows(const void* x, void* y, int B, int K, float eps, int mode, void* stream);
/* y = a*(1-r) + b*r (util.lerp, util/util.py:32-35) */
int ppst_lerp(const void* a, const void* b, void* y, int64_t n, float r, void* stream);
/* GeneratorModulation (generator.py:80-91): y[b,p,c] = x[b,p,c]*scale[b,c] + bias[b,c] */
int ppst_spatial_modulation(const void* x, const void* scale, const void* bias, void* y,
                            int B, int64_t hw, int C, void* stream);
int ppst_spatial_modulation_st(const void* x, const void* scale, const void* bias, void* y,
                               int B, int64_t hw, int C, int y_st /* x (the spatial code) stays fp32 */, void* stream);

/* ------------------------------------------------------ correspondence --- */
/* PPSTModel.Rselfcorr (ppst_model.py:330-339): fea NHWC [B][H][W][C] ->
 * out NHWC slice [B][H/4][W/4][256] at pixel stride out_ld. */
int ppst_rselfcorr(const void* fea, void* out, int B, int H, int W, int C, int out_ld,
                   void* stream);
/* corrm feature prep (ppst_model.py:349-361): per pixel, mean-centre the first
 * `ncenter` channels, then L2-normalise all C channels (+eps). [B][P][C] rows. */
int ppst_corr_prep(const void* fea, void* out, int B, int P, int C, int ncenter, void* stream);
/* corrm with opt.match_kernel = k != 1 (ppst_model.py:345-347): F.unfold(fea, k, padding = k / 2) of an NHWC map
 * [B][H][W][C] written as rows [B][H*W][C*k*k], column c*k*k + ky*k + kx (F.unfold's order), zeros outside; k odd.
 * _bwd: dx [B][H][W][C] from the gradient of those rows.  (ppst_corr_prep takes rows of any length.) */
int ppst_unfold_rows(const void* x, void* out, int B, int H, int W, int C, int k, void* stream);
int ppst_unfold_rows_bwd(const void* g, void* dx, int B, int H, int W, int C, int k, void* stream);
/* fp32 MFMA GEMM, C[b] = alpha * A[b] (MxK, row-major) * B[b]^T (NxK row-major) */
int ppst_gemm_nt_f32(const void* A, const void* Bm, void* C, int batch, int M, int N, int K,
                     float alpha, void* stream);
/* fp32 MFMA GEMM, C[b] = A[b] (MxK row-major) * B[b] (KxN row-major, ldb) -> ldc */
int ppst_gemm_nn_f32(const void* A, const void* Bm, void* C, int batch, int M, int N, int K,
                     int ldb, int ldc, void* stream);
/* The same two products on the bf16 matrix pipe (torch.matmul of ppst_model.py:363 / :377 / :385, fp32 in and out).
 * passes = 6: three bf16 planes per operand (all 24 mantissa bits), fp32-class results -- the cosine logits the softmax
 *             multiplies by 1 / T = 100; K % 16 == 0.
 * passes = 3: two planes (the convs' bf16x3, ~2^-16 relative per product) -- softmax rows times feature / gradient
 *             matrices; K % 32 == 0.
 * A, Bm 16-byte aligned; other values of passes: PPST_EINVAL. */
int ppst_gemm_nt_split(const void* A, const void* Bm, void* C, int batch, int M, int N, int K,
                       float alpha, int passes, void* stream);
int ppst_gemm_nn_split(const void* A, const void* Bm, void* C, int batch, int M, int N, int K,
                       int ldb, int ldc, int passes, void* stream);
/* in-place row softmax of x/div (F.softmax(matmul/0.01, dim=-1), ppst_model.py:363) */
int ppst_softmax_rows(void* x, int64_t rows, int cols, float div, void* stream);
/* PPSTModel.warp unfold/fold plumbing (ppst_model.py:366-387): NCHW image
 * [B][C][H][W] <-> patch rows [B][P][C*s*s] */
int ppst_unfold_patches(const void* x, void* y, int B, int C, int H, int W, int s, void* stream);
int ppst_fold_patches(const void* x, void* y, int B, int C, int H, int W, int s, void* stream);

/* -------------------------------------------------------- pre-process ---- */
/* One pass of Pillow's 8-bit resample (Image.resize(..., BICUBIC): data/base_dataset.py:141-168
 * __make_power_2 / __scale_shortside) over interleaved uint8 images x [B][in_h][in_w][C]:
 * horizontal != 0 resizes the width to out_size, else the height.  bounds int32 [out_size][2] =
 * (first input sample, count), coef int32 [out_size][ksize] = 22-bit fixed-point weights (host
 * logic: Pillow's precompute_coeffs + normalize_coeffs_8bpc, ppst_amd/imageio.py).
 * y = clip8((2^21 + sum in*coef) >> 22): integer arithmetic, bit-exact.  Pillow runs the
 * horizontal pass first, then the vertical one, each rounding to uint8. */
int ppst_resample_u8(const void* x, void* y, int B, int in_h, int in_w, int C, int out_size, int horizontal,
                     const void* bounds, const void* coef, int ksize, void* stream);
/* transforms.ToTensor + transforms.Normalize(mean, std) (base_dataset.py:133-138): uint8 HWC
 * [B][H][W][C] -> fp32 NCHW, (v/255 - mean)/std in that operation order. */
int ppst_u8_to_tensor(const void* x, void* y, int B, int H, int W, int C, float mean, float stdv, void* stream);

/* ------------------------------------------------------- post-process ---- */
/* util.tensor2im quantisation (util/util.py:98-131): NCHW fp32 [-1,1] ->
 * HWC uint8, ((x+1)/2*255) clipped and truncated. */
int ppst_tensor2im_u8(const void* x, void* y, int B, int C, int H, int W, void* stream);
/* colour-guided filter (photo_gif.py:43 cv2.ximgproc.guidedFilter): guide, src
 * uint8 HWC [B][H][W][3]; out fp32 NCHW = (q/255 - 0.5)*2 as
 * PPSTModel.decode does (ppst_model.py:296-305).  work: >= ppst_guided_filter_ws() bytes. */
int64_t ppst_guided_filter_ws(int B, int H, int W);
int ppst_guided_filter(const void* guide_u8, const void* src_u8, void* out, void* out_u8,
                       int B, int H, int W, int r, float eps, void* work, void* stream);
/* tuning aid (process-wide, diagnostic like ppst_prof_enable): rows per block of the two fused launches of the radius-30 filter;
 * 0 = the library's rule, 32 / 64 (/ 128 for the second) = forced.  Results do not depend on it beyond the rounding of the sliding
 * fp32 sums of the second launch. */
int ppst_guided_filter_tune(int vs1, int vs2);

/* local-affine photo smoothing (smooth_filter.py:332-378 smooth_local_affine and its three NVRTC kernels :149-321):
 * output (stylised), input (content = guide), result: planar fp32 [B][3][H][W]; model_ws: >= ppst_smooth_local_affine_ws()
 * bytes, 16-B aligned (the per-pixel 3x4 maps); filtered_model: optional [B][H*W][12] fp32 (the smoothed maps), or NULL.
 * patch_radius = (patch - 1) / 2, filter_radius = f_r, sigma1 = f_r / 3, sigma2 = f_e as the reference's driver sets them. */
int64_t ppst_smooth_local_affine_ws(int B, int H, int W);
int ppst_smooth_local_affine(const void* output, const void* input, void* result, void* model_ws, void* filtered_model,
                             int B, int H, int W, int patch_radius, int filter_radius, float sigma1, float sigma2,
                             void* stream);

/* ------------------------------------------------- train step (backward) ---
 * Gradients of the discriminator update (optimizers/ppst_optimizer.py:96-130; torch autograd
 * of F.conv2d / F.linear in stylegan2_layers.py).  The conv INPUT gradient is
 * ppst_conv2d_mfma itself on a transposed/flipped pack of the weights. */
/* conv weight gradient, exact-fp32 MFMA, reduction over pixels; shares the forward step table
 * (steps: device ppst_conv_step[nsteps]; chunk_start: device int32[nchunks+1], steps of one
 * chunk share chan_off, at most 9 per chunk).  partial: [splits][nsteps][cout][32] fp32. */
int ppst_conv_wgrad_f32(const void* x, const void* dy, const void* steps, const void* chunk_start,
                        void* partial, int B, int in_h, int in_w, int in_ld, int oh, int ow,
                        int dy_ld, int cout, int nsteps, int nchunks, int splits, void* stream);
/* the same gradient on the bf16 matrix pipe, fp32-class through the hi/lo split of both operands (the production path;
 * ppst_conv_wgrad_f32 stays the exact verification path).  Same arguments; rows must be 16-B aligned, taps in [-1,1]^2. */
int ppst_conv_wgrad_bf16x3(const void* x, const void* dy, const void* steps, const void* chunk_start,
                           void* partial, int B, int in_h, int in_w, int in_ld, int oh, int ow,
                           int dy_ld, int cout, int nsteps, int nchunks, int splits, void* stream);
/* the production form of round 3 (same arithmetic: bf16 hi / lo split, three MFMA passes, fp32 accumulation): raw fp32 tiles by
 * LDS-DMA one tile ahead, one conversion pass per tile, transposed LDS reads (ds_read_b64_tr_b16) for the MFMA operands.
 * ``splits`` must be even: the grid covers splits / 2 pixel ranges and every block writes two partial slots.  ``csum`` (NULL or
 * [splits / 2][cout]): partial fp32 column sums of dy, fused into the conversion pass -- summed over the rows (ppst_colsum) they
 * are the bias gradient. */
int ppst_conv_wgrad_tr(const void* x, const void* dy, const void* steps, const void* chunk_start, void* partial, void* csum,
                       int B, int in_h, int in_w, int in_ld, int oh, int ow, int dy_ld, int cout, int nsteps, int nchunks,
                       int splits, void* stream);
/* the two-blocks-per-CU form of it (256-thread blocks, fp32 -> bf16 hi | lo converted in place in LDS, 49 KB per block): the
 * phases of one block (request, wait, convert, MFMA) overlap the other block's.  ``splits`` = partial slots = pixel ranges (any
 * positive count); csum NULL or [splits][cout].  max_taps / min_taps: longest / shortest chunk of the step table (0 = unknown):
 * <= 4 and an even chunk count select the two-chunks-per-block form, min == max == 9 (or 4) the form without per-tap tests.
 * halo: 1 in general; 0 = every step has offset (0, 0) and the input has the output's extent (1x1 convs): with one step per
 * chunk the block then stages no halo and takes four (two) chunks per dY image.
 * passes: 3 = bf16x3 (fp32-class), 1 = single-pass bf16 (precision mode 1: the
 * "bf16 compute, fp32 master weights" of BASELINE configs[3]); anything else PPST_EINVAL. */
int ppst_conv_wgrad_tr2(const void* x, const void* dy, const void* steps, const void* chunk_start, void* partial, void* csum,
                        int B, int in_h, int in_w, int in_ld, int oh, int ow, int dy_ld, int cout, int nsteps, int nchunks,
                        int splits, int max_taps, int min_taps, int halo, int passes, void* stream);   /* max_taps: 0 = unknown (<= 9); 1..4 = the caller's promise that
                        no chunk of the table has more steps: with an even chunk count two chunks then share one block / dY image */
/* dw[n*sn + (src_c+k)*sc + ky*sy + kx*sx] (+)= scale * sum_splits partial[.][step][n][k] */
int ppst_wgrad_scatter(const void* partial, const void* src_c, const void* src_ky, const void* src_kx,
                       void* dw, int64_t sn, int64_t sc, int64_t sy, int64_t sx, int cout, int nsteps,
                       int splits, float scale, int accumulate, const void* csum, void* db, int csum_rows, int db_accumulate,
                       void* stream);   /* csum (NULL or [csum_rows][cout] from ppst_conv_wgrad_tr*): db[n] (+)= sum of its rows */
/* FromRGB (Cin <= 4) weight gradient: dw[n][c] (+)= scale * sum_p dy[p][n]*x[p][c] */
int64_t ppst_wgrad_small_cin_ws(int64_t npix, int cin, int cout);
int ppst_wgrad_small_cin(const void* x, const void* dy, void* dw, void* ws, int64_t npix, int cin,
                         int in_ld, int cout, float scale, int accumulate, void* stream);
/* out[c] (+)= scale * sum_rows x[row][c]  (bias gradients) */
int64_t ppst_colsum_ws(int64_t rows, int C);
int ppst_colsum(const void* x, void* out, void* ws, int64_t rows, int C, int ld, float scale,
                int accumulate, void* stream);
/* F.linear gradients: dW[n][k] (+)= scale*sum_b dY[b][n] X[b][k];  dX[b][k] = scale*sum_n dY[b][n] W[n][k] */
int ppst_linear_wgrad(const void* dy, const void* x, void* dw, int B, int N, int K, float scale,
                      int accumulate, void* stream);
/* ws: ppst_linear_dgrad_ws(B, N, K) bytes of scratch (partial sums over row slices, reduced in a fixed order) */
int64_t ppst_linear_dgrad_ws(int B, int N, int K);
int ppst_linear_dgrad(const void* dy, const void* w, void* dx, void* ws, int B, int N, int K, float scale,
                      void* stream);
/* Round 5: the two linear gradients with the passes around them folded in (the E2 projector chains, encoder_col.py:47-93, ran seven
 * launches per linear and backward).  ppst_linear_wgrad_fused: ``relu_in`` reads x as max(x, 0) (the linear sits behind nn.ReLU);
 * ``db`` (optional, [N]) receives bscale * sum_b dy[b][n] (written, or added with b_accumulate); K % 4 == 0, 16-byte aligned x / dw.
 * ppst_linear_dgrad_gate: dx[b][k] = [gate[b][k] > 0] * scale * sum_n dy[b][n] w[n][k] (the ReLU's backward rides on the slice
 * reduction); ws as ppst_linear_dgrad_ws. */
int ppst_linear_wgrad_fused(const void* dy, const void* x, void* dw, void* db, int B, int N, int K, float scale, float bscale,
                            int accumulate, int b_accumulate, int relu_in, void* stream);
int ppst_linear_dgrad_gate(const void* dy, const void* w, void* dx, void* ws, const void* gate, int B, int N, int K, float scale,
                           void* stream);
/* LSGAN (models/networks/loss.py:11-18): loss = weight*mean((p-target)^2), grad = d loss / d p */
/* torch.nn.L1Loss (mean |a-b|) times weight -> out[0]; ws >= ppst_l1_mean_ws(n) bytes (ppst_model.py:47,183,203). */
int64_t ppst_l1_mean_ws(int64_t n);
int ppst_l1_mean(const void* a, const void* b, void* out, void* ws, int64_t n, float weight, void* stream);
/* rsclLoss.forward (networks/rscl.py:42-64): q, k [n][C] rows, k0 [n0][C] extra negatives, queue [C][K]; out[0] = mean
 * cross entropy at temperature nce_T with the reference's quirk that all n current-batch logits are -10 (its eye(1)
 * mask broadcasts).  n <= 64, K + n0 <= 512; ws >= n floats. */
int ppst_rscl_loss(const void* q, const void* k, const void* k0, const void* queue, void* out, void* ws,
                   int n, int n0, int C, int K, float nce_T, void* stream);
int ppst_lsgan(const void* pred, void* loss, void* grad, int n, float target, float weight, void* stream);
/* torch.optim.Adam step (ppst_optimizer.py:34-49), step counted from 1 */
int ppst_adam_step(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream);

/* ---- generator / encoder update of the train step (optimizers/ppst_optimizer.py:73-94: g_loss.backward()
 * through models/ppst_model.py:161-235).  These replace what torch autograd derives for the reference's
 * InstanceNorm2d + StyleMod (stylegan2_layers.py:361-374, 414-437), F.pad(reflect / replicate), F.interpolate,
 * adaptive avg / max pooling (encoder_col.py:150-251), F.normalize / util.normalize (util/util.py:18-22), L1Loss,
 * softmax and PReLU.  Conv input / weight gradients reuse ppst_conv2d_mfma / ppst_conv_wgrad_f32 with other step
 * tables. ---- */
/* ppst_in_finalize that also returns mean_rstd [B][C][2] (needed by the backward of the norm) */
int ppst_in_finalize_train(const void* partial, int n_partials, const void* style, int style_ld, const void* post_bias,
                           void* scale_shift, void* mean_rstd, int B, int C, double count, float eps, void* stream);
/* instance-norm backward, pass 1: partial [B][n][C][2] = (sum g', sum g'*y) over pixel chunks; g' = g, or with
 * `gate` (the activation FOLLOWS the norm, ConvLayer norm='in') g * lrelu'(gate).  x == partial == NULL: size query. */
int ppst_dual_stats(const void* g, const void* y, const void* gate, void* partial, int B, int64_t hw, int C, int g_ld, int y_ld,
                    int gate_ld, int* n_partials, void* stream);
/* pass 2: coef [B][C][4] = (k0, k1, k2, 0) with dy = k0*g' + k1*y + k2; dstyle [B][2C] = (sum g*n, sum g) when given
 * (style = the StyleMod linear output [B][>=C], row stride style_ld; NULL = no modulation).  mean_rstd == NULL:
 * no norm -- dstyle = (sum g*y, sum g) only (SpatialCodeModulation scale / shift gradient, generator.py:80-91). */
int ppst_in_bwd_finalize(const void* partial, int n_partials, const void* mean_rstd, const void* style, int style_ld, void* coef,
                         void* dstyle, int B, int C, double count, void* stream);
/* pass 3: dx = post * (k0*g' + k1*y + k2); post = lrelu'(y) when post_gate (StyledConv: activation before the norm) */
int ppst_in_bwd_apply(const void* g, const void* y, const void* gate, const void* coef, void* dx, int B, int64_t hw, int C, int g_ld,
                      int y_ld, int gate_ld, int dx_ld, int post_gate, void* stream);
/* PReLU backward over z = a*y + s [+ res]: gpre = g * (z >= 0 ? 1 : slope) (dense [.,C]); ws receives per-block partial
 * sums of dslope = sum g*z*[z<0] (ppst_prelu_bwd_ws bytes; reduce with ppst_sum_partials) */
int64_t ppst_prelu_bwd_ws(int64_t total_elements);
int ppst_prelu_bwd(const void* g, const void* y, const void* scale_shift, const void* res, const void* prelu, void* gpre, void* ws,
                   int B, int64_t hw, int C, int g_ld, int y_ld, int res_ld, void* stream);
int ppst_sum_partials(const void* partial, void* out, int n, float scale, void* stream);
/* F.pad (PPST_PAD_*) of an NHWC tensor and its adjoint (dx = sum of dy over the padded positions that read it) */
int ppst_pad2d(const void* x, void* y, int B, int H, int W, int C, int x_ld, int py0, int py1, int px0, int px1, int mode,
               void* stream);
int ppst_pad2d_bwd(const void* dy, void* dx, int B, int H, int W, int C, int py0, int py1, int px0, int px1, int mode, void* stream);
/* adjoints of ppst_bilinear (dx must be zero-initialised; float atomics), ppst_avgpool and ppst_gap_gmp
 * (v = the forward's [B][2C] output, g = its gradient; the max routes to the FIRST maximal pixel like nn.AdaptiveMaxPool2d;
 * accumulate != 0 adds into dx) */
int ppst_bilinear_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int OH, int OW, int dy_ld, void* stream);
int ppst_avgpool_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int f, int dy_ld, void* stream);
int ppst_gap_gmp_bwd(const void* x, const void* mask, const void* v, const void* g, void* dx, void* arg_ws /* B*C int32 */, int B,
                     int64_t hw, int C, int ld, int accumulate, void* stream);
/* Round 5: GAP || GMP of x * mask (encoder_col.py:162-168, 217-245) for SEVERAL masks in one read of the feature map, and its adjoint
 * in one pass.  heads: h = 0 the unmasked pooling (with_plain = 1), then one per channel of masks [B][hw][nm] (nm in 1..3: the NHWC
 * planes of the one-hot mask pyramid); out / v / g are [(nm + with_plain) * B][2C], head-major.  A head's forward values are the
 * single-head launch's bit for bit (same block geometry and summation order); the backward writes the SUM over the heads once
 * (accumulate: adds into dx).  C % 4 == 0, 16-byte aligned rows; the backward needs hw % 16 == 0.
 * ws: ppst_gap_gmp_multi_ws bytes; arg_ws: (nm + with_plain) * B * C ints. */
int64_t ppst_gap_gmp_multi_ws(int B, int64_t hw, int C, int heads);
int ppst_gap_gmp_multi(const void* x, const void* masks, void* out, void* ws, int B, int H, int W, int C, int ld, int nm,
                       int with_plain, int x_st, void* stream);
int ppst_gap_gmp_multi_bwd(const void* x, const void* masks, const void* v, const void* g, void* dx, void* arg_ws, int B, int64_t hw,
                           int C, int ld, int nm, int with_plain, int accumulate, int st /* x and dx */, void* stream);
/* backward of ppst_l2norm_rows (mode 0: util.normalize, 1: F.normalize), ppst_softmax_rows (in place on g) and
 * ppst_corr_prep (ppst_model.py:343-356) */
int ppst_l2norm_rows_bwd(const void* g, const void* x, void* dx, int B, int K, float eps, int mode, void* stream);
int ppst_softmax_rows_bwd(const void* p, void* g, int64_t rows, int cols, float div, void* stream);
int ppst_corr_prep_bwd(const void* g, const void* x, void* dx, int64_t rows, int C, int ncenter, float eps, void* stream);
/* d(weight * mean|a-b|)/da */
int ppst_l1_grad(const void* a, const void* b, void* da, int64_t n, float weight, void* stream);
/* backward of ppst_rscl_loss wrt the queries (keys / queue are detached, ppst_model.py:214-217): dq (n, C); gout = d/d(loss) (1) */
int ppst_rscl_loss_bwd(const void* q, const void* k, const void* k0, const void* queue, const void* gout, void* dq, int n, int n0,
                       int C, int K, float nce_T, void* stream);
/* backward of ppst_rselfcorr: dfea [B][H][W][64] from dout [B][H/4][W/4][>=256] (pixel stride dout_ld) */
int ppst_rselfcorr_bwd(const void* fea, const void* dout, void* dfea, int B, int H, int W, int C, int dout_ld, void* stream);
/* y = x * s[0], s a device scalar (chain rule through a scalar loss) */
int ppst_scale_by(const void* x, const void* s, void* y, int64_t n, void* stream);
/* NoiseInjection weight gradient: out[0] = sum dpre[p][c] * noise[p] (stylegan2_layers.py:376-399) */
int64_t ppst_noise_wgrad_ws(int64_t npix);
int ppst_noise_wgrad(const void* dpre, const void* noise, void* out, void* ws, int64_t npix, int C, int ld, int accumulate, void* stream);
/* adjoint of ppst_upscale_weight: dw4 (Cin,Cout,4,4) -> dw (Cout,Cin,3,3); accumulate != 0 adds into dw (like the other
 * parameter-gradient entry points: the trainers point them at the flat gradient buffer, no separate accumulation pass) */
int ppst_upscale_weight_bwd(const void* dw4, void* dw, int cout, int cin, float scale, int accumulate, void* stream);
/* NHWC [B][H][W][C] -> space-to-depth [B][ceil(H/2)][ceil(W/2)][4C] (channel block (py*2+px)*C) */
int ppst_space_to_depth(const void* x, void* y, int B, int H, int W, int C, int x_ld, void* stream);

/* ---------------------------------------------------------- profiling ----
 * Opt-in HIP-event timing of the conv launches (bench.py roofline): when
 * enabled every ppst_conv2d_mfma call is bracketed by events on its stream. */
int ppst_prof_enable(int on);
/* launches that were NOT bracketed since the last enable because the event pool was full */
int ppst_prof_dropped(void);
/* after a stream sync: total ms, launches, algorithmic flop of bracketed calls */
int ppst_prof_collect(double* ms, int64_t* launches, double* flop);
/* per-launch detail of bracketed call idx (before ppst_prof_collect resets the pool):
 * info = {B, tile_h, tile_w, nsteps, cout, n_groups, halo, bn}; a ppst_conv_wgrad_bf16x3 / ppst_conv_wgrad_tr launch is bracketed too, with
 * info = {B, oh, ow, nsteps, cout, nchunks, splits, 0} (bn = 0 marks it) */
int ppst_prof_detail(int idx, double* ms, double* flop, int32_t* info);
/* profiling only: number of steps of the NEXT weight-gradient launch that carry real weights (default: all of them) */
int ppst_wgrad_flop_steps(int flop_steps);

/* ---- Round 5: half-precision storage of the TRAINING activations and their gradients (precision mode 1: BASELINE configs[3]'s
 * "bf16" -- the accuracy class of bf16 autocast, where conv outputs and their gradients are bfloat16 tensors).  `_st` twins of the
 * backward kernels: ``st`` = the storage type (PPST_ST_*) of every ACTIVATION-shaped tensor of the call (inputs, gradients, outputs);
 * partial sums, coefficient tables, parameter gradients and pooled vectors stay fp32.  Arithmetic in fp32, one rounding at the store
 * (the contract of the forward `_st` kernels).  With st = PPST_ST_F32 each IS its plain form.  Half tensors: C and every leading
 * dimension a multiple of 4, 8-byte aligned. */
int ppst_dual_stats_st(const void* g, const void* y, const void* gate, void* partial, int B, int64_t hw, int C, int g_ld, int y_ld,
                       int gate_ld, int* n_partials, int st, void* stream);
int ppst_in_bwd_apply_st(const void* g, const void* y, const void* gate, const void* coef, void* dx, int B, int64_t hw, int C, int g_ld,
                         int y_ld, int gate_ld, int dx_ld, int post_gate, int st, void* stream);
int ppst_pad2d_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int py0, int py1, int px0, int px1, int mode, int st,
                  void* stream);
int ppst_pad2d_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int py0, int py1, int px0, int px1, int mode, int st,
                      void* stream);
int ppst_bilinear_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int OH, int OW, int dy_ld, int st, void* stream);
int ppst_gap_gmp_bwd_st(const void* x, const void* mask, const void* v, const void* g, void* dx, void* arg_ws, int B, int64_t hw, int C,
                        int ld, int accumulate, int st, void* stream);          /* st: x and dx */
int ppst_noise_wgrad_st(const void* dpre, const void* noise, void* out, void* ws, int64_t npix, int C, int ld, int accumulate, int st,
                        void* stream);                                          /* st: dpre */
int ppst_space_to_depth_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int st, void* stream);
int ppst_colsum_st(const void* x, void* out, void* ws, int64_t rows, int C, int ld, float scale, int accumulate, int st, void* stream);
int ppst_wgrad_small_cin_st(const void* x, const void* dy, void* dw, void* ws, int64_t npix, int cin, int in_ld, int cout, float scale,
                            int accumulate, int dy_st, void* stream);           /* x stays fp32 (the image / RGB gradient) */
/* ppst_conv_wgrad_tr2 on bf16-STORED operands (st = PPST_ST_BF16, passes = 1 only): the tiles arrive as the MFMA operand type by
 * LDS-DMA -- half the bytes of the fp32 tiles, no conversion pass, double-buffered images (conv_wgrad_tr2b_kernel).  cout, in_ld and
 * dy_ld multiples of 8, x / dy 16-byte aligned.  partial / csum as ppst_conv_wgrad_tr2. */
int ppst_conv_wgrad_tr2_st(const void* x, const void* dy, const void* steps, const void* chunk_start, void* partial, void* csum, int B,
                           int in_h, int in_w, int in_ld, int oh, int ow, int dy_ld, int cout, int nsteps, int nchunks, int splits,
                           int max_taps, int min_taps, int halo, int passes, int st, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPST_HIP_H */
