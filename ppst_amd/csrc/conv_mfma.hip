// Fused implicit-GEMM convolution for gfx950 (the "modulated_conv2d" of the north star =
// the dense conv inside StyledConv / EqualConv2d / nn.Conv2d of the PPST generator,
// encoders and discriminator; reference call sites: stylegan2_layers.py:184-193, :305-347,
// generator.py:10-32,174-238) with the StyledConv epilogue (stylegan2_layers.py:467-475:
// + noise*w, + bias, leaky-relu*sqrt2) and the instance-norm statistics fused in.
//
// GEMM view: M = 16-wide rows of output pixels of one image tile, N = output channels,
// K = (tap, 32 input channels).  Activations are NHWC fp32 in HBM.
//
//  * A (activations): one (TH+2)x(16+2) halo tile x 32 channels is staged per K-chunk into
//    LDS as bf16 hi / lo planes laid out [k-group g][pixel][8 ch] (16-B slots; plane stride a
//    multiple of 256 B so a 16-lane ds_read_b128 group hits 16 distinct slots: conflict
//    free for every tap shift).  The tile is re-used by all taps of the chunk (9x for 3x3)
//    and by all N-waves.  fp32 -> (hi, lo) bf16 split happens once per staged element.
//  * B (weights): pre-packed once (ppst_conv_pack) into per-step blobs that are already the
//    LDS image [hi/lo][g][n][8 k]; staging is a linear 16-B copy.
//  * MFMA: v_mfma_f32_16x16x32_bf16; fp32-class accuracy from 3 passes
//    (hi*hi + hi*lo + lo*hi, "bf16x3"), or 1 pass in bf16 mode.  Wave tile 64 px x 64 ch.
//  * Pipeline: two-deep software pipeline -- global loads of step s+2 and LDS fragment reads
//    of step s+1 are issued under the MFMAs of step s; one __syncthreads per K-step (= per
//    tap), >= 48 MFMAs per wave between barriers; A ring of 3 LDS slots, B ring of 2.
//  * Stride-2 convs run as stride-1 convs over a space-to-depth input, the 4x4 stride-2
//    transposed conv as 4 output-phase groups of 2x2 taps; both are just step tables.
//  * Block -> tile map is XCD-aware: the 8 XCDs each get a contiguous range of an N-major
//    ordering so that co-resident blocks of one XCD stream the same weight blobs from its L2.
#include "common.h"

__device__ __attribute__((aligned(16))) float g_conv_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct ConvKArgs {
  const float* x;
  const unsigned short* wpack;
  const int4* steps;
  float* y;
  const float* bias;
  const float* noise;
  const float* prelu;
  float* stats;
  const float* residual;
  float noise_weight, out_scale;
  int B, in_h, in_w, in_ld, out_h, out_w, out_ld, cout;
  int nsteps, n_groups, pad_mode, in_off_y, in_off_x, out_sy, out_sx, act, res_ld, tile_h, tile_w;
  int tiles_y, tiles_x, n_tiles;
  const float* in_ss;     // optional [B][in_c][2] (scale, shift) applied to the input while staging
  const float* in_prelu;  // slope for in_act == PRELU
  int in_c, in_act;
  int early_a;              // step table guarantees chunks of >= 2 steps: a chunk's global loads go out one step early
  unsigned long long* dbg;  // stamp build: [block][wave][8] cycle sums (else unused)
  KSplitDev ks;             // ks.S > 1: grid row y runs steps [ks.start[y], ks.start[y + 1]) of every group (common.h)
  int prefetch_w;           // small grid (<= 512 blocks): a block requests its weight blobs up front (see the kernel's prologue)
};
// (the storage type of x / residual / y -- ppst_conv_args.io_st, single-pass precision modes only -- is a template parameter IOS of
// the kernels, not a field: the pointers above are then half / bfloat16 tensors behind their `float*` type)

__device__ __forceinline__ int pad_index(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  // replicate, and the safety clamp for reflect on far-out (masked) tile pixels
  i = i < 0 ? 0 : (i >= n ? n - 1 : i);
  return i;
}

// NAS: activation-tile ring depth (0 = default rule below).  1 or 2 only when no group has more
// chunks than that; those variants also ask for two waves per SIMD so two blocks share a CU.
#ifdef PPST_CONV_CLOCK
__device__ unsigned long long* g_clock_buf = nullptr;
__device__ int g_clock_n = 1;
extern "C" int ppst_conv_clock_buffer(void* buf, int n) {   // diagnostic build only (not in include/ppst_hip.h)
  if (n <= 0) return PPST_EINVAL;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_buf), &buf, sizeof(buf)) != hipSuccess) return PPST_EINVAL;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_clock_n), &n, sizeof(n)) == hipSuccess ? PPST_OK : PPST_EINVAL;
}
#endif
// F16 (single-pass builds only): operands are IEEE half instead of bfloat16 (v_mfma_f32_16x16x32_f16): 11 significant
// bits instead of 8 at the same MFMA rate -- the "fp16 generator" mode of BASELINE configs[4]; accumulation, instance-norm
// statistics and StyleMod stay fp32 as in every mode.
typedef _Float16 __attribute__((ext_vector_type(8))) half8;
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
// x = hi + lo with hi, lo IEEE half: 22 significant bits while |x| stays inside the half range (activations do)
__device__ __forceinline__ void split_f16(float x, unsigned short& hi, unsigned short& lo) {
  const _Float16 h = (_Float16)x;
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, (_Float16)(x - (float)h));
}
// X2 (with F16, !X3): two passes -- the activation as fp16 hi + lo (22 significant bits), the weight rounded once to
// fp16 (11 bits): al*b + ah*b.  A measured experiment: 2/3 of the MFMAs of the fp32-class mode.
// PRES (experiment, PPST_EXPERIMENTS builds; VERDICT r2 "lever (i)"): the input is PRE-SPLIT -- per pixel and 8-channel group 32
// bytes [hi x 8 | lo x 8] bf16, same bytes and pixel stride as the fp32 tensor (ppst_presplit) -- and the activation tile is
// staged by LDS-DMA like the weights: no registers, no conversion, no staging store.  Two activation slots (needs chunks of
// >= 4 steps), no normalise-on-load.
template <int WM, int WN, int HALO, bool X3, bool INSS, int NAS = 0, bool F16 = false, bool X2 = false, bool PRES = false,
          int IOS = PPST_ST_F32>
__global__ __launch_bounds__(64 * WM * WN, NAS ? 2 : 1) void conv_mfma_kernel(ConvKArgs a) {
  constexpr bool ALO = X3 || X2;                         // the activation tile has lo planes
  // IOS: storage type of x, residual and y (common.h).  Half storage exists for the single-pass modes only, in the operand type of
  // the mode: a staged element then reaches LDS as loaded (no conversion) unless normalise-on-load rewrites it.
  static_assert(IOS == PPST_ST_F32 || (!X3 && !X2 && !PRES && IOS == (F16 ? PPST_ST_F16 : PPST_ST_BF16)), "half storage: single-pass modes");
  constexpr int ES = IOS == PPST_ST_F32 ? 4 : 2;         // bytes per stored element
  constexpr int NT = 64 * WM * WN;
  constexpr int TH = 4 * WM, TW = 16;
  constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO, HP = HH * HW;
  // bytes.  (The 256-B rounding is layout hygiene, not a bank requirement: a 16-lane ds_read_b128 / ds_write_b64 group stays inside
  // one plane.  The 8-row two-block form drops it so that two 77-KB blocks fit the CU's 160 KB.)
  constexpr int PLANE = (WM == 2) ? HP * 16 : ((HP * 16 + 255) / 256) * 256;
  constexpr int NPL = ALO ? 8 : 4;                       // planes per A buffer (hi g0..3, lo g0..3)
  constexpr int NPLB = X3 ? 8 : 4;                       // planes per weight blob
  constexpr int ABUF = NPL * PLANE;
  constexpr int BN = 64 * WN;
  constexpr int BPLANE = BN * 16;
  constexpr int BBUF = NPLB * BPLANE;
  constexpr int A_ITEMS = HP * 8;                        // float4 items per chunk
  constexpr int A_IT = (A_ITEMS + NT - 1) / NT;
  constexpr int B_ITEMS = BBUF / 16;
  constexpr int B_IT = (B_ITEMS + NT - 1) / NT;
  static_assert(B_ITEMS % NT == 0, "B blob must split evenly");

  // A ring slots (hazard note at the main loop): 3 in general; NAS = 1 / 2 when the step table has no more chunks
  // than that (smaller LDS footprint, two blocks per CU).
  constexpr int NA = PRES ? 2 : (NAS ? NAS : 3);
  static_assert(!PRES || (NT == 512 && HP <= 384 && !INSS && X3), "pre-split staging: the 8-wave tile kernel, 6 pieces per plane");
  constexpr int EPI_TILE = 64 * 36;                              // floats per wave: 64 px x (32 ch + 4 pad)
  constexpr int EPI_BYTES = (NT / 64) * EPI_TILE * 4 + WM * BN * 2 * 4;  // transposition tiles + stats scratch
  constexpr int MAIN_BYTES = NA * ABUF + 2 * BBUF;
  __shared__ __attribute__((aligned(256))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  unsigned char* smA = smem;
  unsigned char* smB = smem + NA * ABUF;

  // ---- XCD-aware block -> (n index, m tile) map (bijective remap, N-major order)
  const int nwg = gridDim.x;
  int wid;
  {
    int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int m_count = a.B * a.tiles_y * a.tiles_x;
  const int nidx = wid / m_count;          // group * n_tiles + ntile
  int midx = wid - nidx * m_count;
  const int group = nidx / a.n_tiles, ntile = nidx - group * a.n_tiles;
  const int b = midx / (a.tiles_y * a.tiles_x);
  midx -= b * a.tiles_y * a.tiles_x;
  const int tyi = midx / a.tiles_x, txi = midx - tyi * a.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int r16 = lane & 15, g = lane >> 4;

#if defined(PPST_CONV_TRACE) || defined(PPST_CONV_CLOCK)
  unsigned long long tr_c0, tr_r0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
#endif
  // constant address space: the uniform-index descriptor reads become scalar loads (lgkmcnt), so using a descriptor
  // never makes hipcc wait on vmcnt -- which would also drain an early activation load that is meant to stay in flight
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) int4* StepPtr;
#else
  typedef const int4* StepPtr;
#endif
  StepPtr steps = (StepPtr)(a.steps + (int64_t)group * a.nsteps);
  const unsigned char* wblob = (const unsigned char*)a.wpack + ((int64_t)nidx * a.nsteps) * BBUF;
  const unsigned char* xb = (const unsigned char*)a.x + (int64_t)b * a.in_h * a.in_w * a.in_ld * ES;
  // across-block K split (ppst_conv_args.ksplit): this block runs `nst` steps from step start[y] on -- the first of them opens a
  // chunk (the caller's promise), so the sub-table is a table of its own; a next-chunk flag on its last step requests one tile that
  // is never stored (drained by the last step's vmcnt(0))
  int nst = a.nsteps;
  if (a.ks.S > 1) {
    int s0, s1;
    ks_range(a.ks, (int)blockIdx.y, s0, s1);
    steps += s0;
    wblob += (int64_t)s0 * BBUF;
    nst = s1 - s0;
  }
  constexpr int A_WCH = (HP + 7) / 8;                      // wave-chunks of 8 pixels x 8 float4
  constexpr int A_IT2 = (A_WCH * 64 + NT - 1) / NT;
  float4 ra[A_IT2];
  constexpr int A_NLOADS = PRES ? 6 : A_IT2 + (INSS ? 2 : 0);   // vector-memory operations of one a_load / a_dma

  // A staging.  One wave-instruction covers 8 pixels x 128 B (fully coalesced global read);
  // inside it lane l -> plane g = l>>4, pixel (l>>1)&7, half h = l&1, so the 16 lanes of a
  // ds_write_b64 group write 128 contiguous bytes of ONE plane (conflict-free; the planes
  // alias each other's banks because their stride is a multiple of 256 B).
  // per-item input pixel offsets (elements, -1 = zero padding) are fixed for the whole tile:
  // compute them once, the per-chunk load is then one add + one 16-B load per item
  int aoff[A_IT2];
#pragma unroll
  for (int it = 0; it < A_IT2; ++it) {
    int i = tid + it * NT;
    int l = i & 63;
    int pix = (i >> 6) * 8 + ((l >> 1) & 7), q4 = (l >> 4) * 2 + (l & 1);
    int o = -1;
    if (pix < HP) {
      int hy = pix / HW, hx = pix - hy * HW;
      int iy = ty0 + hy - HALO + a.in_off_y, ix = tx0 + hx - HALO + a.in_off_x;
      bool inb = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
      if (inb || a.pad_mode != PPST_PAD_ZERO) {
        iy = pad_index(iy, a.in_h, a.pad_mode);
        ix = pad_index(ix, a.in_w, a.pad_mode);
        o = ((iy * a.in_w + ix) * a.in_ld + q4 * 4) * ES;  // bytes, < 2^31: the entry point rejects larger images
      }
    }
    aoff[it] = o;
  }
  // normalise-on-load: the instance-norm / StyleMod affine (and activation) of the producer
  // layer is applied to the tile while it is staged, so that layer needs no apply pass.
  // A thread's 4 channels are the same for all its items: (a, s) x 4 loaded once per chunk.
  float4 ras0 = make_float4(1.f, 0.f, 1.f, 0.f), ras1 = ras0;
  const int q4lane = ((tid & 63) >> 4) * 2 + (tid & 1);
  const float in_slope = (INSS && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  // Buffer loads: address = image base (SGPR descriptor) + per-item byte offset (VGPR, fixed for the tile) + chunk offset (SGPR
  // soffset).  No per-load 64-bit VALU address arithmetic -- with `xb + aoff + chan_off` as a flat pointer hipcc formed each address
  // in the load's own destination registers, and that VALU write to registers of a possibly pending load cost a `s_waitcnt vmcnt(0)`
  // right behind the weight DMA of the step (one L2 round trip per chunk).  Padding items carry the offset -1 (0xffffffff): out of
  // the descriptor's range, the hardware returns zeros and the staging store needs no select.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, a.in_h * a.in_w * a.in_ld * ES, 0x00020000);
  // PRES: piece k (0..5) of this wave is plane (wave + 8k) / 6, pixels 64 j .. 64 j + 63 with j = (wave + 8k) % 6; a piece is one
  // global_load_lds_dwordx4 (1 KB of one plane); out-of-image pixels of a zero-padded conv read a 16-byte zero word
  int poff[6];
  if (PRES) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int pix = 64 * ((wave + 8 * k) % 6) + lane;
      int o = -1;
      if (pix < HP) {
        int hy = pix / HW, hx = pix - hy * HW;
        int iy = ty0 + hy - HALO + a.in_off_y, ix = tx0 + hx - HALO + a.in_off_x;
        bool inb = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
        if (inb || a.pad_mode != PPST_PAD_ZERO) {
          iy = pad_index(iy, a.in_h, a.pad_mode);
          ix = pad_index(ix, a.in_w, a.pad_mode);
          o = (iy * a.in_w + ix) * a.in_ld * 4;
        }
      }
      poff[k] = o;
    }
  }
  auto a_dma = [&](int chan_off, int slot) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = wave + 8 * k, pl = i / 6, j = i - pl * 6;       // wave-uniform
      const unsigned char* src = (const unsigned char*)g_conv_zero;
      if (poff[k] >= 0) src = xb + poff[k] + chan_off * 4 + (pl & 3) * 32 + (pl >> 2) * 16;
      if (64 * j + lane < HP)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(smA + slot * ABUF + pl * PLANE + j * 1024), 16, 0, 0);
    }
  };
  auto a_load = [&](int chan_off) {
#pragma unroll
    for (int it = 0; it < A_IT2; ++it) {
      // always issued: the count of outstanding vector-memory operations must be a compile-time constant for the counted
      // vmcnt wait that lets an early load stay in flight across the step barrier
      if (IOS == PPST_ST_F32) {
        ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff[it], chan_off * 4, 0));
      } else {       // four half elements: the raw 8 bytes ride in .x / .y until a_store
        const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrs, aoff[it], chan_off * 2, 0));
        ra[it].x = __uint_as_float(u.x); ra[it].y = __uint_as_float(u.y);
      }
    }
    if (INSS) {
      const float4* p = (const float4*)(a.in_ss + ((int64_t)b * a.in_c + chan_off + q4lane * 4) * 2);
      ras0 = p[0];
      ras1 = p[1];
    }
  };
  auto in_act = [&](float t) -> float {
    if (a.in_act == PPST_ACT_LRELU) return (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
    if (a.in_act == PPST_ACT_PRELU) return t >= 0.f ? t : t * in_slope;
    return t;
  };
  auto a_store = [&](int slot) {
    unsigned char* base = smA + slot * ABUF;
#pragma unroll
    for (int it = 0; it < A_IT2; ++it) {
      int i = tid + it * NT;
      int l = i & 63;
      int pix = (i >> 6) * 8 + ((l >> 1) & 7);
      if (pix < HP) {
        float4 v = ra[it];                // (padding items: zeros from the out-of-range buffer load)
        const uint2 raw = make_uint2(__float_as_uint(ra[it].x), __float_as_uint(ra[it].y));
        if (IOS != PPST_ST_F32) v = st_unpack4<IOS>(raw);
        if (INSS && aoff[it] >= 0) {  // padding zeros stay zeros (they pad the normalised tensor)
          v.x = in_act(ras0.x * v.x + ras0.y); v.y = in_act(ras0.z * v.y + ras0.w);
          v.z = in_act(ras1.x * v.z + ras1.y); v.w = in_act(ras1.z * v.w + ras1.w);
        }
        uint2 hv, lv;
        if (IOS != PPST_ST_F32 && !INSS) {
          hv = raw; lv = make_uint2(0u, 0u);       // stored in the operand type already
        } else if (F16) {
          unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
          if (X2) { split_f16(v.x, h0, l0); split_f16(v.y, h1, l1); split_f16(v.z, h2, l2); split_f16(v.w, h3, l3); }
          else { h0 = f2h(v.x); h1 = f2h(v.y); h2 = f2h(v.z); h3 = f2h(v.w); l0 = l1 = l2 = l3 = 0; }
          hv = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
          lv = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
        } else {
          split_bf16x4(v, hv, lv);          // two elements per conversion / subtraction instruction (common.h)
        }
        int off = (l >> 4) * PLANE + pix * 16 + (l & 1) * 8;
        *(uint2*)(base + off) = hv;
        if (ALO) *(uint2*)(base + 4 * PLANE + off) = lv;
      }
    }
  };
  // B staging: LDS-DMA (global_load_lds_dwordx4): each wave-instruction copies 1 KB of the
  // pre-packed step blob straight into the ring slot (LDS address = wave-uniform base +
  // lane*16); no VGPR round trip, no ds_write.  Completion is covered by the vmcnt(0) that
  // __syncthreads() places in front of the barrier ending the step.
  constexpr int B_WI = BBUF / 1024;            // wave-instructions per step blob
  constexpr int NW = NT / 64;
  static_assert(B_WI % NW == 0, "B blob must split evenly over the waves");
  auto b_dma = [&](int s, int slot) {
    const unsigned char* src = wblob + (int64_t)s * BBUF + lane * 16;
    unsigned char* dst = smB + slot * BBUF;
#pragma unroll
    for (int it = 0; it < B_WI / NW; ++it) {
      const int wi = it * NW + wave;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + wi * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + wi * 1024), 16, 0, 0);
    }
  };
  // fragment reads
  auto ld_b = [&](bf16x8 (&h)[4], bf16x8 (&lo)[4], int slot) {
    const unsigned char* Bb = smB + slot * BBUF + g * BPLANE + (wn * 64 + r16) * 16;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      h[nt] = *(const bf16x8*)(Bb + nt * 256);
      if (X3) lo[nt] = *(const bf16x8*)(Bb + 4 * BPLANE + nt * 256);
    }
  };
  // LDS byte offset of the A fragment (tile row mt) of a step with tap (dy, dx) in ring slot `slot`
#define A_OFF(slot, dy, dx, mt) ((slot) * ABUF + g * PLANE + (((wm * 4 + (mt) + HALO + (dy)) * HW + HALO + (dx) + r16) * 16))

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline -------------------------------------------------------------
  // During step s a wave (1) issues the global loads of step s+2 (B blob; A chunk if step
  // s+2 opens one), (2) reads the LDS fragments of step s+1 (all B fragments, A fragment
  // mt=0) interleaved with the MFMAs of step s (whose B fragments were read during step
  // s-1; its A fragments mt=1..3 rotate in one tile ahead), (3) writes the step-s+2 data to
  // LDS and hits the one barrier of the step.  Hazards: B(s+2) overwrites the slot of B(s)
  // (last read during step s-1: a barrier ago) -> 2 B slots.  A(chunk of s+2) must not
  // overwrite the chunks of steps s and s+1, both still being read by slower waves -> 3 A
  // slots (chunk index mod 3), which also covers single-step chunks (1x1 convs).
  // step descriptors live in scalars (cur = 0, next = 1); the descriptor of step s+2 is
  // loaded one step before it is needed (its scalar-load latency would otherwise sit
  // between the barrier and the first instruction of every step)
  // Small grids (<= 512 blocks, split or not): every block streams weight blobs nobody has touched since the last optimizer step: the
  // chain of steps then runs at memory latency (two blobs in flight), ~0.35 us per step slower than on hot weights (rocprofv3 of the
  // train step against tests/conv_ksplit_time.py).  The block requests its whole share up front -- one 128-byte line per thread and
  // round, results unused -- so the blobs are in this XCD's L2 when the DMA asks for them.  The requests retire with the prologue's
  // own vmcnt(0) (the same memory round trip), before the accumulators are live: no register of the loop is theirs.
  constexpr int PF_ROUNDS = 6;
  float pf[PF_ROUNDS];
  if (a.prefetch_w) {
    const int pf_bytes = nst * BBUF;
#pragma unroll
    for (int r = 0; r < PF_ROUNDS; ++r) {
      const int o = (r * NT + tid) * 128;
      pf[r] = o < pf_bytes ? *(const volatile float*)(wblob + o) : 0.f;
    }
  }
  int4 d = steps[0];
  int dy0 = d.y, dx0 = d.z, sl0 = 0;
  int dy1 = d.y, dx1 = d.z, sl1 = 0;
  if (PRES) a_dma(d.x, 0); else a_load(d.x);
  b_dma(0, 0);
  if (!PRES) a_store(0);
  if (nst > 1) {
    d = steps[1];
    dy1 = d.y; dx1 = d.z;
    sl1 = (d.w & 1) ? 1 : 0;
    b_dma(1, 1);
    if (!PRES) {                    // (pre-split: chunks span >= 4 steps, steps 1 and 2 open none)
      if (d.w & 1) a_load(d.x);
      if (d.w & 1) a_store(sl1);
      // early mode: the chunk opened by step 2 is staged during step 0 and must already be in registers
      if (a.early_a && (d.w & 2)) a_load(d.w >> 8);
    }
  }
  int4 dE = d, dO = d;                       // descriptor of step s+2, alternating register sets
  if (nst > 2) dE = steps[2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of steps 0/1 (see the note at the loop barrier)
  if (a.prefetch_w) {
#pragma unroll
    for (int r = 0; r < PF_ROUNDS; ++r) asm volatile("" ::"v"(pf[r]));
  }
  __syncthreads();

  bf16x8 b0h[4], b0l[4], b1h[4], b1l[4];
  bf16x8 ah, al;
  ld_b(b0h, b0l, 0);
  ah = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, 0));
  if (ALO) al = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, 0) + 4 * PLANE);
  // drain the prologue's LDS reads so both edges into the loop header carry an empty LDS
  // scoreboard (otherwise hipcc makes the first MFMAs of every step wait for the prefetch
  // reads issued just before them)
  __builtin_amdgcn_s_waitcnt(0xC07F);

// Diagnostic build -DPPST_CONV_TRACE: per-step timeline of every wave of the first blocks.  s_memtime results land
// asynchronously in their own SGPR pairs and are only waited for after the step's barrier, so the stamps do not drain
// the LDS pipeline.  dbg layout: [block < TR_BLOCKS][wave][step < TR_STEPS][8] cycle stamps
// relative to the step start: 1 head issued, 2 MFMA groups 0-1 issued, 3 groups 2-3 issued, 4 staging store done,
// 5 vmcnt wait done, 6 barrier passed; slot 0 = absolute start, 7 = newA2 flag.
#ifdef PPST_CONV_TRACE
#define TR_BLOCKS 8
#define TR_STEPS 160
#define TR(i) asm volatile("s_memtime %0" : "=s"(tr_[i])::"memory");
#define TR_DECL unsigned long long tr_[7];
#define TR_FLUSH(s, flag)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  if (a.dbg && blockIdx.x < TR_BLOCKS && (s) < TR_STEPS && lane == 0) {                               \
    unsigned long long* o_ = a.dbg + ((((int64_t)blockIdx.x * (NT / 64) + wave) * TR_STEPS) + (s)) * 8; \
    o_[0] = tr_[0];                                                                                   \
    for (int q_ = 1; q_ < 7; ++q_) o_[q_] = tr_[q_] - tr_[0];                                         \
    o_[7] = (flag) ? 1 : 0;                                                                           \
  }
#else
#define TR(i)
#define TR_DECL
#define TR_FLUSH(s, flag)
#endif
// timing ablations (tests/conv_ablate.sh): results are WRONG with any of these defined
#ifdef PPST_ABL_NOA
#define ABL_A(c) false
#else
#define ABL_A(c) (c)
#endif
#ifdef PPST_ABL_NOB
#define ABL_B(c) false
#else
#define ABL_B(c) (c)
#endif
#if defined(PPST_ABL_NOLDS) || defined(PPST_ABL_NOLDS_A)
#define ABL_LA(c) false
#else
#define ABL_LA(c) (c)
#endif
#if defined(PPST_ABL_NOLDS) || defined(PPST_ABL_NOLDS_B)
#define ABL_LB(c) false
#else
#define ABL_LB(c) (c)
#endif
#ifdef PPST_ABL_NOBAR
#define ABL_BAR(x)
#else
#define ABL_BAR(x) x
#endif
#define ABL_MFMA_GROUP(bch, bcl, mt)                                                                  \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                  \
    if (X3) {                                                                                         \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bch[nt], acc[mt][nt], 0, 0, 0);       \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcl[nt], acc[mt][nt], 0, 0, 0);       \
    }                                                                                                 \
    if (X2) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al), __builtin_bit_cast(half8, bch[nt]), acc[mt][nt], 0, 0, 0); \
    if (F16) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah), __builtin_bit_cast(half8, bch[nt]), acc[mt][nt], 0, 0, 0); \
    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bch[nt], acc[mt][nt], 0, 0, 0);    \
  }
#define TOP_WORK(bnh, bnl, s, D2)                                                                     \
  if (ABL_B(has2)) b_dma((s) + 2, (s) & 1);                                                           \
  /* the counted vmcnt(A_NLOADS) at the step barrier assumes the weight DMA was ISSUED before the */  \
  /* activation loads (vmcnt retires in issue order): pin that order                              */  \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  {   /* early mode: the chunk that step s+3 opens (stored during step s+1); else the chunk of step s+2 (stored   */ \
      /* in this step).  ONE a_load call site: two would make hipcc merge their results with copies + vmcnt(0).    */ \
    const bool ld_ = a.early_a ? ABL_A(has2 && (D2.w & 2)) : newA2;                                    \
    const int ch_ = a.early_a ? (D2.w >> 8) : D2.x;                                                   \
    if (ld_) { if (PRES) a_dma(ch_, (sl2 == NA - 1) ? 0 : sl2 + 1); else a_load(ch_); a_early = a.early_a != 0; } \
  }                                                                                                   \
  if (ABL_LB(has1)) ld_b(bnh, bnl, ((s) + 1) & 1);
#define STEP_HEAD_IF(cond, bnh, bnl, s, D2, D3)                                                       \
  if (cond) {                                                                                         \
    D3 = steps[(s) + 3];   /* the host pads the table: always in bounds, ignored past the end */       \
    newA2 = ABL_A(has2 && (D2.w & 1));                                                                \
    a_early = false;                                                                                  \
    sl2 = sl1;                                                                                        \
    if (newA2) sl2 = (sl1 == NA - 1) ? 0 : sl1 + 1;                                                   \
    TOP_WORK(bnh, bnl, s, D2)                                                                         \
  }
// MFMA_FIRST (default): a step opens with the 12 MFMAs of its first M-tile group -- their operands were fetched
// during the previous step -- and its non-MFMA head (descriptor, weight DMA issue, global loads, next-step B fragment
// reads) follows while the matrix pipe runs them.  The in-kernel trace (-DPPST_CONV_TRACE) showed the pipe idle for
// the ~365 cycles of the head of every step otherwise.  -DPPST_HEAD_FIRST restores the old order for A/B runs.
#ifdef PPST_HEAD_FIRST
#define MFMA_FIRST 0
#else
#define MFMA_FIRST 1
#endif
// -DPPST_PRIO_SWAP (experiment, measured: no effect -- two same-box A/B pairs, -0.5 % / +1.8 % on the step's conv time): the two waves of a SIMD take the issue priority in turns within a step (waves 4-7 for M-tile
// groups 0-1, waves 0-3 for groups 2-3) instead of waves 0-3 leading every step and waiting ~800 cycles at its barrier.
#ifdef PPST_PRIO_SWAP
#define PRIO_SWAP(mt)                                                                                 \
  if ((mt) == 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); } \
  if ((mt) == 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
#else
#define PRIO_SWAP(mt)
#endif
#define CONV_STEP(bch, bcl, bnh, bnl, s, D2, D3, H1, H2)                                              \
  {                                                                                                   \
    TR_DECL TR(0)                                                                                     \
    const bool has1 = (H1), has2 = (H2);   /* steps s+1 / s+2 exist (compile-time true in the steady-state loop) */ \
    bool newA2, a_early;                                                                              \
    int sl2;                                                                                          \
    STEP_HEAD_IF(MFMA_FIRST == 0, bnh, bnl, s, D2, D3)                                                \
    TR(1)                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                \
      bf16x8 nh, nl;                                                                                  \
      if (!ABL_LA(true)) {                                                                            \
        nh = ah; nl = al;                                                                             \
      } else if (mt < 3) {                                                                            \
        nh = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, mt + 1));                                    \
        if (ALO) nl = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, mt + 1) + 4 * PLANE);               \
      } else if (has1) {                                                                              \
        nh = *(const bf16x8*)(smA + A_OFF(sl1, dy1, dx1, 0));                                         \
        if (ALO) nl = *(const bf16x8*)(smA + A_OFF(sl1, dy1, dx1, 0) + 4 * PLANE);                    \
      }                                                                                               \
      PRIO_SWAP(mt)                                                                                   \
      ABL_MFMA_GROUP(bch, bcl, mt)                                                                    \
      if (MFMA_FIRST && mt == 0) {                                                                    \
        /* the matrix pipe is running the 12 MFMAs of group 0 (operands fetched during the previous */ \
        /* step): the step's non-MFMA head goes here instead of in front of them                    */ \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        STEP_HEAD_IF(true, bnh, bnl, s, D2, D3)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }                                                                                               \
      if (mt == 1) { TR(2) }                                                                          \
      if (mt == 3) { TR(3) }                                                                          \
      ah = nh;                                                                                        \
      if (ALO) al = nl;                                                                               \
      /* convert + write the next chunk's tile while the last MFMA group executes: the VALU   */     \
      /* work of the staging store overlaps the matrix pipe instead of following it            */     \
      if (!PRES && mt == 2 && newA2) a_store(sl2);                                                    \
    }                                                                                                 \
    TR(4)                                                                                             \
    /* hipcc (ROCm 7.2) does NOT add vmcnt(0) for an in-flight LDS-DMA at this barrier (it only  */  \
    /* emits lgkmcnt(0)): without the explicit wait a slow (cold-cache) B copy lands after the   */  \
    /* next step has started reading the slot.                                                   */  \
    /* vmcnt counts in issue order: the B copy (older) must have landed; an early chunk load (younger,   */  \
    /* A_NLOADS operations) may stay in flight across the barrier.                                       */  \
    if (a_early) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_NLOADS) : "memory");                      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    TR(5)                                                                                             \
    /* raw barrier: __syncthreads() carries a fence for which hipcc drains vmcnt to 0 -- including the early    */  \
    /* activation load.  LDS writes of this step (staging store) are drained here by hand.                      */  \
    ABL_BAR(asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");)                          \
    TR(6)                                                                                             \
    TR_FLUSH(s, newA2)                                                                                \
    dy0 = dy1; dx0 = dx1; sl0 = sl1;                                                                  \
    if (has2) { dy1 = D2.y; dx1 = D2.z; }                                                             \
    sl1 = sl2;                                                                                        \
  }
  // steady state: both steps of the unrolled pair have successors s+1 and s+2 (no existence tests / branches);
  // the last <= 3 steps run the general form
  int s = 0;
  for (; s + 3 < nst; s += 2) {
    CONV_STEP(b0h, b0l, b1h, b1l, s, dE, dO, true, true)
    CONV_STEP(b1h, b1l, b0h, b0l, s + 1, dO, dE, true, true)
  }
  for (; s < nst; s += 2) {
    CONV_STEP(b0h, b0l, b1h, b1l, s, dE, dO, s + 1 < nst, s + 2 < nst)
    if (s + 1 < nst) CONV_STEP(b1h, b1l, b0h, b0l, s + 1, dO, dE, s + 2 < nst, s + 3 < nst)
  }
#undef CONV_STEP
#undef STEP_HEAD_IF
#undef TOP_WORK
#undef A_OFF

  // ---- across-block K split: rows 0 .. S-2 hand their partial sums over and leave; row S-1 adds them (row order) and goes on
  if (a.ks.S > 1) {
    const int S = a.ks.S, y = (int)blockIdx.y;
    float* const slot0 = a.ks.scratch + (int64_t)wid * (S - 1) * (64 * NT) + tid;
    if (y < S - 1) {
      float* const dst = slot0 + (int64_t)y * (64 * NT);
      KS_SCATTER(acc, 4, 4, NT, dst)
      ks_publish(a.ks, wid, y, tid);
      return;
    }
    ks_wait(a.ks, wid, tid);
    KS_GATHER(acc, 4, 4, NT, S, slot0);
  }

  // ---- epilogue: + bias + noise [+ residual] -> act -> * out_scale -> store, tile statistics.
  // The accumulators (lane = channel, registers = pixels) are transposed through a per-wave
  // LDS tile so that every lane stores 16 contiguous bytes of one pixel (128-B segments per
  // pixel per wave-instruction instead of 64-B ones, 4x fewer store instructions, and the
  // pixel address arithmetic runs once per float4).  Two passes of 32 channels each.
  const int gy = group >> 1, gx = group & 1;  // output phase of the transposed conv
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;  // residual joins after the activation (resnet skip)
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  float* tw = (float*)smem + wave * EPI_TILE;            // main-loop buffers are dead: last barrier passed
  float* red = (float*)smem + (NT / 64) * EPI_TILE;      // [WM][BN][2]
  const int f8 = lane & 7, prow = lane >> 3;
  // Output addressing: 32-bit element offsets inside image b (the entry point rejects images of >= 2^31 elements); the tile row
  // and the column half of a store are wave-uniform, so their strides are scalar.  (With `opix` as a 64-bit product per store
  // the epilogue was bound by v_mul_lo_u32 / v_mad_u64_u32: ~45 vector instructions per store, a third of them quarter-rate.)
  const int egy = a.n_groups > 1 ? gy : 0, egx = a.n_groups > 1 ? gx : 0;
  const int tyb = ty0 + wm * 4, oyb = tyb * a.out_sy + egy;               // first tile / output row of this wave (uniform)
  const int txl = tx0 + prow, oxl = txl * a.out_sx + egx;                 // this lane's column for even `it` (odd: 8 further)
  const bool okx0 = txl < a.tile_w && oxl < a.out_w, okx1 = txl + 8 < a.tile_w && oxl + 8 * a.out_sx < a.out_w;
  const int pix0 = oyb * a.out_w + oxl, rs_pix = a.out_sy * a.out_w;
  const int64_t img = (int64_t)b * a.out_h * a.out_w;
  unsigned char* const yb = (unsigned char*)a.y + img * a.out_ld * ES;
  const float* const nzb = a.noise ? a.noise + img : nullptr;
  const unsigned char* const rb = a.residual ? (const unsigned char*)a.residual + img * a.res_ld * ES : nullptr;
  // Bias and noise of the whole wave tile are fetched BEFORE the first store: vmcnt counts stores too and retires in order, so a
  // load issued behind a pass's stores made its s_waitcnt sit out their acknowledgements (one HBM round trip per pass).
  // (buffer loads, every request unconditional -- a pixel outside the tile / image and a launch without noise read out of range and
  //  get zero: as conditional global loads hipcc issued them one by one, each with its own wait; conv_mfma2.hip's UP9 epilogue)
  float nzv[8];
  float4 bva[2];
  const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(nzb ? (const void*)nzb : (const void*)yb), 0,
                                                                       nzb ? a.out_h * a.out_w * 4 : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i >> 1, c8 = i & 1;
    const bool ok = tyb + r < a.tile_h && oyb + r * a.out_sy < a.out_h && (c8 ? okx1 : okx0);
    nzv[i] = a.noise_weight * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  nrs, ok ? (pix0 + r * rs_pix + c8 * 8 * a.out_sx) * 4 : (int)0x80000000, 0, 0));
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int n0 = ntile * BN + wn * 64 + pass * 32 + f8 * 4;
    // (unconditional as the noise loads above: channels beyond cout and a launch without bias read out of range -> zeros)
    bva[pass] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.y), 0, a.bias ? a.cout * 4 : 0, 0x00020000), n0 * 4, 0, 0));
  }
  // The passes are instantiated per (activation, residual mode): as run-time values they cost ~30 VALU instructions per element
  // (both activation branches evaluated and selected), ~10 when specialised (conv_mfma2.hip).
  auto epi_passes = [&](auto act_c, auto res_c) {
#pragma clang fp contract(off)   // no fused multiply-add here: every kernel family's epilogue must round like the others'
    const int ACT = act_c.value, RES = res_c.value;   // RES: 0 none, 1 joins before the activation, 2 after
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ntl = 0; ntl < 2; ++ntl)
#pragma unroll
        for (int j = 0; j < 4; ++j) tw[(mt * 16 + g * 4 + j) * 36 + ntl * 16 + r16] = acc[mt][pass * 2 + ntl][j];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): only this wave touches its tile
    __builtin_amdgcn_wave_barrier();
    const int nl0 = wn * 64 + pass * 32 + f8 * 4;
    const int n0 = ntile * BN + nl0;
    const bool nok = n0 < a.cout;
    const float4 bv = bva[pass];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int yo0 = pix0 * a.out_ld + n0, ro0 = pix0 * a.res_ld + n0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = it * 8 + prow;
      const int r = it >> 1, c8 = it & 1;              // tile row p >> 4 and column half (p & 15) >> 3 of this store
      float4 v = *(const float4*)(tw + p * 36 + f8 * 4);
      // (oy / ox bounds: odd scattered extents -- the input gradient of a stride-2 conv)
      const bool rowok = tyb + r < a.tile_h && oyb + r * a.out_sy < a.out_h;
      if (nok && rowok && (c8 ? okx1 : okx0)) {
        const int d = r * rs_pix + c8 * 8 * a.out_sx;  // uniform pixel step from (row 0, even column half)
        const float nz = nzv[it];
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RES) rv = st_ld4<IOS>(rb, ro0 + d * a.res_ld);
        float o[4] = {v.x + bv.x + nz, v.y + bv.y + nz, v.z + bv.z + nz, v.w + bv.w + nz};
        const float r4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float t = o[c];
          if (RES == 1) t += r4[c];
          if (ACT == PPST_ACT_LRELU) t = (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
          else if (ACT == PPST_ACT_PRELU) t = t >= 0.f ? t : t * slope;
          if (RES == 2) t += r4[c];
          o[c] = t * a.out_scale;
        }
        if (IOS == PPST_ST_F32) PPST_EPI_STORE((float*)yb + (yo0 + d * a.out_ld), o);
        else st_st4<IOS>(yb, yo0 + d * a.out_ld, make_float4(o[0], o[1], o[2], o[3]));
        s1.x += o[0]; s1.y += o[1]; s1.z += o[2]; s1.w += o[3];
        s2.x += o[0] * o[0]; s2.y += o[1] * o[1]; s2.z += o[2] * o[2]; s2.w += o[3] * o[3];
      }
    }
    if (a.stats) {
      // sum over the 8 pixel rows of the wave (lanes with equal f8): xor 8, 16, 32
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
      }
      if (prow == 0) {
        float* r = red + (wm * BN + nl0) * 2;
        r[0] = s1.x; r[1] = s2.x; r[2] = s1.y; r[3] = s2.y; r[4] = s1.z; r[5] = s2.z; r[6] = s1.w; r[7] = s2.w;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();  // tile reads done before the next pass overwrites it
  }
  };
  {
    const int resm = a.residual ? (res_after ? 2 : 1) : 0;
#define EPI_GO(A_)                                                                                    \
  do {                                                                                                \
    if (resm == 0) epi_passes(EpiC<A_>{}, EpiC<0>{});                                                 \
    else if (resm == 1) epi_passes(EpiC<A_>{}, EpiC<1>{});                                            \
    else epi_passes(EpiC<A_>{}, EpiC<2>{});                                                           \
  } while (0)
    if (!X3) epi_passes(EpiR{act}, EpiR{resm});        // reduced-precision kernels: one generic instance
    else if (act == PPST_ACT_LRELU) EPI_GO(PPST_ACT_LRELU);
    else if (act == PPST_ACT_PRELU) EPI_GO(PPST_ACT_PRELU);
    else EPI_GO(PPST_ACT_NONE);
#undef EPI_GO
  }
  if (a.stats) {
    __syncthreads();
    const int tiles = a.tiles_y * a.tiles_x;
    for (int nl = tid; nl < BN; nl += NT) {
      int n = ntile * BN + nl;
      if (n < a.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += red[(w * BN + nl) * 2]; s2 += red[(w * BN + nl) * 2 + 1]; }
        float* o = a.stats + ((((int64_t)b * a.n_groups + group) * tiles + tyi * a.tiles_x + txi) * a.cout + n) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
#ifdef PPST_CONV_TRACE
  if (a.dbg && lane == 0 && wave == 0) {   // in-kernel clock: shader cycles / 100 MHz reference ticks over the whole block
    unsigned long long c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    unsigned long long* o = a.dbg + (int64_t)TR_BLOCKS * (NT / 64) * TR_STEPS * 8 + (int64_t)blockIdx.x * 2;
    o[0] = c1 - tr_c0;
    o[1] = r1 - tr_r0;
    // absolute start stamp of the traced blocks (prologue = first step's stamp - this), behind the per-block pairs
    if (blockIdx.x < TR_BLOCKS) a.dbg[(int64_t)TR_BLOCKS * (NT / 64) * TR_STEPS * 8 + (int64_t)gridDim.x * 2 + blockIdx.x] = tr_c0;
  }
#endif
#ifdef PPST_CONV_CLOCK
  // clock-only diagnostic build: two stamps per block (no per-step work), written to the buffer registered with
  // ppst_conv_clock_buffer(); lets the in-kernel clock be sampled inside the real benchmark loop
  if (g_clock_buf && lane == 0 && wave == 0) {
    unsigned long long c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    unsigned long long* o = g_clock_buf + (int64_t)(blockIdx.x % g_clock_n) * 2;
    o[0] = c1 - tr_c0;
    o[1] = r1 - tr_r0;
  }
#endif
}

// fp32 NHWC -> pre-split layout (experiment): per pixel and 8-channel group [hi x 8 | lo x 8] bf16 in the group's own 32 bytes
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ x, unsigned* __restrict__ y, int64_t npix, int C, int x_ld,
                                                       int y_ld) {
  const int groups = C >> 3;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < npix * groups; t += (int64_t)gridDim.x * 256) {
    const int64_t p = t / groups;
    const int g8 = (int)(t - p * groups);
    const float* src = x + p * x_ld + g8 * 8;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned short h0, l0, h1, l1;
      split_bf16(src[2 * q], h0, l0);
      split_bf16(src[2 * q + 1], h1, l1);
      hi[q] = (unsigned)h0 | ((unsigned)h1 << 16);
      lo[q] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    unsigned* o = y + p * y_ld + g8 * 8;
    *(uint4*)o = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *(uint4*)(o + 4) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}
extern "C" int ppst_presplit(const void* x, void* y, int64_t npix, int C, int x_ld, int y_ld, void* stream) {
  if (npix < 0 || C <= 0 || C % 8 || x_ld < C || y_ld < C || x_ld % 4 || y_ld % 4) return PPST_EINVAL;
  if (npix == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t blocks = cdiv64(npix * (C >> 3), 256);
  if (blocks > 65536) blocks = 65536;
  PPST_LAUNCH(presplit_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)x, (unsigned*)y, npix, C, x_ld, y_ld);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------ weight packing --
// out[group][ntile][step][hilo][g][n_local][j] = split_bf16(scale * w[n][src_c+8g+j][ky][kx])
struct PackJob {            // == ppst_pack_job of include/ppst_hip.h
  const float* w;
  int64_t sn, sc, sy, sx;
  const int* src_c;
  const int* src_ky;
  const int* src_kx;
  unsigned short* out;
  int64_t total;            // n_groups * n_tiles * nsteps * 4 * bn items (one item = 8 k-values of one output channel)
  int64_t block0;           // first block of this job inside a batched launch
  float scale;
  int cout, bn, nsteps, n_groups, x3, f16, nblocks;
  int dual;                 // the N tile is [column phase 0 | column phase 1] x bn / 2 channels: src_kx = kx0 | kx1 << 8
};
__device__ __forceinline__ void pack_item(const PackJob& j, int64_t t) {
  const float* __restrict__ w = j.w;
  const int bn = j.bn, nsteps = j.nsteps, cout = j.cout;
  const int bnc = j.dual ? bn / 2 : bn;            // channels per N tile
  const int n_tiles = (cout + bnc - 1) / bnc;
  const int npl = j.x3 ? 8 : 4;
  int nl = (int)(t % bn);
  int64_t r = t / bn;
  int g = (int)(r % 4); r /= 4;
  int s = (int)(r % nsteps); r /= nsteps;
  int ntile = (int)(r % n_tiles);
  int group = (int)(r / n_tiles);
  int n = ntile * bnc + (nl % bnc);
  int gs = group * nsteps + s;
  int c0 = j.src_c[gs] + 8 * g, ky = j.src_ky[gs], kx = j.src_kx[gs];
  if (j.dual) kx = nl < bnc ? (kx & 0xff) : (kx >> 8);
  unsigned short hi[8], lo[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float v = 0.f;
    if (n < cout && j.src_c[gs] >= 0) v = w[n * j.sn + (int64_t)(c0 + q) * j.sc + ky * j.sy + kx * j.sx] * j.scale;  // src_c < 0: zero-weight pad step
    if (j.x3 == 2) {      // K64: one rounding to the operand type; the "lo" planes hold channels 32-63 of the step
      float v2 = 0.f;
      if (n < cout && j.src_c[gs] >= 0) v2 = w[n * j.sn + (int64_t)(c0 + 32 + q) * j.sc + ky * j.sy + kx * j.sx] * j.scale;
      hi[q] = j.f16 ? __builtin_bit_cast(unsigned short, (_Float16)v) : f2bf(v);
      lo[q] = j.f16 ? __builtin_bit_cast(unsigned short, (_Float16)v2) : f2bf(v2);
    } else if (j.f16) { hi[q] = __builtin_bit_cast(unsigned short, (_Float16)v); lo[q] = 0; }
    else split_bf16(v, hi[q], lo[q]);
  }
  int64_t blob = (((int64_t)group * n_tiles + ntile) * nsteps + s) * ((int64_t)npl * bn * 8);
  unsigned short* oh = j.out + blob + ((int64_t)g * bn + nl) * 8;
#pragma unroll
  for (int q = 0; q < 8; ++q) oh[q] = hi[q];
  if (j.x3) {
    unsigned short* ol = oh + (int64_t)4 * bn * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) ol[q] = lo[q];
  }
}
__global__ __launch_bounds__(256) void conv_pack_kernel(PackJob j) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < j.total; t += (int64_t)gridDim.x * 256) pack_item(j, t);
}
// All plans of a network in ONE launch (after every Adam step: 216 single launches per train step before): block b belongs to the
// job whose [block0, block0 + nblocks) range holds it (binary search over the table, a handful of scalar loads).
__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const PackJob j = jobs[lo];
  for (int64_t t = (b - j.block0) * 256 + threadIdx.x; t < j.total; t += (int64_t)j.nblocks * 256) pack_item(j, t);
}

static int pack_blocks(int64_t total) {
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  return (int)blocks;
}
extern "C" int ppst_conv_pack(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int bn,
                              const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                              int precision, void* out, void* stream) {
  if (cout <= 0 || (bn != 64 && bn != 128 && bn != 256) || nsteps <= 0 || (n_groups != 1 && n_groups != 4) ||
      (precision != 0 && precision != 1 && precision != 3 && precision != 4))
    return PPST_EINVAL;
  if (!w || !src_c || !src_ky || !src_kx || !out) return PPST_ENULL;
  int n_tiles = (cout + bn - 1) / bn;
  PackJob j;
  j.w = (const float*)w; j.sn = sn; j.sc = sc; j.sy = sy; j.sx = sx; j.src_c = src_c; j.src_ky = src_ky; j.src_kx = src_kx;
  j.out = (unsigned short*)out; j.total = (int64_t)n_groups * n_tiles * nsteps * 4 * bn; j.block0 = 0; j.scale = scale;
  j.cout = cout; j.bn = bn; j.nsteps = nsteps; j.n_groups = n_groups; j.x3 = precision == 0 ? 1 : 0;
  j.f16 = (precision == 3 || precision == 4) ? 1 : 0;
  j.dual = 0;
  j.nblocks = pack_blocks(j.total);
  PPST_LAUNCH(conv_pack_kernel, dim3((unsigned)j.nblocks), dim3(256), 0, as_stream(stream), j);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_conv_pack_dual(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout,
                                   const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                                   int precision, void* out, void* stream) {
  if (cout <= 0 || nsteps <= 0 || n_groups != 2 || (precision != 0 && precision != 1 && precision != 3)) return PPST_EINVAL;
  if (!w || !src_c || !src_ky || !src_kx || !out) return PPST_ENULL;
  PackJob j;
  j.w = (const float*)w; j.sn = sn; j.sc = sc; j.sy = sy; j.sx = sx; j.src_c = src_c; j.src_ky = src_ky; j.src_kx = src_kx;
  j.out = (unsigned short*)out; j.total = (int64_t)n_groups * cdiv(cout, 128) * nsteps * 4 * 256; j.block0 = 0; j.scale = scale;
  j.cout = cout; j.bn = 256; j.nsteps = nsteps; j.n_groups = n_groups; j.x3 = precision == 0 ? 1 : 0; j.f16 = precision == 3 ? 1 : 0; j.dual = 1;
  j.nblocks = pack_blocks(j.total);
  PPST_LAUNCH(conv_pack_kernel, dim3((unsigned)j.nblocks), dim3(256), 0, as_stream(stream), j);
  return PPST_LAUNCH_CHECK();
}
// Weights for ppst_conv_args.variant 11 (conv_mfma2.hip UP9): blob of step s = 4 * chunk + shift in the N-256 kernel's LDS image
// [hi | lo][k-group g][256 columns][8 k], column = u type t * 64 + N-wave wn * 16 + r <-> output channel 64 * ntile + 16 * wn + r; the
// tap of (shift, type) -- shift (0,0): ee w[0][0], eo w[0][1], oe w[1][0], oo w[1][1]; (-1,0): ee w[2][0], eo w[2][1]; (0,-1): ee
// w[0][2], oe w[1][2]; (-1,-1): ee w[2][2] -- of the UN-BLURRED 3x3 kernel w (Cout, Cin, 3, 3) * scale; zeros elsewhere.
__global__ __launch_bounds__(256) void conv_pack_up9_kernel(const float* __restrict__ w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale,
                                                            int cout, int cin, unsigned short* __restrict__ out, int64_t total) {
  const int nsteps = (cin / 32) * 4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int col = (int)(t % 256);
    int64_t r = t / 256;
    const int g = (int)(r % 4); r /= 4;
    const int s = (int)(r % nsteps);
    const int ntile = (int)(r / nsteps);
    const int chunk = s >> 2, sh = s & 3, ty = col >> 6, wn = (col >> 4) & 3, n = ntile * 64 + wn * 16 + (col & 15);   // columns: [type][N-wave][16 ch]
    // (ky, kx) of (shift, type), -1: no tap
    const int kyt[4][4] = {{0, 0, 1, 1}, {2, 2, -1, -1}, {0, -1, 1, -1}, {2, -1, -1, -1}};
    const int kxt[4][4] = {{0, 1, 0, 1}, {0, 1, -1, -1}, {2, -1, 2, -1}, {2, -1, -1, -1}};
    const int ky = kyt[sh][ty], kx = kxt[sh][ty];
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = 0.f;
      if (ky >= 0 && n < cout) v = w[n * sn + (int64_t)(chunk * 32 + g * 8 + q) * sc + ky * sy + kx * sx] * scale;
      split_bf16(v, hi[q], lo[q]);
    }
    unsigned short* oh = out + (((int64_t)ntile * nsteps + s) * 8 * 256 + (int64_t)g * 256 + col) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) { oh[q] = hi[q]; oh[(int64_t)4 * 256 * 8 + q] = lo[q]; }
  }
}
extern "C" int64_t ppst_conv_pack_up9_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cout % 64 || cin % 32) return 0;
  return (int64_t)(cout / 64) * (cin / 32) * 4 * 8 * 256 * 16;
}
extern "C" int ppst_conv_pack_up9(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int cin, void* out,
                                  void* stream) {
  if (cout <= 0 || cin <= 0 || cout % 64 || cin % 32) return PPST_EINVAL;
  if (!w || !out) return PPST_ENULL;
  const int64_t total = (int64_t)(cout / 64) * (cin / 32) * 4 * 4 * 256;
  PPST_LAUNCH(conv_pack_up9_kernel, dim3((unsigned)pack_blocks(total)), dim3(256), 0, as_stream(stream), (const float*)w, sn, sc, sy, sx, scale,
              cout, cin, (unsigned short*)out, total);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_conv_pack_k64(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int bn,
                                  const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                                  int precision, int dual, void* out, void* stream) {
  if (cout <= 0 || nsteps <= 0 || (precision != 1 && precision != 3) || (dual ? (bn != 256 || n_groups != 2) : ((bn != 128 && bn != 256) || (n_groups != 1 && n_groups != 4))))
    return PPST_EINVAL;
  if (!w || !src_c || !src_ky || !src_kx || !out) return PPST_ENULL;
  PackJob j;
  j.w = (const float*)w; j.sn = sn; j.sc = sc; j.sy = sy; j.sx = sx; j.src_c = src_c; j.src_ky = src_ky; j.src_kx = src_kx;
  j.out = (unsigned short*)out; j.total = (int64_t)n_groups * cdiv(cout, dual ? bn / 2 : bn) * nsteps * 4 * bn; j.block0 = 0; j.scale = scale;
  j.cout = cout; j.bn = bn; j.nsteps = nsteps; j.n_groups = n_groups; j.x3 = 2; j.f16 = precision == 3 ? 1 : 0; j.dual = dual ? 1 : 0;
  j.nblocks = pack_blocks(j.total);
  PPST_LAUNCH(conv_pack_kernel, dim3((unsigned)j.nblocks), dim3(256), 0, as_stream(stream), j);
  return PPST_LAUNCH_CHECK();
}
// jobs: DEVICE array of ppst_pack_job (block0 / nblocks filled by the caller: consecutive ranges, nblocks = ppst_pack_job_blocks(total))
extern "C" int ppst_pack_job_blocks(int64_t total) { return total > 0 ? pack_blocks(total) : 0; }
extern "C" int ppst_conv_pack_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
  static_assert(sizeof(PackJob) == sizeof(ppst_pack_job), "ppst_pack_job layout");
  if (njobs < 0 || total_blocks < 0) return PPST_EINVAL;
  if (njobs == 0 || total_blocks == 0) return PPST_OK;
  if (!jobs) return PPST_ENULL;
  PPST_LAUNCH(conv_pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), (const PackJob*)jobs, njobs);
  return PPST_LAUNCH_CHECK();
}

// EqualizedConv2d fused-upscale weight (stylegan2_layers.py:314-319):
// w (Cout,Cin,3,3) -> out (Cin,Cout,4,4) = sum of the 4 unit shifts of the zero-padded kernel
struct UpscaleJob {         // == ppst_upscale_job
  const float* w;
  float* out;
  int64_t total, block0;
  float scale;
  int cout, cin, nblocks;
};
__device__ __forceinline__ void upscale_item(const UpscaleJob& j, int64_t t) {
  int kx = (int)(t & 3), ky = (int)((t >> 2) & 3);
  int64_t r = t >> 4;
  int n = (int)(r % j.cout), c = (int)(r / j.cout);
  const float* wp = j.w + ((int64_t)n * j.cin + c) * 9;
  const float scale = j.scale;
  auto at = [&](int y, int x) -> float { return (y >= 0 && y < 3 && x >= 0 && x < 3) ? wp[y * 3 + x] * scale : 0.f; };
  // padded p[y][x] = w[y-1][x-1]; out[ky][kx] = p[ky+1][kx+1] + p[ky][kx+1] + p[ky+1][kx] + p[ky][kx]
  j.out[t] = at(ky, kx) + at(ky - 1, kx) + at(ky, kx - 1) + at(ky - 1, kx - 1);
}
__global__ __launch_bounds__(256) void upscale_weight_kernel(UpscaleJob j) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < j.total; t += (int64_t)gridDim.x * 256) upscale_item(j, t);
}
__global__ __launch_bounds__(256) void upscale_weight_batch_kernel(const UpscaleJob* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const UpscaleJob j = jobs[lo];
  for (int64_t t = (b - j.block0) * 256 + threadIdx.x; t < j.total; t += (int64_t)j.nblocks * 256) upscale_item(j, t);
}
extern "C" int ppst_upscale_weight(const void* w, void* out, int cout, int cin, float scale, void* stream) {
  if (cout <= 0 || cin <= 0) return PPST_EINVAL;
  if (!w || !out) return PPST_ENULL;
  UpscaleJob j;
  j.w = (const float*)w; j.out = (float*)out; j.total = (int64_t)cin * cout * 16; j.block0 = 0; j.scale = scale; j.cout = cout; j.cin = cin;
  j.nblocks = pack_blocks(j.total);
  PPST_LAUNCH(upscale_weight_kernel, dim3((unsigned)j.nblocks), dim3(256), 0, as_stream(stream), j);
  return PPST_LAUNCH_CHECK();
}
// Input gradient of the stride-2 3x3 conv as ONE stride-1 conv with 2 x 2 taps whose output channels stack the four output phases
// (round 5, ops kind "dgrad_s2ds"): out[(py*2+px)*cin + n][c][ty][tx] = w[c][n][ky][kx] where a phase uses tap offset t = 0 with
// k = (p == 0 ? 0 : 1) and t = 1 (one input position back) with k = 2 for p == 0 only; 0 elsewhere.  w: the forward (cout, cin, 3, 3).
__global__ __launch_bounds__(256) void dgrad_s2d_stack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                                                     int64_t total) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int tx = (int)(t & 1), ty = (int)((t >> 1) & 1);
    int64_t r = t >> 2;
    const int c = (int)(r % cout); r /= cout;
    const int n = (int)(r % cin), g = (int)(r / cin);
    const int py = g >> 1, px = g & 1;
    const int ky = ty == 0 ? (py == 0 ? 0 : 1) : (py == 0 ? 2 : -1), kx = tx == 0 ? (px == 0 ? 0 : 1) : (px == 0 ? 2 : -1);
    out[t] = (ky >= 0 && kx >= 0) ? w[((int64_t)c * cin + n) * 9 + ky * 3 + kx] : 0.f;
  }
}
extern "C" int ppst_dgrad_s2d_stack_weight(const void* w, void* out, int cout, int cin, void* stream) {
  if (cout <= 0 || cin <= 0) return PPST_EINVAL;
  if (!w || !out) return PPST_ENULL;
  const int64_t total = (int64_t)4 * cin * cout * 4;
  PPST_LAUNCH(dgrad_s2d_stack_weight_kernel, dim3((unsigned)pack_blocks(total)), dim3(256), 0, as_stream(stream), (const float*)w, (float*)out,
              cout, cin, total);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_upscale_weight_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
  static_assert(sizeof(UpscaleJob) == sizeof(ppst_upscale_job), "ppst_upscale_job layout");
  if (njobs < 0 || total_blocks < 0) return PPST_EINVAL;
  if (njobs == 0 || total_blocks == 0) return PPST_OK;
  if (!jobs) return PPST_ENULL;
  PPST_LAUNCH(upscale_weight_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), (const UpscaleJob*)jobs, njobs);
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------------------- profiling --
#define PROF_MAX 16384
static int g_prof_on = 0;
static hipEvent_t g_ev[PROF_MAX][2];
static double g_flop[PROF_MAX];
static int g_info[PROF_MAX][8];
static int g_ev_made = 0, g_ev_used = 0;
static int g_prof_dropped = 0;   // launches not bracketed because PROF_MAX events were in use
extern "C" int ppst_prof_dropped(void) { return g_prof_dropped; }

// bracket one launch on ``st`` (used by ppst_conv2d_mfma here and by the weight-gradient entry points of train.hip): returns the event
// slot or -1 when profiling is off / the pool is full
int ppst_prof_begin_(double flop, const int* info8, hipStream_t st) {
  if (!g_prof_on) return -1;
  if (g_ev_used >= PROF_MAX) { ++g_prof_dropped; return -1; }
  while (g_ev_made <= g_ev_used) {
    if (hipEventCreate(&g_ev[g_ev_made][0]) != hipSuccess || hipEventCreate(&g_ev[g_ev_made][1]) != hipSuccess) return -1;
    ++g_ev_made;
  }
  const int slot = g_ev_used++;
  g_flop[slot] = flop;
  for (int i = 0; i < 8; ++i) g_info[slot][i] = info8[i];
  (void)hipEventRecord(g_ev[slot][0], st);
  return slot;
}
void ppst_prof_end_(int slot, hipStream_t st) {
  if (slot >= 0) (void)hipEventRecord(g_ev[slot][1], st);
}

extern "C" int ppst_prof_enable(int on) {
  g_prof_on = on;
  g_ev_used = 0;
  if (on) g_prof_dropped = 0;
  return PPST_OK;
}
extern "C" int ppst_prof_collect(double* ms, int64_t* launches, double* flop) {
  double t = 0.0, f = 0.0;
  for (int i = 0; i < g_ev_used; ++i) {
    float e = 0.f;
    hipError_t err = hipEventElapsedTime(&e, g_ev[i][0], g_ev[i][1]);
    if (err != hipSuccess) return (int)err;
    t += e;
    f += g_flop[i];
  }
  if (ms) *ms = t;
  if (launches) *launches = g_ev_used;
  if (flop) *flop = f;
  g_ev_used = 0;
  return PPST_OK;
}
// per-launch detail (call before ppst_prof_collect): info = {B, tile_h, tile_w, cin_steps(nsteps), cout, n_groups, halo,
// bn | variant << 12 | k64 << 20 | precision << 24}; a weight-gradient launch has bn = 0
extern "C" int ppst_prof_detail(int idx, double* ms, double* flop, int32_t* info) {
  if (idx < 0 || idx >= g_ev_used) return PPST_EINVAL;
  float e = 0.f;
  hipError_t err = hipEventElapsedTime(&e, g_ev[idx][0], g_ev[idx][1]);
  if (err != hipSuccess) return (int)err;
  if (ms) *ms = e;
  if (flop) *flop = g_flop[idx];
  if (info) for (int i = 0; i < 8; ++i) info[i] = g_info[idx][i];
  return PPST_OK;
}

int ppst_conv2d_mfma2_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st);   // conv_mfma2.hip
int ppst_conv1x1_stream_launch(const ppst_conv_args* a, int n_tiles, int tiles, hipStream_t st);                     // conv1x1.hip
int ppst_conv_direct_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st);         // conv1x1.hip
int ppst_conv3x3_direct_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st);      // conv1x1.hip
int ppst_conv_wino_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st);            // conv_wino.hip
#ifdef PPST_EXPERIMENTS
int ppst_conv_ksplit_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st);         // conv_ksplit.hip
#endif
// The measured-and-off kernel forms (variants 1 / 3 / 7 / 8 / 9, the two-pass fp16 mode, the 8-row two-block tile) are compiled
// only into a PPST_EXPERIMENTS=1 build (python -m ppst_amd.build with that variable set); the production library rejects them.
extern "C" int ppst_has_experiments(void) {
#ifdef PPST_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// ---- state of the across-block K split (common.h): one scratch + flag buffer per (device, stream) that has used it -- launches on one
// stream are ordered, so a buffer is never shared by two launches in flight; the epoch makes a flag of an earlier launch stale
// without a fill between launches.  Allocated at first use (64 MB + 16 KB per pair), never freed.
struct KsState { int dev; hipStream_t st; float* scratch; unsigned* flags; unsigned epoch; };
static KsState g_ks[16];
static int g_ks_n = 0;
int ppst_ksplit_prepare_(int S, const int32_t* starts, int64_t tiles, int nsteps, int acc_regs, int threads, hipStream_t st, KSplitDev* out) {
  if ((S != 2 && S != 4 && S != 8) || (!starts && nsteps % S) || tiles <= 0 || acc_regs <= 0 || acc_regs % 4 || threads <= 0) return PPST_EINVAL;
  for (int i = 0; i <= S; ++i) {
    out->start[i] = starts ? starts[i] : i * (nsteps / S);
    if (i && out->start[i] <= out->start[i - 1]) return PPST_EINVAL;
  }
  if (out->start[0] != 0 || out->start[S] != nsteps) return PPST_EINVAL;
  const int64_t slots = (int64_t)(S - 1) * tiles;
  if (slots > KS_MAX_SLOTS || slots > KS_FLAG_WORDS - 1 || (size_t)slots * threads * acc_regs * 4 > KS_SCRATCH_BYTES) return PPST_EINVAL;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PPST_EINVAL;
  KsState* e = nullptr;
  for (int i = 0; i < g_ks_n; ++i)
    if (g_ks[i].dev == dev && g_ks[i].st == st) e = &g_ks[i];
  if (!e) {
    if (g_ks_n == 16) return PPST_EINVAL;           // (sixteen (device, stream) pairs per process)
    KsState n;
    n.dev = dev; n.st = st; n.epoch = 0; n.scratch = nullptr; n.flags = nullptr;
    if (hipMalloc((void**)&n.scratch, KS_SCRATCH_BYTES) != hipSuccess) return (int)hipGetLastError();
    if (hipMalloc((void**)&n.flags, KS_FLAG_WORDS * 4) != hipSuccess || hipMemset(n.flags, 0, KS_FLAG_WORDS * 4) != hipSuccess) {
      (void)hipFree(n.scratch);
      if (n.flags) (void)hipFree(n.flags);
      return (int)hipGetLastError();
    }
    g_ks[g_ks_n] = n;
    e = &g_ks[g_ks_n++];
  }
  if (++e->epoch == 0) ++e->epoch;            // (0 is the value of a fresh flag)
  out->scratch = e->scratch; out->flags = e->flags; out->epoch = e->epoch; out->S = S;
  return PPST_OK;
}
// 1 if a block of a K-split launch on `stream` ever gave up waiting for its partners (the results of that launch are wrong), 0 if
// none did, < 0 if the stream has no K-split state; resets the marker.  Synchronises the stream.
extern "C" int ppst_conv_ksplit_check(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  for (int i = 0; i < g_ks_n; ++i)
    if (g_ks[i].dev == dev && g_ks[i].st == as_stream(stream)) {
      unsigned v = 0, z = 0;
      if (hipStreamSynchronize(g_ks[i].st) != hipSuccess) return -1;
      if (hipMemcpy(&v, g_ks[i].flags + KS_FLAG_WORDS - 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      if (v) (void)hipMemcpy(g_ks[i].flags + KS_FLAG_WORDS - 1, &z, 4, hipMemcpyHostToDevice);
      return v ? 1 : 0;
    }
  return -1;
}

// (tile_rows 15 = variant 11: blocks of 15 x 15 input positions)
extern "C" int ppst_conv_tiles(int tile_h, int tile_w, int tile_rows) { return cdiv(tile_h, tile_rows) * cdiv(tile_w, tile_rows == 15 ? 15 : 16); }

template <int WM, int WN, int HALO, bool X3, int NAS = 0, bool F16 = false, bool X2 = false, int IOS = PPST_ST_F32>
static void launch_conv(const ConvKArgs& k, int blocks, hipStream_t st) {
  const dim3 grid(blocks, k.ks.S > 1 ? k.ks.S : 1);
  if (k.in_ss) PPST_LAUNCH((conv_mfma_kernel<WM, WN, HALO, X3, true, NAS, F16, X2, false, IOS>), grid, dim3(64 * WM * WN), 0, st, k);
  else PPST_LAUNCH((conv_mfma_kernel<WM, WN, HALO, X3, false, NAS, F16, X2, false, IOS>), grid, dim3(64 * WM * WN), 0, st, k);
}

extern "C" int ppst_conv2d_mfma(const ppst_conv_args* a, void* stream) {
  if (!a) return PPST_ENULL;
  if (a->B < 0 || a->in_h <= 0 || a->in_w <= 0 || a->in_ld <= 0 || a->in_ld % 4 || a->out_h <= 0 || a->out_w <= 0 ||
      a->out_ld < a->cout || a->cout <= 0 || a->cout % 4 || a->out_ld % 4 || (a->residual && a->res_ld % 4) || a->nsteps <= 0 ||
      (a->n_groups != 1 && a->n_groups != 4 && !(a->n_groups == 2 && a->dual_b)) || a->pad_mode < 0 ||
      (a->dual_b && (a->variant != 2 || a->bn != 256 || a->n_groups != 2 || a->halo != 1 || (a->precision != 0 && a->precision != 1 && a->precision != 3) || a->out_sy != 2 || a->out_sx != 2 || !a->early_a)) ||
      a->pad_mode > 2 || a->tile_h <= 0 || a->tile_w <= 0 || a->out_sy <= 0 || a->out_sx <= 0 ||
      (a->precision != 0 && a->precision != 1 && a->precision != 3 && a->precision != 4) || a->halo < 0 || a->halo > 1 || (a->bn != 64 && a->bn != 128 && a->bn != 256) || (a->residual && a->res_ld < a->cout) ||
      a->variant < 0 || a->variant > 11 || (a->variant == 0 && a->bn == 256) ||
      // variant 11 = conv_mfma2.hip UP9: the fused 4x4 stride-2 upscale as the un-blurred 3x3 transposed conv (nine products per input
      // pixel instead of sixteen) + its 2x2 box sum in the epilogue; wpack from ppst_conv_pack_up9, steps = per 32-channel chunk the
      // four input shifts (0,0), (-1,0), (0,-1), (-1,-1) in this order (the CALLER's promise); blocks of 15 x 15 input positions
      (a->variant == 11 && (a->precision != 0 || a->bn != 256 || a->halo != 1 || a->n_groups != 1 || a->out_sy != 2 || a->out_sx != 2 ||
                            !a->early_a || a->tile_rows != 15 || a->pad_mode != PPST_PAD_ZERO || a->in_scale_shift || a->residual ||
                            (a->act & 0xff) == PPST_ACT_PRELU || a->cout % 64 || a->nsteps % 4 || a->in_off_y || a->in_off_x ||
                            a->tile_h != a->in_h || a->tile_w != a->in_w || a->out_h != 2 * a->in_h || a->out_w != 2 * a->in_w || a->io_st)) ||
      // variant 10 = conv_wino.hip: Winograd F(2,3) along x for plain 3x3 stride-1 tables (the CALLER promises the (chunk, dy, dx)
      // step order, as with variant 6, and a wpack from ppst_conv_pack_wino); bf16x3, bn 128, 16-row tiles, one group, unit strides
      (a->variant == 10 && (a->precision != 0 || a->bn != 128 || a->halo != 1 || a->n_groups != 1 || a->out_sy != 1 || a->out_sx != 1 ||
                            a->in_off_y != 0 || a->in_off_x != 0 || a->tile_h != a->out_h || a->tile_w != a->out_w ||
                            a->in_h != a->out_h || a->in_w != a->out_w || a->nsteps % 9 != 0 || a->tile_rows != 16 ||
                            // (its LDS copy of the normalise-on-load table holds 32 chunks = 1024 input channels)
                            (a->in_scale_shift && a->nsteps / 9 > 32))) ||
      ((a->variant >= 1 && a->variant <= 3 || a->variant == 7 || a->variant == 9) &&
       ((a->precision != 0 && !((a->variant == 2 || a->variant == 7 || (a->variant == 9 && a->k64)) && (a->precision == 1 || a->precision == 3))) || a->bn == 64 || !a->early_a)) ||
      (a->variant == 7 && (a->bn != 128 || a->tile_rows != 32)) ||
      // variant 9 = conv_mfma2.hip with 6 m-tiles per wave: block tile 24 x 16 px x 128 ch, two activation slots
      (a->variant == 9 && (a->bn != 128 || a->tile_rows != 24)) ||
      // variant 8 = conv_ksplit.hip: two K-groups of four 128 px x 64 ch waves; bn = 128, bf16x3, chunks of >= 2 steps (and the
      // caller's promise: 2-step chunks start at even steps, steps[i].w bit 2 = parity of step i's chunk index)
      (a->variant == 8 && (a->precision != 0 || a->bn != 128 || !a->early_a)) ||
      (a->variant == 2 && a->bn != 256) || (a->variant == 3 && a->bn != 128) ||
      // variant 4 = the 1x1 streaming kernel (conv1x1.hip): all taps (0,0), one group, unit strides, bf16x3, 64-wide blobs
      (a->variant == 4 && ((a->precision != 0 && a->precision != 1 && a->precision != 3) || a->bn != 64 || a->halo != 0 || a->n_groups != 1 || a->out_sy != 1 || a->out_sx != 1 ||
                           a->in_off_y != 0 || a->in_off_x != 0 || a->tile_h != a->out_h || a->tile_w != a->out_w ||
                           a->in_h != a->out_h || a->in_w != a->out_w)) ||
      // in_res: the 1x1 streaming kernel only, beside a normalise-on-load table, fp32-class mode
      (a->in_res && (a->variant != 4 || !a->in_scale_shift || a->precision != 0 || a->io_st || a->in_res_ld < a->in_c || a->in_res_ld % 4 ||
                     ((uintptr_t)a->in_res % 16))) ||
      // variant 5 = the direct form of the same kernel for thin layers with taps (one group, unit output stride)
      // (6: its register-reuse form for plain 3x3 stride-1 tables: the CALLER promises the (chunk, dy, dx) step order)
      ((a->variant == 5 || a->variant == 6) && ((a->precision != 0 && a->precision != 1 && a->precision != 3) ||
                                                (a->bn != 64 && !(a->variant == 6 && a->bn == 128 && a->precision == 0)) || a->n_groups != 1 || a->out_sy != 1 || a->out_sx != 1 ||
                           a->in_off_y != 0 || a->in_off_x != 0 || a->tile_h != a->out_h || a->tile_w != a->out_w)) ||
      // tile_rows 8 = two 4-wave blocks per CU (8 x 16 px x 128 ch, two activation slots): variant 0, bn 128, halo 1, early_a, bf16x3
      (a->tile_rows != 16 && !(a->variant == 7 && a->tile_rows == 32) && !(a->variant == 9 && a->tile_rows == 24) && !(a->variant == 11 && a->tile_rows == 15) &&
       !(a->variant == 0 && a->tile_rows == 8 && a->bn == 128 && a->halo == 1 && a->early_a && a->precision == 0)) || (a->in_scale_shift && a->in_c <= 0) || a->a_slots < 0 || a->a_slots > 3)
    return PPST_EINVAL;
#ifndef PPST_EXPERIMENTS
  if (a->variant == 1 || a->variant == 3 || (a->variant == 7 && a->precision == 0) || a->variant == 8 || (a->variant == 9 && !a->k64) || a->precision == 4 ||
      a->in_presplit)
    return PPST_EINVAL;          // experiment forms: not in this build (variant 7 is a production form in the single-pass modes)
#endif
  // pre-split input (experiment): the 8-wave tile kernel only, chunks of >= 4 steps (the caller's promise with early_a), no
  // normalise-on-load
  // half-precision activation storage (x, residual, y): the single-pass modes, in the operand type of the mode (1 -> bfloat16,
  // 3 -> IEEE half); kernel families 0 (16-row tiles), 2 (N-256), 4 / 5 / 6 (streaming 1x1 / direct)
  if (a->io_st && (a->io_st != (a->precision == 3 ? PPST_ST_F16 : a->precision == 1 ? PPST_ST_BF16 : -1) ||
                   (a->tile_rows != 16 && a->variant != 7 && !(a->variant == 9 && a->k64)) || a->in_presplit ||
                   !(a->variant == 0 || a->variant == 2 || a->variant == 4 || a->variant == 5 || a->variant == 6 || a->variant == 7 || (a->variant == 9 && a->k64))))
    return PPST_EINVAL;
  // 64 input channels per step (conv_mfma2.hip K64): single-pass modes on half-stored activations, the N-256 geometry (plain or phase
  // pairs) or the 24 x 16 px x 128 ch tile
  if (a->k64 && (!a->io_st || !a->halo || !((a->variant == 2 && a->bn == 256) || (a->variant == 9 && a->bn == 128 && a->tile_rows == 24)) || !a->early_a))
    return PPST_EINVAL;
  if (a->in_presplit && (a->variant != 0 || a->bn != 128 || a->halo != 1 || a->precision != 0 || !a->early_a || a->tile_rows != 16 ||
                         a->in_scale_shift))
    return PPST_EINVAL;
  // the epilogues address one image with 32-bit element offsets
  // (+ one tile row of slack: lanes beyond the image edge form their offset too, and only then mask the access)
  {
    const int64_t px = (int64_t)a->out_h * a->out_w + 64 * (int64_t)a->out_sx;
    if (px * a->out_ld > 0x7fffffff || (a->residual && px * a->res_ld > 0x7fffffff)) return PPST_EINVAL;
  }
  // the input side: every kernel family addresses one input image with 32-bit BYTE offsets (buffer loads / int offsets)
  if ((int64_t)a->in_h * a->in_w * a->in_ld * 4 > 0x7fffffff) return PPST_EINVAL;
  if (a->io_st && (((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->residual) % 8)) return PPST_EINVAL;
  // (the streaming kernels read a lane's 8 channels as one 16-byte item)
  if (a->io_st && a->variant >= 4 && a->variant <= 6 && (a->in_ld % 8 || (uintptr_t)a->x % 16)) return PPST_EINVAL;
  // the scattered output must reach into the output tensor (elements beyond it are dropped)
  if ((a->tile_h - 1) * a->out_sy >= a->out_h || (a->tile_w - 1) * a->out_sx >= a->out_w) return PPST_EINVAL;
  if (a->B == 0) return PPST_OK;
  if (!a->x || !a->wpack || !a->steps || !a->y) return PPST_ENULL;
  ConvKArgs k;
  k.x = (const float*)a->x; k.wpack = (const unsigned short*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.in_h = a->in_h; k.in_w = a->in_w; k.in_ld = a->in_ld; k.out_h = a->out_h; k.out_w = a->out_w;
  k.out_ld = a->out_ld; k.cout = a->cout; k.nsteps = a->nsteps; k.n_groups = a->n_groups; k.pad_mode = a->pad_mode;
  k.in_off_y = a->in_off_y; k.in_off_x = a->in_off_x; k.out_sy = a->out_sy; k.out_sx = a->out_sx; k.act = a->act;
  k.res_ld = a->res_ld; k.tile_h = a->tile_h; k.tile_w = a->tile_w;
  k.tiles_y = cdiv(a->tile_h, a->tile_rows); k.tiles_x = cdiv(a->tile_w, a->variant == 11 ? 15 : 16);
  k.n_tiles = cdiv(a->cout, a->variant == 11 ? 64 : a->dual_b ? a->bn / 2 : a->bn);
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu;
  k.in_c = a->in_c; k.in_act = a->in_act;
  k.early_a = a->early_a ? 1 : 0;
  k.dbg = nullptr;
  k.ks.scratch = nullptr; k.ks.flags = nullptr; k.ks.epoch = 0; k.ks.S = 1;
#ifdef PPST_CONV_TRACE
  k.dbg = (unsigned long long*)a->prelu;  // diagnostic builds: the (unused) prelu slot carries the debug buffer
  k.prelu = nullptr;
#endif
  int64_t blocks64 = (int64_t)a->n_groups * k.n_tiles * a->B * k.tiles_y * k.tiles_x;
  if (blocks64 > 0x7fffffff) return PPST_EINVAL;
  int blocks = (int)blocks64;
  hipStream_t st = as_stream(stream);
  if (a->ksplit > 1) {
    // the tile kernel (variant 0, 16-row tiles); the N-256 and Winograd kernels take theirs in their own launchers
    if (a->variant == 0) {
      if ((a->tile_rows != 16 && a->tile_rows != 8) || a->in_presplit) return PPST_EINVAL;
      const int e0 = ppst_ksplit_prepare_(a->ksplit, a->ksplit_starts, blocks, a->nsteps, 64, (a->bn == 128 && a->tile_rows == 16) ? 512 : 256, st, &k.ks);
      if (e0 != PPST_OK) return e0;
    } else if (a->variant != 2 && a->variant != 10) return PPST_EINVAL;
  }
  k.prefetch_w = (int64_t)blocks * k.ks.S <= 512 ? 1 : 0;
  int slot = -1;
  if (g_prof_on) {
    // info[7]: bits 0-11 the N tile, 12-19 the kernel variant that runs the launch, 20 k64, 24-27 the precision mode (bench.py derives
    // the MFMA flop the pipe ISSUES from them: Winograd 2/3, nine-product upscale 9/16 over 16 x 16-position blocks, passes per mode)
    const int inf[8] = {a->B, a->tile_h, a->tile_w, a->nsteps, a->cout, a->n_groups, a->halo,
                        a->bn | (a->variant << 12) | ((a->k64 ? 1 : 0) << 20) | (a->precision << 24)};
    slot = ppst_prof_begin_(2.0 * 32.0 * (a->flop_steps > 0 ? a->flop_steps : a->nsteps) * (double)((a->dual_b || a->variant == 11) ? 4 : a->n_groups) * a->cout * (double)a->B * a->tile_h * a->tile_w,
                            inf, st);
  }
  if (a->variant >= 1) {
    int e2 = a->variant == 4   ? ppst_conv1x1_stream_launch(a, k.n_tiles, k.tiles_y * k.tiles_x, st)
             : a->variant == 5 ? ppst_conv_direct_launch(a, k.n_tiles, k.tiles_y, k.tiles_x, st)
             : a->variant == 6 ? ppst_conv3x3_direct_launch(a, k.n_tiles, k.tiles_y, k.tiles_x, st)
             : a->variant == 10 ? ppst_conv_wino_launch(a, k.n_tiles, k.tiles_y, k.tiles_x, st)
#ifdef PPST_EXPERIMENTS
             : a->variant == 8 ? ppst_conv_ksplit_launch(a, k.n_tiles, k.tiles_y, k.tiles_x, st)
#endif
                               : ppst_conv2d_mfma2_launch(a, k.n_tiles, k.tiles_y, k.tiles_x, st);
    if (slot >= 0) (void)hipEventRecord(g_ev[slot][1], st);
    return e2;
  }
  const bool x3 = a->precision == 0;
  const bool f16 = a->precision == 3, x2 = a->precision == 4, hs = a->io_st != 0;
#ifdef PPST_EXPERIMENTS
#define X2_LAUNCH(WM_, WN_, H_) launch_conv<WM_, WN_, H_, false, 0, true, true>(k, blocks, st)
#else
#define X2_LAUNCH(WM_, WN_, H_) (void)0      /* precision 4 was rejected above */
#endif
#define DISPATCH(WM_, WN_)                                                                                         \
  do {                                                                                                             \
    if (a->halo) { if (x3) launch_conv<WM_, WN_, 1, true>(k, blocks, st); else if (x2) X2_LAUNCH(WM_, WN_, 1); else if (f16 && hs) launch_conv<WM_, WN_, 1, false, 0, true, false, PPST_ST_F16>(k, blocks, st); else if (f16) launch_conv<WM_, WN_, 1, false, 0, true>(k, blocks, st); else if (hs) launch_conv<WM_, WN_, 1, false, 0, false, false, PPST_ST_BF16>(k, blocks, st); else launch_conv<WM_, WN_, 1, false>(k, blocks, st); } \
    else         { if (x3) launch_conv<WM_, WN_, 0, true>(k, blocks, st); else if (x2) X2_LAUNCH(WM_, WN_, 0); else if (f16 && hs) launch_conv<WM_, WN_, 0, false, 0, true, false, PPST_ST_F16>(k, blocks, st); else if (f16) launch_conv<WM_, WN_, 0, false, 0, true>(k, blocks, st); else if (hs) launch_conv<WM_, WN_, 0, false, 0, false, false, PPST_ST_BF16>(k, blocks, st); else launch_conv<WM_, WN_, 0, false>(k, blocks, st); } \
  } while (0)
  // (round 1 measured 8-row tiles -- two 80-KB blocks per CU -- 35 % slower: only one block was ever resident.  Round 3
  //  re-instantiates them at 77 KB with the two-slot ring of the early_a tables: see ops.TWO_BLOCK_8ROW.)
  // small-K layers on the 64-channel tile: a shallower activation ring (59 / 48 / 80 KB of LDS instead of
  // 145 / 112) puts two blocks on a CU, so one block's loads and stores overlap the other's MFMAs
#ifdef PPST_EXPERIMENTS
  if (a->in_presplit) PPST_LAUNCH((conv_mfma_kernel<4, 2, 1, true, false, 0, false, false, true>), dim3(blocks), dim3(512), 0, st, k);
  else
#endif
  // 8 x 16-px tiles, two 4-wave blocks per CU: the same per-pixel MFMA sequence as the 16-row tile (bit-identical outputs); a
  // wash on full grids, but twice the blocks where ONE small image per launch leaves the chip under-filled
  // (64 x 64 layers of the train step at batch 2: 64 blocks for 256 CUs)
  if (a->tile_rows == 8) launch_conv<2, 2, 1, true, 2>(k, blocks, st);
  else
  if (a->bn == 64 && x3 && a->halo == 1 && a->a_slots == 1) launch_conv<4, 1, 1, true, 1>(k, blocks, st);
  else if (a->bn == 64 && x3 && a->halo == 0 && a->a_slots == 1) launch_conv<4, 1, 0, true, 1>(k, blocks, st);
  else if (a->bn == 64 && x3 && a->halo == 0 && a->a_slots == 2) launch_conv<4, 1, 0, true, 2>(k, blocks, st);
  else if (a->bn == 128) DISPATCH(4, 2); else DISPATCH(4, 1);
#undef DISPATCH
#undef X2_LAUNCH
  int e = PPST_LAUNCH_CHECK();
  if (slot >= 0) (void)hipEventRecord(g_ev[slot][1], st);
  return e;
}
