// Fused implicit-GEMM convolution, "one fat wave per SIMD" variant (round 2).
//
// Same operation, step tables, weight blobs' meaning, padding modes, normalise-on-load and epilogue as conv_mfma.hip
// (the StyledConv / EqualConv2d conv of stylegan2_layers.py:184-193, 305-347, 467-475) -- a different decomposition:
//
//   conv_mfma.hip : 512 threads = 8 waves (2 per SIMD), wave tile 64 px x 64 ch  (4 x 4 MFMA tiles, 48 MFMAs / K-step)
//   this file     : 256 threads = 4 waves (1 per SIMD), wave tile 128 px x 16*NT ch (8 x NT tiles, NT = 4 or 8:
//                   96 / 192 MFMAs per wave and K-step), block tile 16x16 px x (128 | 256) channels, up to 512 VGPRs.
//
// Why (profiles/r01_conv_trace.txt): the 8-wave kernel's K-step took ~2450 cycles for the 1536 its
// MFMAs need.  v_mfma_f32_16x16x32_bf16 holds its SIMD's issue port for 8 of its 16 cycles, so a step leaves 768 issue
// cycles for everything else, and the two co-resident waves needed ~2 x 100 non-MFMA instructions (LDS fragment reads,
// address arithmetic, descriptor handling, DMA issue, waits) ~ 800-1000 cycles: issue-bound, not matrix-bound.  Per
// K-step the non-MFMA work of a wave is almost independent of its tile, so doubling / quadrupling the MFMAs per wave
// (and halving the wave count) puts the same overhead beside 2-4x the matrix work: 24 or 32 ds_read_b128 for 96 or
// 192 MFMAs (0.25 / 0.17 per MFMA instead of 0.33).  The 256-channel tile also halves the number of times an
// activation tile is fetched on the 256/512-channel layers (PMC showed reads at 2.0x the algorithmic bytes).
//
// LDS: activation ring of 2 slots (43 KB each; legal because every chunk of the step table spans >= 2 steps -- the
// host only selects this kernel for such tables) + weight ring of 2 slots (16 / 32 KB) = 118 / 151 KB, one block per CU.
#include "common.h"

struct Conv2KArgs {
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
  const float* in_ss;
  const float* in_prelu;
  int in_c, in_act;
  int early_a;
  unsigned long long* dbg;   // -DPPST_CONV_TRACE builds only
  KSplitDev ks;              // across-block K split (common.h): grid row y runs steps [ks.start[y], ks.start[y + 1])
};

// Diagnostic build -DPPST_CONV_TRACE (tests/build_variant.sh): the per-step timeline of conv_mfma.hip's trace, same buffer
// layout [block < 8][wave][step < 160][8]: 0 absolute start; relative: 1 head (descriptor, DMA, activation loads) issued,
// 2 m-tiles 0-3 issued, 3 all m-tiles issued, 4 weight-fragment reload issued, 5 vmcnt wait done, 6 barrier passed; 7 newA2.
#ifdef PPST_CONV_TRACE
#define TR2(i) asm volatile("s_memtime %0" : "=s"(tr_[i])::"memory");
#define TR2_DECL unsigned long long tr_[7];
#define TR2_FLUSH(s, flag)                                                                            \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  if (a.dbg && blockIdx.x < 8 && (s) < 160 && lane == 0) {                                            \
    unsigned long long* o_ = a.dbg + ((((int64_t)blockIdx.x * NWV + wave) * 160) + (s)) * 8;          \
    o_[0] = tr_[0];                                                                                   \
    for (int q_ = 1; q_ < 7; ++q_) o_[q_] = tr_[q_] - tr_[0];                                         \
    o_[7] = (flag) ? 1 : 0;                                                                           \
  }
#else
#define TR2(i)
#define TR2_DECL
#define TR2_FLUSH(s, flag)
#endif

__device__ __forceinline__ int pad_index2(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// WNW: waves along N (2: one wave per SIMD, 256 threads; 4: two waves per SIMD, 512 threads, wave tile 128 px x 64 ch,
// block tile 16x16 px x 256 ch -- the "8-phase template" geometry of the CDNA GEMM guide).  BDB: weight fragments of the
// next step double-buffered in registers (WNW = 2) or reloaded at the end of the step (WNW = 4: 256 registers per wave).
// NA_: activation ring slots.  1 (with WNW = 2, NT = 4, BDB = false): 75 KB of LDS and <= 256 registers, so TWO 4-wave blocks
// share a CU -- the 128 px x 64 ch wave tile for layers whose Cout is only 128.  With one slot a new chunk is stored at the
// END of the step before it is used, between two barriers; the CU's other block fills that bubble.
// WMW: waves along M (8 tile rows each).  4 (with WNW = 2, NT = 4, BDB = false, NA_ = 1): block tile 32 x 16 px x 128 ch --
// the 128 px x 64 ch wave tile, two waves per SIMD, for layers whose Cout is only 128 (variant 7).  Its activation slot is
// 78 KB, so there is one, and a new chunk is stored at the end of the step before it is used (the NA_ = 1 path).
// PREC: 0 the fp32-class bf16 hi+lo form (three MFMAs per product); 1 / 3 single-pass bf16 / fp16 (the reduced-precision modes
// of ops.set_precision: one plane set in LDS, one MFMA) -- instantiated for the production geometry (variant 2) only.
typedef _Float16 __attribute__((ext_vector_type(8))) half8_2;
__device__ __forceinline__ unsigned short f2h_2(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
// MT_: m-tiles (16-pixel rows) per wave.  6 (with WMW = 4, WNW = 2, NT = 4, BDB = false, NA_ = 2): block tile 24 x 16 px x 128 ch,
// wave tile 96 px x 64 ch (variant 9) -- the largest wave tile for Cout = 128 layers whose activation ring still has TWO slots
// (2 x 60 KB + 32 KB of weights = 152 KB): 20 ds_read_b128 per 72 MFMAs, and the chunk store overlaps the MFMAs as in variant 2.
// DUAL (round 4, with WNW = 4, bn = 256): the fused upscale with Cout = 128 as TWO output-phase pairs instead of four phases -- the
// N tile of 256 is [phase b = 0: 128 channels | phase b = 1: 128 channels] of one row phase a (= the group), N-waves 0-1 / 2-3.  A
// step is a tap ROW dy with one tap COLUMN per b (steps[i].z = (dx_b0 + 1) | (dx_b1 + 1) << 8): every wave multiplies in every
// step, the activation tile is staged once for two phases, and the layer runs on this kernel's 128 x 64 wave tiles instead of
// the tile kernel's 64 x 64.  Per output element the MFMA sequence is the tile kernel's (dy-major taps): bit-identical outputs.
template <int NT, int HALO, bool INSS, int WNW = 2, bool BDB = true, int NA_ = 2, int WMW = 2, int PREC = 0, int MT_ = 8, bool DUAL = false,
          int IOS = PPST_ST_F32, bool UP9 = false, bool K64 = false, bool KS = false>
__global__ __launch_bounds__(64 * WNW * WMW, WNW * WMW == 12 ? 3 : ((WNW * WMW == 8 || NA_ == 1) ? 2 : 1)) void conv_mfma2_kernel(Conv2KArgs a) {
  constexpr bool X3 = PREC == 0;
  // K64 (single-pass modes on half-stored activations): a step covers 64 input channels instead of 32 -- channels 0-31 of the chunk sit
  // where the x3 form keeps its hi planes, channels 32-63 where it keeps its lo planes (activation tile and weight blob alike), and a
  // product is two MFMAs (first half x first half + second half x second half) instead of three.  Same LDS image sizes, registers
  // and step pipeline as the fp32-class kernel; half the steps (barriers, fragment waits, DMA issues) per MFMA of the 32-channel
  // single-pass form, which spent two thirds of a step beside its matrix work (0.27-0.41 of the single-pass ceiling at 1024^2).
  static_assert(!K64 || (PREC != 0 && IOS != PPST_ST_F32 && !UP9 && !BDB), "K64: single-pass precision on half-stored activations");
  constexpr bool X3L = X3 || K64;                          // the LDS images have eight planes
  static_assert(!UP9 || (NT == 4 && HALO == 1 && !INSS && WNW == 4 && !BDB && NA_ == 2 && WMW == 2 && PREC == 0 && MT_ == 8 && !DUAL &&
                         IOS == PPST_ST_F32), "UP9: the production geometry, fp32-class, fp32 storage");
  // IOS: storage type of x, residual and y (ppst_conv_args.io_st; conv_mfma.hip): the single-pass modes, in their operand type
  static_assert(IOS == PPST_ST_F32 || IOS == (PREC == 3 ? PPST_ST_F16 : PREC == 1 ? PPST_ST_BF16 : -1), "half storage: single-pass modes");
  constexpr int ES = IOS == PPST_ST_F32 ? 4 : 2;
  constexpr int NWV = WMW * WNW;                          // waves per block
  constexpr int NTH = 64 * NWV;
  constexpr int MT = MT_;                                 // m-tiles (16-pixel rows) per wave
  constexpr int TH = MT * WMW, TW = 16;
  constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO, HP = HH * HW;
  constexpr int PLANE = ((HP * 16 + 255) / 256) * 256;    // bytes
  constexpr int ABUF = (X3L ? 8 : 4) * PLANE;             // hi g0..3 [, lo g0..3]
  constexpr int BN = WNW * 16 * NT;                       // WNW N-waves
  constexpr int BPLANE = BN * 16;
  constexpr int BBUF = (X3L ? 8 : 4) * BPLANE;
  constexpr int NA = NA_;
  constexpr int EPI_TILE = 64 * 36;
  constexpr int EPI_BYTES = NWV * EPI_TILE * 4 + WMW * BN * 2 * 4;
  constexpr int MAIN_BYTES = NA * ABUF + 2 * BBUF;
  __shared__ __attribute__((aligned(256))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  unsigned char* smA = smem;
  unsigned char* smB = smem + NA * ABUF;
#ifdef PPST_CONV_TRACE
  unsigned long long tr_c0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0)::"memory");
#endif

  // XCD-aware block -> (n index, m tile) map (bijective remap, N-major order): as conv_mfma.hip
  const int nwg = gridDim.x;
  int wid;
  {
    int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int m_count = a.B * a.tiles_y * a.tiles_x;
  const int nidx = wid / m_count;
  int midx = wid - nidx * m_count;
  const int group = nidx / a.n_tiles, ntile = nidx - group * a.n_tiles;
  const int b = midx / (a.tiles_y * a.tiles_x);
  midx -= b * a.tiles_y * a.tiles_x;
  const int tyi = midx / a.tiles_x, txi = midx - tyi * a.tiles_x;
  // (UP9: a block's 16 x 16 grid of u values yields the outputs of 15 x 15 input positions: the tiles overlap by one row / column)
  const int ty0 = tyi * (UP9 ? 15 : TH), tx0 = txi * (UP9 ? 15 : TW);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WNW, wm = wave / WNW;
  const int r16 = lane & 15, g = lane >> 4;

#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) int4* StepPtr;
#else
  typedef const int4* StepPtr;
#endif
  StepPtr steps = (StepPtr)(a.steps + (int64_t)group * a.nsteps);
  const unsigned char* wblob = (const unsigned char*)a.wpack + ((int64_t)nidx * a.nsteps) * BBUF;
  const unsigned char* xb = (const unsigned char*)a.x + (int64_t)b * a.in_h * a.in_w * a.in_ld * ES;
  // across-block K split (template KS: instances of their own -- the hand-over code costs the 128-accumulator kernels a few spilled
  // registers, which the plain instances, the swap path's, must not pay): this block's share of the table (conv_mfma.hip)
  int nst = a.nsteps;
  if (KS && a.ks.S > 1) {
    int s0, s1;
    ks_range(a.ks, (int)blockIdx.y, s0, s1);
    steps += s0;
    wblob += (int64_t)s0 * BBUF;
    nst = s1 - s0;
  }

  // ---- A staging (as conv_mfma.hip: one wave-instruction = 8 pixels x 128 B; fp32 -> bf16 hi/lo planes)
  constexpr int A_WCH = (HP + 7) / 8;
  constexpr int A_IT2 = (A_WCH * 64 + NTH - 1) / NTH;
  float4 ra[A_IT2];
  constexpr int A_NLOADS = (K64 ? 2 : 1) * (A_IT2 + (INSS ? 2 : 0));
  int aoff[A_IT2];
#pragma unroll
  for (int it = 0; it < A_IT2; ++it) {
    int i = tid + it * NTH;
    int l = i & 63;
    int pix = (i >> 6) * 8 + ((l >> 1) & 7), q4 = (l >> 4) * 2 + (l & 1);
    int o = -1;
    if (pix < HP) {
      int hy = pix / HW, hx = pix - hy * HW;
      int iy = ty0 + hy - HALO + a.in_off_y, ix = tx0 + hx - HALO + a.in_off_x;
      bool inb = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
      if (inb || a.pad_mode != PPST_PAD_ZERO) {
        iy = pad_index2(iy, a.in_h, a.pad_mode);
        ix = pad_index2(ix, a.in_w, a.pad_mode);
        o = ((iy * a.in_w + ix) * a.in_ld + q4 * 4) * ES;  // bytes, < 2^31: the entry point rejects larger images
      }
    }
    aoff[it] = o;
  }
  float4 ras0 = make_float4(1.f, 0.f, 1.f, 0.f), ras1 = ras0, ras2 = ras0, ras3 = ras0;    // (ras2 / ras3: K64, channels + 32)
  const int q4lane = ((tid & 63) >> 4) * 2 + (tid & 1);
  const float in_slope = (INSS && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  // buffer loads (SGPR descriptor + per-item byte offset + SGPR chunk offset; padding items out of range -> zeros): conv_mfma.hip
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, a.in_h * a.in_w * a.in_ld * ES, 0x00020000);
  auto a_load = [&](int chan_off) {
#pragma unroll
    for (int it = 0; it < A_IT2; ++it) {
      if (IOS == PPST_ST_F32) {
        ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff[it], chan_off * 4, 0));
      } else {       // four half elements: the raw 8 bytes ride in .x / .y until a_store
        const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrs, aoff[it], chan_off * 2, 0));
        ra[it].x = __uint_as_float(u.x); ra[it].y = __uint_as_float(u.y);
        if (K64) {   // the same four channel positions of the chunk's second half (channels 32-63) in .z / .w
          const uint2 u2 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(xrs, aoff[it], (chan_off + 32) * 2, 0));
          ra[it].z = __uint_as_float(u2.x); ra[it].w = __uint_as_float(u2.y);
        }
      }
    }
    if (INSS) {
      const float4* p = (const float4*)(a.in_ss + ((int64_t)b * a.in_c + chan_off + q4lane * 4) * 2);
      ras0 = p[0];
      ras1 = p[1];
      if (K64) { ras2 = p[16]; ras3 = p[17]; }      // (32 channels = 16 float4 of (a, s) pairs further)
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
      int i = tid + it * NTH;
      int l = i & 63;
      int pix = (i >> 6) * 8 + ((l >> 1) & 7);
      if (pix < HP) {
        float4 v = ra[it];                // (padding items: zeros from the out-of-range buffer load)
        const uint2 raw = make_uint2(__float_as_uint(ra[it].x), __float_as_uint(ra[it].y));
        if (IOS != PPST_ST_F32) v = st_unpack4<IOS>(raw);
        if (INSS && aoff[it] >= 0) {
          v.x = in_act(ras0.x * v.x + ras0.y); v.y = in_act(ras0.z * v.y + ras0.w);
          v.z = in_act(ras1.x * v.z + ras1.y); v.w = in_act(ras1.z * v.w + ras1.w);
        }
        uint2 hv, lv = make_uint2(0u, 0u);
        if (X3) split_bf16x4(v, hv, lv);     // two elements per conversion / subtraction instruction (common.h)
        else if (IOS != PPST_ST_F32 && !INSS) hv = raw;      // stored in the operand type already
        else if (PREC == 3) {
          const unsigned short h0 = f2h_2(v.x), h1 = f2h_2(v.y), h2 = f2h_2(v.z), h3 = f2h_2(v.w);
          hv = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        } else hv = f2bf_x4(v);
        if (K64) {      // the chunk's second 32 channels take the place of the lo planes
          const uint2 raw2 = make_uint2(__float_as_uint(ra[it].z), __float_as_uint(ra[it].w));
          if (!INSS) lv = raw2;
          else {
            float4 w = st_unpack4<IOS>(raw2);
            if (aoff[it] >= 0) {
              w.x = in_act(ras2.x * w.x + ras2.y); w.y = in_act(ras2.z * w.y + ras2.w);
              w.z = in_act(ras3.x * w.z + ras3.y); w.w = in_act(ras3.z * w.w + ras3.w);
            }
            lv = st_pack4<IOS>(w);
          }
        }
        int off = (l >> 4) * PLANE + pix * 16 + (l & 1) * 8;
        *(uint2*)(base + off) = hv;
        if (X3L) *(uint2*)(base + 4 * PLANE + off) = lv;
      }
    }
  };
  // ---- B staging: LDS-DMA of the pre-packed step blob (global_load_lds_dwordx4, 1 KB per wave-instruction)
  constexpr int B_WI = BBUF / 1024;
  constexpr int B_PER_WAVE = (B_WI + NWV - 1) / NWV;     // (12 waves: the last round is issued by the first B_WI % NWV waves only)
  // (UP9: the blob's columns are u-type major -- [type][N-wave][16 ch], so a 1-KB piece is ONE type of one plane and wave w moves the
  //  pieces of type w & 3: a step that feeds fewer types moves only their pieces, `tmask` = the types of the step being fetched)
  auto b_dma = [&](int s, int slot, int tmask = 0xF) {
    if (UP9 && !((tmask >> (wave & 3)) & 1)) return;
    const unsigned char* src = wblob + (int64_t)s * BBUF + lane * 16;
    unsigned char* dst = smB + slot * BBUF;
#pragma unroll
    for (int it = 0; it < B_PER_WAVE; ++it) {
      const int wi = it * NWV + wave;
      if (B_WI % NWV != 0 && wi >= B_WI) break;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + wi * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + wi * 1024), 16, 0, 0);
    }
  };
  // fragment addresses
#define B_ADDR(slot, nt) (smB + (slot) * BBUF + g * BPLANE + ((UP9 ? (nt) * 4 + wn : wn * NT + (nt)) * 16 + r16) * 16)
#define A_OFF(slot, dy, dx, mt) ((slot) * ABUF + g * PLANE + (((wm * MT + (mt) + HALO + (dy)) * HW + HALO + (dx) + r16) * 16))
#define DXW(z) (DUAL ? ((((z) >> (wn >= WNW / 2 ? 8 : 0)) & 0xff) - 1) : (z))   /* this wave's tap column of the step */

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // K-split instances: the block's share of the weight blobs requested up front, one 128-byte line per thread and round, so that the
  // chain of steps finds them in this XCD's L2 instead of running at memory latency (conv_mfma.hip; the requests retire with the
  // prologue's vmcnt(0), before the accumulators are live)
  constexpr int PF_ROUNDS = KS ? 12 : 1;
  float pf[PF_ROUNDS];
  if (KS && a.ks.S > 1) {
    const int pf_bytes = nst * BBUF;
#pragma unroll
    for (int r = 0; r < PF_ROUNDS; ++r) {
      const int o = (r * (64 * WNW * WMW) + tid) * 128;
      pf[r] = o < pf_bytes ? *(const volatile float*)(wblob + o) : 0.f;
    }
  }
  // ---- prologue: steps 0 and 1 staged, descriptors of step 2 fetched
  int4 d = steps[0];
  int dy0 = d.y, dx0 = DXW(d.z), sl0 = 0;
  int dy1 = d.y, dx1 = DXW(d.z), sl1 = 0;
  a_load(d.x);
  b_dma(0, 0);
  a_store(0);
  if (nst > 1) {
    d = steps[1];
    dy1 = d.y; dx1 = DXW(d.z);
    sl1 = (d.w & 1) ? 1 : 0;
    b_dma(1, 1, UP9 ? 0x3 : 0xF);
    if (d.w & 1) a_load(d.x);
    if (d.w & 1) a_store(sl1);
    if (NA > 1 && a.early_a && (d.w & 2)) a_load(d.w >> 8);
  }
  int4 dE = d, dO = d;
  if (nst > 2) dE = steps[2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (KS && a.ks.S > 1) {
#pragma unroll
    for (int r = 0; r < PF_ROUNDS; ++r) asm volatile("" ::"v"(pf[r]));
  }
  __syncthreads();

  bf16x8 b0h[NT], b0l[NT], b1h[BDB ? NT : 1], b1l[BDB ? NT : 1];
  bf16x8 ah, al;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    b0h[nt] = *(const bf16x8*)B_ADDR(0, nt);
    if (X3L) b0l[nt] = *(const bf16x8*)(B_ADDR(0, nt) + 4 * BPLANE);
  }
  ah = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, 0));
  if (X3L) al = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, 0) + 4 * PLANE);
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // One K-step.  bc* = this step's weight fragments (read during the previous step); bn* receive the next step's:
  // one n-tile pair per m-tile group, so the 2*NT reads are spread over the step.  The next A fragment (m-tile mt+1, or
  // m-tile 0 of step s+1) is read one group ahead.  The non-MFMA head (descriptor, weight DMA, activation loads) follows
  // the first group's MFMAs; the staging store of a new chunk sits under the second-to-last group.
// -DPPST_ABL_HALFBAR (timing ablation, results WRONG: races): only every second step ends in a barrier -- what a 4-slot weight ring
// with one barrier per step pair could gain at most.  Measured: +-1 % on the N-256 layers; <= 3 % with 64 px x 64 ch waves (the tile
// kernel's geometry instantiated on this loop, itself 3-5 % slower than conv_mfma.hip: commit bf39429).
#ifdef PPST_ABL_HALFBAR
#define HALFBAR_IF(s) if (((s) & 1) != 0)
#else
#define HALFBAR_IF(s)
#endif
#define STEP2(bch, bcl, bnh, bnl, s, D2, D3, H1, H2) STEP2M(bch, bcl, bnh, bnl, s, D2, D3, H1, H2, 0xF, 0xF)
#define STEP2M(bch, bcl, bnh, bnl, s, D2, D3, H1, H2, MASK, MASKN) STEP2N(bch, bcl, bnh, bnl, s, D2, D3, H1, H2, MASK, MASKN, 0xF)
#define STEP2N(bch, bcl, bnh, bnl, s, D2, D3, H1, H2, MASK, MASKN, MASK2)   /* MASK2: the types of step s + 2 (its weight DMA) */   /* MASK / MASKN: n-tiles this / the next step multiplies (UP9) */ \
  {                                                                                                           \
    const bool has1 = (H1), has2 = (H2);                                                                      \
    TR2_DECL TR2(0)                                                                                           \
    bool newA2 = false, a_early = false;                                                                      \
    int sl2 = sl1;                                                                                            \
    if (NA == 1 && freshA) {   /* this step's chunk was stored at the end of the previous step */             \
      ah = *(const bf16x8*)(smA + A_OFF(0, dy0, dx0, 0));                                                     \
      if (X3L) al = *(const bf16x8*)(smA + A_OFF(0, dy0, dx0, 0) + 4 * PLANE);                                         \
    }                                                                                                         \
    const bool storeA = NA == 1 && pendA;   /* the next step opens a chunk: store it at the end of this one */ \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                       \
      bf16x8 nh, nl;                                                                                          \
      if (mt < MT - 1) {                                                                                      \
        nh = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, mt + 1));                                            \
        if (X3L) nl = *(const bf16x8*)(smA + A_OFF(sl0, dy0, dx0, mt + 1) + 4 * PLANE);                                \
      } else if (has1 && !storeA) {                                                                           \
        nh = *(const bf16x8*)(smA + A_OFF(sl1, dy1, dx1, 0));                                                 \
        if (X3L) nl = *(const bf16x8*)(smA + A_OFF(sl1, dy1, dx1, 0) + 4 * PLANE);                                     \
      }                                                                                                       \
      if (BDB && has1 && mt >= 1 && mt - 1 < NT) {   /* next step's B fragment pair nt = mt - 1 */            \
        bnh[mt - 1] = *(const bf16x8*)B_ADDR(((s) + 1) & 1, mt - 1);                                          \
        if (X3L) bnl[mt - 1] = *(const bf16x8*)(B_ADDR(((s) + 1) & 1, mt - 1) + 4 * BPLANE);                           \
      }                                                                                                       \
      if (BDB && NT == 8 && has1 && mt == MT - 1) {                                                           \
        bnh[NT - 1] = *(const bf16x8*)B_ADDR(((s) + 1) & 1, NT - 1);                                          \
        if (X3L) bnl[NT - 1] = *(const bf16x8*)(B_ADDR(((s) + 1) & 1, NT - 1) + 4 * BPLANE);                           \
      }                                                                                                       \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                     \
        if (!(((MASK) >> nt) & 1)) continue;                                                                  \
        if (X3) {                                                                                             \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bch[nt], acc[mt][nt], 0, 0, 0);           \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bcl[nt], acc[mt][nt], 0, 0, 0);           \
        }                                                                                                     \
        if (K64) {   /* channels 32-63 of the chunk: the "lo" images */                                       \
          if (PREC == 3)                                                                                      \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_2, al), __builtin_bit_cast(half8_2, bcl[nt]), acc[mt][nt], 0, 0, 0); \
          else                                                                                                \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bcl[nt], acc[mt][nt], 0, 0, 0);         \
        }                                                                                                     \
        if (PREC == 3)                                                                                        \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_2, ah), __builtin_bit_cast(half8_2, bch[nt]), acc[mt][nt], 0, 0, 0); \
        else                                                                                                  \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bch[nt], acc[mt][nt], 0, 0, 0);           \
      }                                                                                                       \
      if (mt == 0) {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        D3 = steps[(s) + 3];                                                                                  \
        newA2 = has2 && (D2.w & 1);                                                                           \
        if (NA > 1 && newA2) sl2 = sl1 ^ 1;                                                                   \
        if (has2) b_dma((s) + 2, (s) & 1, (MASK2));                                                           \
        __builtin_amdgcn_sched_barrier(0);   /* DMA issued before the activation loads: counted vmcnt below */ \
        {                                                                                                     \
          const bool early_ = NA > 1 && a.early_a;                                                            \
          const bool ld_ = early_ ? (has2 && (D2.w & 2)) : newA2;                                             \
          const int ch_ = early_ ? (D2.w >> 8) : D2.x;                                                        \
          if (ld_) { a_load(ch_); a_early = NA == 1 || early_; }                                              \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        TR2(1)                                                                                                \
      }                                                                                                       \
      if (mt == 3) { TR2(2) }                                                                                 \
      ah = nh;                                                                                                \
      al = nl;                                                                                                \
      if (NA > 1 && mt == MT - 2 && newA2) a_store(sl2);                                                      \
    }                                                                                                         \
    TR2(3)                                                                                                    \
    if (storeA) {   /* every wave has read the old chunk -> overwrite the only slot */                        \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
      a_store(0);                                                                                             \
    }                                                                                                         \
    if (!BDB && has1) {   /* single register set: the next step's weight fragments replace this step's, now dead */ \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                     \
        if (!(((MASKN) >> nt) & 1)) continue;                                                                 \
        bnh[nt] = *(const bf16x8*)B_ADDR(((s) + 1) & 1, nt);                                                  \
        if (X3L) bnl[nt] = *(const bf16x8*)(B_ADDR(((s) + 1) & 1, nt) + 4 * BPLANE);                                   \
      }                                                                                                       \
    }                                                                                                         \
    TR2(4)                                                                                                    \
    if (a_early) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_NLOADS) : "memory");                              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    TR2(5)                                                                                                    \
    HALFBAR_IF(s) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                             \
    TR2(6)                                                                                                    \
    TR2_FLUSH(s, newA2)                                                                                       \
    freshA = storeA; pendA = newA2;                                                                           \
    dy0 = dy1; dx0 = dx1; sl0 = sl1;                                                                          \
    if (has2) { dy1 = D2.y; dx1 = DXW(D2.z); }                                                                \
    sl1 = sl2;                                                                                                \
  }
  int s = 0;
  bool freshA = false, pendA = false;
  if (UP9) {
    // chunks of four steps = the four input shifts (0,0), (-1,0), (0,-1), (-1,-1); the n-tiles of a wave are the four u types
    // [ee, eo, oe, oo] of its 16 channels, and a shift feeds 4 / 2 / 2 / 1 of them: nine products per input pixel instead of the
    // sixteen of the four-phase form (masks are compile-time: the skipped MFMAs and fragment reads are not in the code)
    for (; s + 5 < nst; s += 4) {
      STEP2N(b0h, b0l, b0h, b0l, s, dE, dO, true, true, 0xF, 0x3, 0x5)
      STEP2N(b0h, b0l, b0h, b0l, s + 1, dO, dE, true, true, 0x3, 0x5, 0x1)
      STEP2N(b0h, b0l, b0h, b0l, s + 2, dE, dO, true, true, 0x5, 0x1, 0xF)
      STEP2N(b0h, b0l, b0h, b0l, s + 3, dO, dE, true, true, 0x1, 0xF, 0x3)
    }
    for (; s < nst; s += 4) {
      STEP2N(b0h, b0l, b0h, b0l, s, dE, dO, s + 1 < nst, s + 2 < nst, 0xF, 0x3, 0x5)
      STEP2N(b0h, b0l, b0h, b0l, s + 1, dO, dE, s + 2 < nst, s + 3 < nst, 0x3, 0x5, 0x1)
      STEP2N(b0h, b0l, b0h, b0l, s + 2, dE, dO, s + 3 < nst, s + 4 < nst, 0x5, 0x1, 0xF)
      STEP2N(b0h, b0l, b0h, b0l, s + 3, dO, dE, s + 4 < nst, s + 5 < nst, 0x1, 0xF, 0x3)
    }
  } else
  if (BDB) {
    for (; s + 3 < nst; s += 2) {
      STEP2(b0h, b0l, b1h, b1l, s, dE, dO, true, true)
      STEP2(b1h, b1l, b0h, b0l, s + 1, dO, dE, true, true)
    }
    for (; s < nst; s += 2) {
      STEP2(b0h, b0l, b1h, b1l, s, dE, dO, s + 1 < nst, s + 2 < nst)
      if (s + 1 < nst) STEP2(b1h, b1l, b0h, b0l, s + 1, dO, dE, s + 2 < nst, s + 3 < nst)
    }
  } else {
    for (; s + 3 < nst; s += 2) {
      STEP2(b0h, b0l, b0h, b0l, s, dE, dO, true, true)
      STEP2(b0h, b0l, b0h, b0l, s + 1, dO, dE, true, true)
    }
    for (; s < nst; s += 2) {
      STEP2(b0h, b0l, b0h, b0l, s, dE, dO, s + 1 < nst, s + 2 < nst)
      if (s + 1 < nst) STEP2(b0h, b0l, b0h, b0l, s + 1, dO, dE, s + 2 < nst, s + 3 < nst)
    }
  }
#undef STEP2
#undef STEP2M
#undef STEP2N
#undef A_OFF
#undef DXW
#undef B_ADDR

  if constexpr (UP9) {
    // ---- UP9 epilogue.  acc[mt][t][j] = u_t at grid position (row wm * 8 + mt, column g * 4 + j) for channel ntile * 64 + wn * 16 +
    // r16, t = 0 ee (u[2y][2x]), 1 eo (u[2y][2x+1]), 2 oe (u[2y+1][2x]), 3 oo (u[2y+1][2x+1]) of the un-blurred transposed conv;
    // the fused upscale (stylegan2_layers.py:312-321: weights = sum of the four unit shifts of the padded kernel) is its 2 x 2 box sum
    //   out[2y][2x]     = ee[y][x] + eo[y][x]   + oe[y][x]   + oo[y][x]
    //   out[2y][2x+1]   = eo[y][x] + ee[y][x+1] + oo[y][x]   + oe[y][x+1]
    //   out[2y+1][2x]   = oe[y][x] + oo[y][x]   + ee[y+1][x] + eo[y+1][x]
    //   out[2y+1][2x+1] = oo[y][x] + oe[y][x+1] + eo[y+1][x] + ee[y+1][x+1]
    // for y, x = 0 .. 14 of the block's grid (row / column 15 only feed their neighbours).  The sums run in registers -- x + 1 is the
    // next register or lane + 16, y + 1 the next m-tile; row 8 comes from the upper M-wave through LDS -- then, row by row, the four
    // N-waves' 16-channel slices meet in an LDS tile [2 M-waves][2 output rows][32 output columns][64 ch] so that 16 lanes store the
    // 256 contiguous bytes of one output pixel; bias, noise, activation and the tile statistics ride on that store pass.
    constexpr int TST = 68;                              // floats per pixel of the pass tile (64 + 4: 16-byte aligned rows)
    constexpr int TPASS = 2 * 2 * 32 * TST;              // floats per pass buffer
    float* const xch = (float*)smem;                     // [wn][2 types][16 columns][16 ch]   (main-loop buffers are dead)
    float* const tile = (float*)smem + 4 * 2 * 16 * 16;  // two pass buffers
    float* const red = tile + 2 * TPASS;                 // [8 waves][64 ch][2]
    static_assert((4 * 2 * 16 * 16 + 2 * TPASS + 8 * 64 * 2) * 4 <= MAIN_BYTES, "UP9 epilogue buffers");
    if (wm == 1) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) xch[((wn * 2 + t) * 16 + g * 4 + j) * 16 + r16] = acc[0][t][j];
    }
    __syncthreads();
    float e8[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) e8[t][j] = wm == 0 ? xch[((wn * 2 + t) * 16 + g * 4 + j) * 16 + r16] : 0.f;
    const int act = a.act & 0xff;
    // store pass: thread = (output column ox = tid >> 4, channel quad c4 = tid & 15); per pass 2 M-waves x 2 output rows
    const int c4 = tid & 15, sox = tid >> 4;
    const int n0 = ntile * 64 + c4 * 4;
    const float4 bv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.y), 0, a.bias ? a.cout * 4 : 0, 0x00020000), n0 * 4, 0, 0));
    const int64_t img = (int64_t)b * a.out_h * a.out_w;
    float* const yb = a.y + img * a.out_ld;
    const float* const nzb = a.noise ? a.noise + img : nullptr;
    const int oxg = 2 * tx0 + sox;                        // global output column of this thread
    const bool okx = (sox >> 1) < 15 && tx0 + (sox >> 1) < a.tile_w && oxg < a.out_w;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    // the noise of this thread's 32 output pixels, all requested before the first store (vmcnt counts stores and retires in order:
    // a load behind a pass's stores waits for their acknowledgement -- and a load per pass sat in front of its own use)
    // (buffer loads: every request is unconditional -- pixels outside the tile / image and a launch without noise read out of range
    //  and get zeros; as conditional global loads hipcc waited for each of the 32 in turn)
    float nzv[8][4];
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(nzb ? nzb : (const float*)yb), 0,
                                                                         nzb ? a.out_h * a.out_w * 4 : 0, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int yl = (q >> 1) * 8 + mt, oy = 2 * (ty0 + yl) + (q & 1);
        const bool ok = okx && yl < 15 && ty0 + yl < a.tile_h && oy < a.out_h;
        nzv[mt][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nrs, ok ? (oy * a.out_w + oxg) * 4 : (int)0x80000000, 0, 0));
      }
#ifdef UP9_ABL_NOEPI
    if (a.out_scale != 123.f) return;
#endif
    // x + 1 neighbours of the first row
    float ee_n = __shfl_down(acc[0][0][0], 16, 64);       // ee[y][column 4 (g + 1)] for j = 3
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma clang fp contract(off)
      float een[4], eon[4];                               // ee / eo of row y + 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        een[j] = mt < 7 ? acc[mt < 7 ? mt + 1 : 7][0][j] : e8[0][j];
        eon[j] = mt < 7 ? acc[mt < 7 ? mt + 1 : 7][1][j] : e8[1][j];
      }
      const float ee_x = ee_n;                                          // ee[y][x + 1] for j = 3
      const float oe_x = __shfl_down(acc[mt][2][0], 16, 64);            // oe[y][x + 1] for j = 3
      const float een_x = __shfl_down(een[0], 16, 64);                  // ee[y + 1][x + 1] for j = 3
      ee_n = een_x;
      float* tb = tile + (mt & 1) * TPASS + (wm * 2 * 32) * TST + wn * 16 + r16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ee = acc[mt][0][j], eo = acc[mt][1][j], oe = acc[mt][2][j], oo = acc[mt][3][j];
        const float ee1 = j < 3 ? acc[mt][0][j < 3 ? j + 1 : 3] : ee_x, oe1 = j < 3 ? acc[mt][2][j < 3 ? j + 1 : 3] : oe_x;
        const float een1 = j < 3 ? een[j < 3 ? j + 1 : 3] : een_x;
        const float o00 = ((ee + eo) + oe) + oo;
        const float o01 = ((eo + ee1) + oo) + oe1;
        const float o10 = ((oe + oo) + een[j]) + eon[j];
        const float o11 = ((oo + oe1) + eon[j]) + een1;
        const int ox = 2 * (g * 4 + j);
        tb[(0 * 32 + ox) * TST] = o00;
        tb[(0 * 32 + ox + 1) * TST] = o01;
        tb[(1 * 32 + ox) * TST] = o10;
        tb[(1 * 32 + ox + 1) * TST] = o11;
      }
      // (LDS only: __syncthreads() would also wait for the previous pass's global stores -- one HBM round trip per pass, 3 k cycles)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // (one barrier per pass: the buffer of pass mt is rewritten by pass mt + 2, behind the barrier of pass mt + 1)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int wq = q >> 1, aq = q & 1;                 // M-wave, output row phase
        const int yl = wq * 8 + mt;                        // grid row
        const int oy = 2 * (ty0 + yl) + aq;
        if (okx && yl < 15 && ty0 + yl < a.tile_h && oy < a.out_h) {
          const float4 v = *(const float4*)(tile + (mt & 1) * TPASS + ((wq * 2 + aq) * 32 + sox) * TST + c4 * 4);
          const int pix = oy * a.out_w + oxg;
          const float nz = a.noise_weight * nzv[mt][q];
          float o[4] = {v.x + bv.x + nz, v.y + bv.y + nz, v.z + bv.z + nz, v.w + bv.w + nz};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float t = o[c];
            if (act == PPST_ACT_LRELU) t = (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
            o[c] = t * a.out_scale;
          }
          PPST_EPI_STORE(yb + ((int64_t)pix * a.out_ld + n0), o);
          s1.x += o[0]; s1.y += o[1]; s1.z += o[2]; s1.w += o[3];
          s2.x += o[0] * o[0]; s2.y += o[1] * o[1]; s2.z += o[2] * o[2]; s2.w += o[3] * o[3];
        }
      }
    }
    if (a.stats) {
      // lanes of a wave with equal c4 (lane & 15): xor 16, 32; then the eight waves through LDS
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
      }
      if (lane < 16) {
        float* r = red + (wave * 64 + c4 * 4) * 2;
        r[0] = s1.x; r[1] = s2.x; r[2] = s1.y; r[3] = s2.y; r[4] = s1.z; r[5] = s2.z; r[6] = s1.w; r[7] = s2.w;
      }
      __syncthreads();
      if (tid < 64) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { t0 += red[(w * 64 + tid) * 2]; t1 += red[(w * 64 + tid) * 2 + 1]; }
        float* o = a.stats + ((((int64_t)b * a.tiles_y + tyi) * a.tiles_x + txi) * a.cout + ntile * 64 + tid) * 2;
        o[0] = t0;
        o[1] = t1;
      }
    }
    return;
  }
  // ---- across-block K split (conv_mfma.hip): rows 0 .. S-2 hand their partial sums over and leave, row S-1 adds them and goes on
  if (KS && a.ks.S > 1) {
    const int S = a.ks.S, y = (int)blockIdx.y;
    constexpr int NTHR = 64 * WNW * WMW;
    float* const slot0 = a.ks.scratch + (int64_t)wid * (S - 1) * (MT * NT * 4 * NTHR) + tid;
    if (y < S - 1) {
      float* const dst = slot0 + (int64_t)y * (MT * NT * 4 * NTHR);
      KS_SCATTER(acc, MT, NT, NTHR, dst)
      ks_publish(a.ks, wid, y, tid);
      return;
    }
    ks_wait(a.ks, wid, tid);
    KS_GATHER(acc, MT, NT, NTHR, S, slot0);
  }
  // ---- epilogue (as conv_mfma.hip): per wave, passes of 64 pixels x 32 channels through an LDS transposition tile
  const int gy = DUAL ? group : group >> 1, gx = DUAL ? wn / (WNW / 2) : group & 1;   // output row / column phase
  constexpr int BNC = DUAL ? BN / 2 : BN;                                              // channels per N tile
  const int nw0 = DUAL ? (wn % (WNW / 2)) * (16 * NT) : wn * (16 * NT);               // this wave's first channel in the N tile
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  float* tw = (float*)smem + wave * EPI_TILE;
  float* red = (float*)smem + NWV * EPI_TILE;          // [2 (wm)][BN][2]
  const int f8 = lane & 7, prow = lane >> 3;
  // output addressing as in conv_mfma.hip: 32-bit element offsets inside image b, row / column-half strides wave-uniform
  const int egy = a.n_groups > 1 ? gy : 0, egx = a.n_groups > 1 ? gx : 0;   // (DUAL: n_groups = 2)
  const int tyb = ty0 + wm * MT, oyb = tyb * a.out_sy + egy;
  const int txl = tx0 + prow, oxl = txl * a.out_sx + egx;
  const bool okx0 = txl < a.tile_w && oxl < a.out_w, okx1 = txl + 8 < a.tile_w && oxl + 8 * a.out_sx < a.out_w;
  const int pix0 = oyb * a.out_w + oxl, rs_pix = a.out_sy * a.out_w;
  const int64_t img = (int64_t)b * a.out_h * a.out_w;
  unsigned char* const yb = (unsigned char*)a.y + img * a.out_ld * ES;
  const float* const nzb = a.noise ? a.noise + img : nullptr;
  const unsigned char* const rb = a.residual ? (const unsigned char*)a.residual + img * a.res_ld * ES : nullptr;
  // bias and noise of the whole wave tile fetched before the first store (vmcnt counts stores and retires in order: conv_mfma.hip)
  // (buffer loads, every request unconditional -- a pixel outside the tile / image and a launch without noise read out of range and
  //  get zero: as conditional global loads hipcc issued them one by one, each with its own wait; conv_mfma2.hip's UP9 epilogue)
  float nzv[2 * MT];
  float4 bva[NT / 2];
  const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(nzb ? (const void*)nzb : (const void*)yb), 0,
                                                                       nzb ? a.out_h * a.out_w * 4 : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) {
    const int r = i >> 1, c8 = i & 1;
    const bool ok = tyb + r < a.tile_h && oyb + r * a.out_sy < a.out_h && (c8 ? okx1 : okx0);
    nzv[i] = a.noise_weight * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  nrs, ok ? (pix0 + r * rs_pix + c8 * 8 * a.out_sx) * 4 : (int)0x80000000, 0, 0));
  }
#pragma unroll
  for (int pass = 0; pass < NT / 2; ++pass) {
    const int n0 = ntile * BNC + nw0 + pass * 32 + f8 * 4;
    bva[pass] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.y), 0, a.bias ? a.cout * 4 : 0, 0x00020000), n0 * 4, 0, 0));   // (out of range -> zeros)
  }
  float4 s1a[NT / 2], s2a[NT / 2];
#pragma unroll
  for (int i = 0; i < NT / 2; ++i) { s1a[i] = make_float4(0.f, 0.f, 0.f, 0.f); s2a[i] = s1a[i]; }
  // The pass loops are instantiated per (activation, residual mode): with both as run-time values hipcc evaluated the
  // leaky-ReLU AND the PReLU branch of every element and selected (~30 VALU instructions per element, 27.7 k cycles per tile
  // = 9 % of a 72-step tile, profiles/r02_conv_trace_v2.txt); a specialised pass needs ~10.
  auto epi_passes = [&](auto act_c, auto res_c) {
#pragma clang fp contract(off)   // no fused multiply-add here: every kernel family's epilogue must round like the others'
    const int ACT = act_c.value, RES = res_c.value;   // RES: 0 none, 1 joins before the activation, 2 after
#pragma unroll
  for (int mh = 0; mh < (MT + 3) / 4; ++mh) {
    const int MG = MT - mh * 4 < 4 ? MT - mh * 4 : 4;     // m-tiles of this group (MT = 6: 4 + 2)
#pragma unroll
    for (int pass = 0; pass < NT / 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ntl = 0; ntl < 2; ++ntl)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mt < MG) tw[(mt * 16 + g * 4 + j) * 36 + ntl * 16 + r16] = acc[mh * 4 + mt < MT ? mh * 4 + mt : 0][pass * 2 + ntl][j];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      const int nl0 = nw0 + pass * 32 + f8 * 4;
      const int n0 = ntile * BNC + nl0;
      const bool nok = n0 < a.cout;
      const float4 bv = bva[pass];
      const int yo0 = pix0 * a.out_ld + n0, ro0 = pix0 * a.res_ld + n0;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (it >= 2 * MG) continue;
        const int p = it * 8 + prow;
        const int r = mh * 4 + (it >> 1), c8 = it & 1;   // tile row of the wave and column half of this store
        float4 v = *(const float4*)(tw + p * 36 + f8 * 4);
        const bool rowok = tyb + r < a.tile_h && oyb + r * a.out_sy < a.out_h;
        if (nok && rowok && (c8 ? okx1 : okx0)) {
          const int d = r * rs_pix + c8 * 8 * a.out_sx;
          const float nz = nzv[mh * 8 + it];
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
          s1a[pass].x += o[0]; s1a[pass].y += o[1]; s1a[pass].z += o[2]; s1a[pass].w += o[3];
          s2a[pass].x += o[0] * o[0]; s2a[pass].y += o[1] * o[1]; s2a[pass].z += o[2] * o[2]; s2a[pass].w += o[3] * o[3];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
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
    if (!(WNW == 4 && NA_ == 2 && X3)) epi_passes(EpiR{act}, EpiR{resm});   // experiments / reduced precision: one generic instance
    else if (act == PPST_ACT_LRELU) EPI_GO(PPST_ACT_LRELU);
    else if (act == PPST_ACT_PRELU) EPI_GO(PPST_ACT_PRELU);
    else EPI_GO(PPST_ACT_NONE);
#undef EPI_GO
  }
  if (a.stats) {
    __syncthreads();     // the transposition tiles are done: `red` lies behind them, but other waves may still be in their last pass
#pragma unroll
    for (int pass = 0; pass < NT / 2; ++pass) {
      float4 s1 = s1a[pass], s2 = s2a[pass];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
      }
      if (prow == 0) {
        float* r = red + (wm * BN + wn * (16 * NT) + pass * 32 + f8 * 4) * 2;
        r[0] = s1.x; r[1] = s2.x; r[2] = s1.y; r[3] = s2.y; r[4] = s1.z; r[5] = s2.z; r[6] = s1.w; r[7] = s2.w;
      }
    }
    __syncthreads();
    const int tiles = a.tiles_y * a.tiles_x;
    for (int nl = tid; nl < BN; nl += NTH) {
      // (DUAL: the N tile is two column phases of BNC channels; the partial slots are the four phases' of the 4-group form)
      const int n = DUAL ? ntile * BNC + (nl % BNC) : ntile * BN + nl;
      const int og = DUAL ? group * 2 + nl / BNC : group, ogn = DUAL ? 4 : a.n_groups;
      if (n < a.cout) {
        float* o = a.stats + ((((int64_t)b * ogn + og) * tiles + tyi * a.tiles_x + txi) * a.cout + n) * 2;
        float t0 = red[nl * 2], t1 = red[nl * 2 + 1];
#pragma unroll
        for (int w = 1; w < WMW; ++w) { t0 += red[(w * BN + nl) * 2]; t1 += red[(w * BN + nl) * 2 + 1]; }
        o[0] = t0;
        o[1] = t1;
      }
    }
  }
#ifdef PPST_CONV_TRACE
  if (a.dbg && blockIdx.x < 8 && tid == 0) {   // block start / end stamps behind the step records: [8 blocks][2]
    unsigned long long c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    a.dbg[(int64_t)8 * NWV * 160 * 8 + blockIdx.x * 2] = tr_c0;
    a.dbg[(int64_t)8 * NWV * 160 * 8 + blockIdx.x * 2 + 1] = c1;
  }
#endif
}

// Entry used by ppst_conv2d_mfma (conv_mfma.hip) when a->bn is 256, or 128 with the fat-wave variant requested.
int ppst_conv2d_mfma2_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st) {
  Conv2KArgs k;
  k.x = (const float*)a->x; k.wpack = (const unsigned short*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.in_h = a->in_h; k.in_w = a->in_w; k.in_ld = a->in_ld; k.out_h = a->out_h; k.out_w = a->out_w;
  k.out_ld = a->out_ld; k.cout = a->cout; k.nsteps = a->nsteps; k.n_groups = a->n_groups; k.pad_mode = a->pad_mode;
  k.in_off_y = a->in_off_y; k.in_off_x = a->in_off_x; k.out_sy = a->out_sy; k.out_sx = a->out_sx; k.act = a->act;
  k.res_ld = a->res_ld; k.tile_h = a->tile_h; k.tile_w = a->tile_w;
  k.tiles_y = tiles_y; k.tiles_x = tiles_x; k.n_tiles = n_tiles;
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu;
  k.in_c = a->in_c; k.in_act = a->in_act;
  k.early_a = a->early_a ? 1 : 0;
  k.dbg = nullptr;
#ifdef PPST_CONV_TRACE
  k.dbg = (unsigned long long*)a->prelu;   // diagnostic builds: the (unused) prelu slot carries the debug buffer
  k.prelu = nullptr;
#endif
  const int blocks = a->n_groups * n_tiles * a->B * tiles_y * tiles_x;
  k.ks.scratch = nullptr; k.ks.flags = nullptr; k.ks.epoch = 0; k.ks.S = 1;
  if (a->ksplit > 1) {
    // across-block K split: the N-256 kernel in its plain forms (8 m-tiles x 4 n-tiles per wave, 8 waves) -- fp32-class mode on fp32
    // tensors, single-pass modes on half-stored ones (what the train step launches); S <= 4 (128 accumulator registers per thread)
    if (a->variant != 2 || a->dual_b || a->ksplit > 4 || (a->precision == 0 ? a->io_st != 0 : a->io_st == 0) || (a->k64 && !a->halo)) return PPST_EINVAL;
    const int e0 = ppst_ksplit_prepare_(a->ksplit, a->ksplit_starts, blocks, a->nsteps, 128, 512, st, &k.ks);
    if (e0 != PPST_OK) return e0;
    const dim3 gridk(blocks, k.ks.S);
#define LKS(HALO_, PREC_, IOS_, K64_)                                                                            \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, true, 4, false, 2, 2, PREC_, 8, false, IOS_, false, K64_, true>), gridk, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, false, 4, false, 2, 2, PREC_, 8, false, IOS_, false, K64_, true>), gridk, dim3(512), 0, st, k);          \
  } while (0)
    if (a->precision == 0) { if (a->halo) LKS(1, 0, PPST_ST_F32, false); else LKS(0, 0, PPST_ST_F32, false); }
    else if (a->precision == 1) { if (a->k64) LKS(1, 1, PPST_ST_BF16, true); else if (a->halo) LKS(1, 1, PPST_ST_BF16, false); else LKS(0, 1, PPST_ST_BF16, false); }
    else { if (a->k64) LKS(1, 3, PPST_ST_F16, true); else if (a->halo) LKS(1, 3, PPST_ST_F16, false); else LKS(0, 3, PPST_ST_F16, false); }
#undef LKS
    return PPST_LAUNCH_CHECK();
  }
  const dim3 grid(blocks);
  if (a->k64) {               // single-pass modes on half-stored activations, 64 input channels per step (the entry point checked the shape)
#define LK(PREC_, IOS_)                                                                                          \
  do {                                                                                                          \
    if (a->variant == 2 && a->dual_b) {                                                                         \
      if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, 1, true, 4, false, 2, 2, PREC_, 8, true, IOS_, false, true>), grid, dim3(512), 0, st, k);   \
      else PPST_LAUNCH((conv_mfma2_kernel<4, 1, false, 4, false, 2, 2, PREC_, 8, true, IOS_, false, true>), grid, dim3(512), 0, st, k);          \
    } else if (a->variant == 2 && a->halo) {                                                                    \
      if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, 1, true, 4, false, 2, 2, PREC_, 8, false, IOS_, false, true>), grid, dim3(512), 0, st, k);  \
      else PPST_LAUNCH((conv_mfma2_kernel<4, 1, false, 4, false, 2, 2, PREC_, 8, false, IOS_, false, true>), grid, dim3(512), 0, st, k);         \
    } else if (a->variant == 9 && a->halo) {   /* Cout = 128-class layers: 24 x 16 px x 128 ch tiles (two 61-KB activation slots) */ \
      if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, 1, true, 2, false, 2, 4, PREC_, 6, false, IOS_, false, true>), grid, dim3(512), 0, st, k);  \
      else PPST_LAUNCH((conv_mfma2_kernel<4, 1, false, 2, false, 2, 4, PREC_, 6, false, IOS_, false, true>), grid, dim3(512), 0, st, k);         \
    } else return PPST_EINVAL;                                                                                  \
  } while (0)
    if (a->precision == 3) LK(3, PPST_ST_F16); else LK(1, PPST_ST_BF16);
#undef LK
    return PPST_LAUNCH_CHECK();
  }
  if (a->variant == 11) {     // the fused upscale as nine products per input pixel + box sum (UP9); shape conditions checked by the entry point
    PPST_LAUNCH((conv_mfma2_kernel<4, 1, false, 4, false, 2, 2, 0, 8, false, PPST_ST_F32, true>), grid, dim3(512), 0, st, k);
    return PPST_LAUNCH_CHECK();
  }
#define L2(NT_, HALO_, WNW_, BDB_, NA_)                                                                         \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<NT_, HALO_, true, WNW_, BDB_, NA_>), grid, dim3(128 * WNW_), 0, st, k);  \
    else PPST_LAUNCH((conv_mfma2_kernel<NT_, HALO_, false, WNW_, BDB_, NA_>), grid, dim3(128 * WNW_), 0, st, k);         \
  } while (0)
#define L7(HALO_)                                                                                               \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, true, 2, false, 1, 4>), grid, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, false, 2, false, 1, 4>), grid, dim3(512), 0, st, k);          \
  } while (0)
#define L9(HALO_)                                                                                               \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, true, 2, false, 2, 4, 0, 6>), grid, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, false, 2, false, 2, 4, 0, 6>), grid, dim3(512), 0, st, k);          \
  } while (0)
#define L2Q(HALO_, PREC_, IOS_)                                                                                 \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, true, 4, false, 2, 2, PREC_, 8, false, IOS_>), grid, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, false, 4, false, 2, 2, PREC_, 8, false, IOS_>), grid, dim3(512), 0, st, k);          \
  } while (0)
#define L2P(HALO_, PREC_)                                                                                       \
  do {                                                                                                          \
    if (a->io_st) L2Q(HALO_, PREC_, (PREC_ == 3 ? PPST_ST_F16 : PPST_ST_BF16)); else L2Q(HALO_, PREC_, PPST_ST_F32); \
  } while (0)
#define L2D(PREC_, IOS_)                                                                                        \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, 1, true, 4, false, 2, 2, PREC_, 8, true, IOS_>), grid, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, 1, false, 4, false, 2, 2, PREC_, 8, true, IOS_>), grid, dim3(512), 0, st, k);          \
  } while (0)
  if (a->variant == 2 && a->dual_b) {     // Cout % 128 == 0 fused upscale as two phase pairs (halo 1)
    if (a->precision == 3) { if (a->io_st) L2D(3, PPST_ST_F16); else L2D(3, PPST_ST_F32); }
    else if (a->precision == 1) { if (a->io_st) L2D(1, PPST_ST_BF16); else L2D(1, PPST_ST_F32); }
    else L2D(0, PPST_ST_F32);
  } else
  // single-pass modes, Cout = 128-class layers: block tile 32 x 16 px x 128 ch on this kernel's 128 px x 64 ch wave tiles with TWO
  // activation slots (the hi planes alone are 39 KB per slot) -- the tile kernel's 64 x 64 wave tiles reach 0.22 of the single-pass
  // ceiling on 128 -> 128 @1024^2, bound by fragment reads and barriers per MFMA
#define L7Q(HALO_, PREC_, IOS_)                                                                                 \
  do {                                                                                                          \
    if (k.in_ss) PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, true, 2, false, 2, 4, PREC_, 8, false, IOS_>), grid, dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv_mfma2_kernel<4, HALO_, false, 2, false, 2, 4, PREC_, 8, false, IOS_>), grid, dim3(512), 0, st, k);          \
  } while (0)
#define L7P(PREC_)                                                                                              \
  do {                                                                                                          \
    if (!a->halo) return PPST_EINVAL;                                                                           \
    if (a->io_st) L7Q(1, PREC_, (PREC_ == 3 ? PPST_ST_F16 : PPST_ST_BF16)); else L7Q(1, PREC_, PPST_ST_F32);   \
  } while (0)
  if (a->variant == 7 && a->precision == 1) L7P(1);
  else if (a->variant == 7 && a->precision == 3) L7P(3);
  else
  if (a->variant == 2 && a->precision == 1) { if (a->halo) L2P(1, 1); else L2P(0, 1); }
  else if (a->variant == 2 && a->precision == 3) { if (a->halo) L2P(1, 3); else L2P(0, 3); }
  else if (a->variant == 2) {            // 8 waves, wave tile 128 px x 64 ch, N tile 256: the production form
    if (a->halo) L2(4, 1, 4, false, 2); else L2(4, 0, 4, false, 2);
  }
#ifdef PPST_EXPERIMENTS                  // measured and off: only in a PPST_EXPERIMENTS=1 build
  else if (a->variant == 9) {                 // 8 waves = 4 (M) x 2 (N), wave tile 96 px x 64 ch, block 24 x 16 px x 128 ch, two slots
    if (a->halo) L9(1); else L9(0);
  } else if (a->variant == 7) {                 // 8 waves = 4 (M) x 2 (N), wave tile 128 px x 64 ch, block 32 x 16 px x 128 ch, one slot
    if (a->halo) L7(1); else L7(0);
  } else if (a->variant == 3) {          // two 4-wave blocks per CU, wave tile 128 px x 64 ch, N tile 128, one activation slot
    if (a->halo) L2(4, 1, 2, false, 1); else L2(4, 0, 2, false, 1);
  } else if (a->bn == 256) { if (a->halo) L2(8, 1, 2, true, 2); else L2(8, 0, 2, true, 2); }
  else { if (a->halo) L2(4, 1, 2, true, 2); else L2(4, 0, 2, true, 2); }
#else
  else return PPST_EINVAL;
#endif
#undef L2
  return PPST_LAUNCH_CHECK();
}
