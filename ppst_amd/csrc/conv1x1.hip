// Streaming kernel for the 1x1 convolutions of the path (round 2): the skip / feature-head / ToSpatialCode convs of
// stylegan2_layers.py:167-202 (EqualConv2d k = 1) and generator.py's 1x1 heads -- every ppst_conv2d_mfma call whose
// step table has halo 0 (all taps (0,0)), one output group and unit output stride.
//
// Why a second kernel: conv_mfma.hip stages a 16x16-pixel activation tile per 32-channel chunk through LDS and walks the
// chunks with one barrier per step.  A 1x1 conv has ONE step per chunk, so every step waits for a fresh tile from HBM, and
// the per-tile prologue + epilogue (13 k + 15 k cycles, profiles/r02_conv_trace_thin.txt) exceed the 4-16 steps of work:
// these layers ran at 1.4-3 TB/s of algorithmic traffic although they are plain HBM-bound GEMMs [pixels x Cin] x [Cin x Cout].
//
// Design (HBM-bound: algorithmic bytes = 4 * pixels * (Cin + Cout) per launch, x Cout/64 input re-reads served by L2):
//   * no activation staging: a lane loads its own MFMA fragment -- 8 consecutive channels of one pixel, two 16-B loads,
//     4 lanes cover the 128-B chunk of a pixel -- splits it to bf16 hi / lo in registers and feeds the matrix pipe; the
//     next step's fragment is in flight while this step computes (64 KB of loads in flight per CU at full occupancy);
//   * weights: up to 8 step blobs (8 KB each, the ppst_conv_pack layout for bn = 64) are copied to LDS once per group of 8
//     steps; after that barrier the 8 waves of a block run independently -- no per-step barrier;
//   * block = 512 threads = 8 waves x (32 px x 64 ch), two blocks per CU (4 waves / SIMD, <= 128 VGPRs);
//   * operands swapped (weights as the MFMA's A operand): the accumulator of a lane then holds 4 CONSECUTIVE CHANNELS of
//     one pixel, so bias / residual / store are 16-B accesses straight from registers -- no LDS transposition;
//   * same MFMA sequence per output element as conv_mfma.hip (al*bh, ah*bl, ah*bh per step, steps in table order), so the
//     outputs are bit-identical to it (tests/gpu_diag.py t_conv_variants); tile statistics: one (sum, sumsq) row per
//     256-pixel block, fixed summation order (depends on H*W only, not on the batch).
#include "common.h"

struct C1Args {
  const float* x;
  const unsigned char* wpack;
  const int4* steps;
  float* y;
  const float* bias;
  const float* noise;
  const float* prelu;
  float* stats;
  const float* residual;
  float noise_weight, out_scale;
  int B, hw, in_ld, out_ld, cout, nsteps, act, res_ld, n_tiles, tiles;
  int in_h, in_w, out_h, out_w, tiles_x, pad_mode;    // TAPS kernels only
  const float* in_ss;
  const float* in_prelu;
  int in_c, in_act;
  const float* in_res;           // INSS == 2: added to a*x + s before in_act (ppst_conv_args.in_res); pixel stride in_res_ld
  int in_res_ld;
};

#define C1_GROUP 8          // step blobs resident in LDS at a time
#define C1_BLOB 8192        // bytes per step blob (bn = 64: 8 planes x 64 n x 16 B)

// PREC of the kernels below: 0 = bf16 hi + lo on both sides (fp32-class, three MFMAs per product); 1 / 3 = single-pass bf16 /
// fp16 (ops.set_precision(1 | 3): one plane set per blob, one MFMA).
typedef _Float16 __attribute__((ext_vector_type(8))) c1_half8;
template <int PREC>
__device__ __forceinline__ void c1_cvt(float v, unsigned short& h, unsigned short& l) {
  if (PREC == 0) split_bf16(v, h, l);
  else if (PREC == 3) { h = __builtin_bit_cast(unsigned short, (_Float16)v); l = 0; }
  else { h = f2bf(v); l = 0; }
}
// eight values -> the hi / lo operands; PREC 0 / 1 convert pairs (split_bf16x4 / f2bf_x4 of common.h: a third of the vector
// instructions of the element-wise form, same bits)
template <int PREC>
__device__ __forceinline__ void c1_cvt8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  if (PREC == 3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned short h, l;
      c1_cvt<PREC>(v[j], h, l);
      hi[j] = (short)h;
      lo[j] = (short)l;
    }
    return;
  }
  uint2 h0, l0 = make_uint2(0u, 0u), h1, l1 = make_uint2(0u, 0u);
  if (PREC == 0) {
    split_bf16x4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    split_bf16x4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
  } else {
    h0 = f2bf_x4(make_float4(v[0], v[1], v[2], v[3]));
    h1 = f2bf_x4(make_float4(v[4], v[5], v[6], v[7]));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}
// 8 stored half elements (one 16-byte load, riding in a float4) -> fp32
template <int IOS>
__device__ __forceinline__ void c1_unpack8(float4 r, float (&v)[8]) {
  const float4 a = st_unpack4<IOS>(make_uint2(__float_as_uint(r.x), __float_as_uint(r.y)));
  const float4 b = st_unpack4<IOS>(make_uint2(__float_as_uint(r.z), __float_as_uint(r.w)));
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <int PREC>
__device__ __forceinline__ f32x4 c1_mfma(bf16x8 bh, bf16x8 bl, bf16x8 xh, bf16x8 xl, f32x4 acc) {
  if (PREC == 0) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, xh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xh, acc, 0, 0, 0);
  }
  if (PREC == 3) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_half8, bh), __builtin_bit_cast(c1_half8, xh), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xh, acc, 0, 0, 0);
}

__device__ __forceinline__ int c1_pad(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// Epilogue shared by the kernels of this file: lane = pixel r16 of each m-tile, channels n0 + 4g .. 4g+3 of each n-tile
// (operands swapped in the MFMAs), everything a 16-B access straight from the accumulators; tile statistics through `red`.
template <int MT, int NT, bool TAPS, bool SPEC = true, int IOS = PPST_ST_F32>
__device__ __forceinline__ void c1_epilogue(const C1Args& a, f32x4 (&acc)[MT][NT], float (&red)[8][NT > 4 ? 128 : 64][2], int b, int mblk,
                                            int ntile, int pbase, int oy0, int ox, int tid, int wave, int r16, int g) {
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  float4 s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = make_float4(0.f, 0.f, 0.f, 0.f); s2[nt] = s1[nt]; }
  // Output addressing with 32-bit offsets inside image b; bias and noise of the whole wave tile are fetched before the first
  // store (vmcnt counts stores and retires in order: a load behind a store waits for that store's acknowledgement) -- as in
  // conv_mfma.hip.
  const int64_t img = (int64_t)b * (TAPS ? a.out_h * a.out_w : a.hw);
  constexpr int ES = IOS == PPST_ST_F32 ? 4 : 2;     // storage type of residual / y: ppst_conv_args.io_st
  unsigned char* const yb = (unsigned char*)a.y + img * a.out_ld * ES;
  const float* const nzb = a.noise ? a.noise + img : nullptr;
  const unsigned char* const rb = a.residual ? (const unsigned char*)a.residual + img * a.res_ld * ES : nullptr;
  const int nbase = ntile * (NT > 4 ? 128 : 64) + g * 4;
  // (buffer loads, every request unconditional -- a pixel outside the tile / image and a launch without noise read out of range and
  //  get zero: as conditional global loads hipcc issued them one by one, each with its own wait; conv_mfma2.hip's UP9 epilogue)
  float nzv[MT];
  float4 bva[NT];
  const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(nzb ? (const void*)nzb : (const void*)yb), 0,
                                                                       nzb ? (TAPS ? a.out_h * a.out_w : a.hw) * 4 : 0, 0x00020000);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = pbase + mt * 16 + r16;
    const bool ok = TAPS ? (oy0 + mt < a.out_h && ox < a.out_w) : p < a.hw;
    nzv[mt] = a.noise_weight * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                   nrs, ok ? (TAPS ? (oy0 + mt) * a.out_w + ox : p) * 4 : (int)0x80000000, 0, 0));
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    bva[nt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.y), 0, a.bias ? a.cout * 4 : 0, 0x00020000), (nbase + nt * 16) * 4, 0, 0));   // (out of range -> zeros)
  // instantiated per (activation, residual mode), as in conv_mfma.hip / conv_mfma2.hip (same arithmetic, no FMA contraction)
  auto epi_passes = [&](auto act_c, auto res_c) {
#pragma clang fp contract(off)
    const int ACT = act_c.value, RES = res_c.value;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = pbase + mt * 16 + r16;
    if (TAPS ? (oy0 + mt >= a.out_h || ox >= a.out_w) : p >= a.hw) continue;
    const int pin = TAPS ? (oy0 + mt) * a.out_w + ox : p;      // pixel index inside the image
    const int yo = pin * a.out_ld, ro = pin * a.res_ld;
    const float nz = nzv[mt];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n0 = nbase + nt * 16;
      if (n0 >= a.cout) continue;
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 bv = bva[nt];
      if (RES) rv = st_ld4<IOS>(rb, ro + n0);
      float o[4] = {acc[mt][nt][0] + bv.x + nz, acc[mt][nt][1] + bv.y + nz, acc[mt][nt][2] + bv.z + nz, acc[mt][nt][3] + bv.w + nz};
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
      st_st4<IOS>(yb, yo + n0, make_float4(o[0], o[1], o[2], o[3]));
      s1[nt].x += o[0]; s1[nt].y += o[1]; s1[nt].z += o[2]; s1[nt].w += o[3];
      s2[nt].x += o[0] * o[0]; s2[nt].y += o[1] * o[1]; s2[nt].z += o[2] * o[2]; s2[nt].w += o[3] * o[3];
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
    if (!SPEC) epi_passes(EpiR{act}, EpiR{resm});      // reduced-precision / experiment kernels: one generic instance
    else if (act == PPST_ACT_LRELU) EPI_GO(PPST_ACT_LRELU);
    else if (act == PPST_ACT_PRELU) EPI_GO(PPST_ACT_PRELU);
    else EPI_GO(PPST_ACT_NONE);
#undef EPI_GO
  }
  if (a.stats) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      // sum over the 16 pixels (r16) that share a channel quad: DPP within the 16-lane row (quad swaps, half mirror, row
      // mirror) -- every lane ends with the row total; no LDS crossbar traffic as __shfl_xor (ds_bpermute) would cost
      float4 u = s1[nt], w = s2[nt];
      float* uv[8] = {&u.x, &u.y, &u.z, &u.w, &w.x, &w.y, &w.z, &w.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float t = *uv[q];
        t += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
        t += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
        t += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0x141, 0xf, 0xf, true));   // row_half_mirror
        t += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0x140, 0xf, 0xf, true));   // row_mirror
        *uv[q] = t;
      }
      if (r16 == 0) {
        float* r = &red[wave][nt * 16 + g * 4][0];
        r[0] = u.x; r[1] = w.x; r[2] = u.y; r[3] = w.y; r[4] = u.z; r[5] = w.z; r[6] = u.w; r[7] = w.w;
      }
    }
    __syncthreads();
    if (tid < (NT > 4 ? 128 : 64)) {
      const int n = ntile * (NT > 4 ? 128 : 64) + tid;
      if (n < a.cout) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) { t0 += red[wv][tid][0]; t1 += red[wv][tid][1]; }
        float* o = a.stats + (((int64_t)b * a.tiles + mblk) * a.cout + n) * 2;
        o[0] = t0;
        o[1] = t1;
      }
    }
  }
}

// TAPS = false: 1x1 convs, a block covers 256 consecutive pixels of one image.
// TAPS = true : "direct" form for the thin 3x3 / stride-2 layers (Cin <= 128, Cout <= 128: E1 / E2 / D stems and the
//               generator's 32-64 channel tail): a block covers a 16x16-pixel tile (the statistics rows of conv_mfma.hip),
//               wave w its rows 2w, 2w+1; every step of the table -- (channel offset, dy, dx) -- is one fragment load at the
//               shifted pixel, padding resolved per lane (zero: value 0; reflect / replicate: index map).  The 9 taps
//               re-read a pixel 9 times through L1 / L2 instead of staging a halo tile in LDS: that costs TA bandwidth the
//               HBM-bound layers have to spare, and removes the tile's serial prologue (load -> split -> LDS -> barrier) and
//               the per-step barriers that left these layers at 0.10-0.25 of the MFMA ceiling.
// NT_: 16-channel tiles per wave (2 when Cout <= 32).
// INSS: 0 input as stored; 1 normalise-on-load in_act(a*x + s); 2 (1x1 form, fp32 storage) in_act(a*x + s + in_res): the resnet merge
// of the producer applied on load (its tensor is never written) -- the same fp32 operations in the same order as ppst_affine_act.
template <int INSS, bool TAPS = false, int NT_ = 4, int PREC = 0, int IOS = PPST_ST_F32>
__global__ __launch_bounds__(512, 4) void conv1x1_stream_kernel(C1Args a) {
  constexpr int MT = 2, NT = NT_;
  constexpr bool X3 = PREC == 0;
  // IOS: storage type of x, residual and y (ppst_conv_args.io_st): the single-pass modes, in their operand type
  static_assert(IOS == PPST_ST_F32 || IOS == (PREC == 3 ? PPST_ST_F16 : PREC == 1 ? PPST_ST_BF16 : -1), "half storage: single-pass modes");
  constexpr int ES = IOS == PPST_ST_F32 ? 4 : 2;
  constexpr int BLOB1 = X3 ? C1_BLOB : C1_BLOB / 2;      // single-pass blobs carry the hi planes only
  __shared__ uint4 sB[C1_GROUP * BLOB1 / 16];
  __shared__ float red[8][64][2];

  // XCD-aware order: each XCD gets a contiguous range of work ids; inside it the N tiles of one pixel block are adjacent
  const int nwg = gridDim.x;
  int wid;
  {
    int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ntile = wid % a.n_tiles;
  int mblk = wid / a.n_tiles;
  const int b = mblk / a.tiles;
  mblk -= b * a.tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int pbase = mblk * 256 + wave * 32;
  const int tyi = TAPS ? mblk / a.tiles_x : 0, txi = TAPS ? mblk - tyi * a.tiles_x : 0;
  const int oy0 = tyi * 16 + wave * 2, ox = txi * 16 + r16;       // TAPS: output pixel of m-tile mt = (oy0 + mt, ox)
  const unsigned char* xb = (const unsigned char*)a.x + (int64_t)b * (TAPS ? a.in_h * a.in_w : a.hw) * a.in_ld * ES;
  const unsigned char* wblob = a.wpack + (int64_t)ntile * a.nsteps * BLOB1;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) int4* StepPtr;
#else
  typedef const int4* StepPtr;
#endif
  StepPtr steps = (StepPtr)a.steps;

  int64_t aoff[MT];
  bool pok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = pbase + mt * 16 + r16;
    pok[mt] = TAPS ? true : p < a.hw;
    aoff[mt] = TAPS ? 0 : (int64_t)(pok[mt] ? p : 0) * a.in_ld + g * 8;
  }
  const float in_slope = (INSS && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  auto in_act = [&](float t) -> float {
    if (a.in_act == PPST_ACT_LRELU) return (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
    if (a.in_act == PPST_ACT_PRELU) return t >= 0.f ? t : t * in_slope;
    return t;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 raw[MT][2], ss[4];
  float4 rres[INSS == 2 ? MT : 1][2];
  auto a_load = [&](int4 d) {
    const int chan = d.x;
    if (TAPS) {
      int ix = ox + d.z;
      bool xok = ix >= 0 && ix < a.in_w;
      ix = c1_pad(ix, a.in_w, a.pad_mode);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int iy = oy0 + mt + d.y;
        const bool yok = iy >= 0 && iy < a.in_h;
        iy = c1_pad(iy, a.in_h, a.pad_mode);
        pok[mt] = a.pad_mode != PPST_PAD_ZERO || (xok && yok);
        aoff[mt] = ((int64_t)iy * a.in_w + ix) * a.in_ld + g * 8;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float4* p = (const float4*)(xb + (aoff[mt] + chan) * ES);
      raw[mt][0] = p[0];                                  // (half storage: the lane's 8 channels are these 16 bytes)
      if (IOS == PPST_ST_F32) raw[mt][1] = p[1];
      if (INSS == 2) {
        const int pp = pbase + mt * 16 + r16;
        const float4* q = (const float4*)(a.in_res + ((int64_t)b * a.hw + (pok[mt] ? pp : 0)) * a.in_res_ld + g * 8 + chan);
        rres[mt][0] = q[0];
        rres[mt][1] = q[1];
      }
    }
    if (INSS) {
      const float4* q = (const float4*)(a.in_ss + ((int64_t)b * a.in_c + chan + g * 8) * 2);
      ss[0] = q[0]; ss[1] = q[1]; ss[2] = q[2]; ss[3] = q[3];
    }
  };
  a_load(steps[0]);

  for (int s0 = 0; s0 < a.nsteps; s0 += C1_GROUP) {
    const int ng = a.nsteps - s0 < C1_GROUP ? a.nsteps - s0 : C1_GROUP;
    if (s0) __syncthreads();                    // every wave has finished reading the previous group's blobs
    for (int i = tid; i < ng * (BLOB1 / 16); i += 512) sB[i] = ((const uint4*)(wblob + (int64_t)s0 * BLOB1))[i];
    __syncthreads();
    for (int sl = 0; sl < ng; ++sl) {
      const int s = s0 + sl;
      // this step's fragments: normalise-on-load, fp32 -> bf16 hi / lo
      bf16x8 ah[MT], al[MT];
      bool vok[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) vok[mt] = pok[mt];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (IOS != PPST_ST_F32 && !INSS) {        // stored in the operand type already: the fragment is the loaded 16 bytes
          const float4 z = vok[mt] ? raw[mt][0] : make_float4(0.f, 0.f, 0.f, 0.f);
          ah[mt] = __builtin_bit_cast(bf16x8, z);
          al[mt] = ah[mt];
          continue;
        }
        float v[8] = {raw[mt][0].x, raw[mt][0].y, raw[mt][0].z, raw[mt][0].w, raw[mt][1].x, raw[mt][1].y, raw[mt][1].z, raw[mt][1].w};
        if (IOS != PPST_ST_F32) c1_unpack8<IOS>(raw[mt][0], v);
        if (INSS) {
          const float sc[8] = {ss[0].x, ss[0].z, ss[1].x, ss[1].z, ss[2].x, ss[2].z, ss[3].x, ss[3].z};
          const float sh[8] = {ss[0].y, ss[0].w, ss[1].y, ss[1].w, ss[2].y, ss[2].w, ss[3].y, ss[3].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float t = sc[j] * v[j] + sh[j];
            if (INSS == 2) {
              const float4 r4 = rres[mt][j >> 2];
              t += (j & 3) == 0 ? r4.x : (j & 3) == 1 ? r4.y : (j & 3) == 2 ? r4.z : r4.w;
            }
            v[j] = in_act(t);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!vok[mt]) v[j] = 0.f;
        c1_cvt8<PREC>(v, ah[mt], al[mt]);
      }
      if (s + 1 < a.nsteps) a_load(steps[s + 1]);            // in flight while the matrix pipe works on this step
      const unsigned char* bs = (const unsigned char*)sB + sl * BLOB1 + g * 1024 + r16 * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 bh = *(const bf16x8*)(bs + nt * 256);
        bf16x8 bl = bh;
        if (X3) bl = *(const bf16x8*)(bs + nt * 256 + 4096);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = c1_mfma<PREC>(bh, bl, ah[mt], al[mt], acc[mt][nt]);
      }
    }
  }

  c1_epilogue<MT, NT, TAPS, X3, IOS>(a, acc, red, b, mblk, ntile, pbase, oy0, ox, tid, wave, r16, g);
}

// Plain 3x3 stride-1 layers with few channels (variant 6): the direct form above re-reads every pixel 9 times through
// L1 / L2 and was L2-bandwidth-bound (no faster than the tile kernel).  Here a wave loads each of the 4 input rows its two
// output rows need ONCE per 32-channel chunk -- 16 pixels + the two edge pixels (lane 0 / lane 15 of each 16-lane row) --
// splits them to bf16 hi / lo once, and makes the dx = -1 / +1 fragments with DPP row shifts (row_shr:1 / row_shl:1, the
// edge pixel arriving through the `old` operand): 2.25x instead of 9x fragment traffic, a third of the conversions.
// The table order of a 'conv' k = 3 plan is (chunk, dy, dx) ascending; rows are visited ascending, so every output element
// sees the same MFMA sequence as in conv_mfma.hip (bit-identical).  Weights: the 9 step blobs of a chunk in LDS (72 KB).
#define C3_GROUP 9
template <int CTRL>
__device__ __forceinline__ bf16x8 c3_shift(bf16x8 edge, bf16x8 centre) {
  typedef int __attribute__((ext_vector_type(4))) i4;
  const i4 e = __builtin_bit_cast(i4, edge), c = __builtin_bit_cast(i4, centre);
  i4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_update_dpp(e[i], c[i], CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(bf16x8, r);
}

// NT_ = 8 (Cout = 65..128, blobs packed for bn = 128): wave tile 32 px x 128 ch, 256 registers, one block per CU (152 KB of
// LDS) -- the generator's 128 -> 128 @512^2 StyledConv, which the tile kernel ran at 0.43 of the ceiling (36-step tiles
// pay 20 % for their serial prologue + epilogue): here the two waves of a SIMD overlap one's conversions with the other's MFMAs.
template <bool INSS, int NT_, int PREC = 0, int IOS = PPST_ST_F32>
__global__ __launch_bounds__(512, NT_ > 4 ? 2 : 4) void conv3x3_direct_kernel(C1Args a) {
  constexpr int MT = 2, NT = NT_;
  constexpr bool X3 = PREC == 0;
  static_assert(IOS == PPST_ST_F32 || IOS == (PREC == 3 ? PPST_ST_F16 : PREC == 1 ? PPST_ST_BF16 : -1), "half storage: single-pass modes");
  constexpr int ES = IOS == PPST_ST_F32 ? 4 : 2;
  constexpr int PSTR = (NT > 4 ? 128 : 64) * 16;          // bytes per plane of a step blob (bn x 16 B)
  constexpr int BLOB = (X3 ? 8 : 4) * PSTR;               // hi planes g0..3 [, lo planes]
  __shared__ uint4 sB[C3_GROUP * BLOB / 16];
  __shared__ float red[8][NT > 4 ? 128 : 64][2];
  // persistent blocks (the launch caps the grid at two per CU): block i walks the contiguous range of work ids
  // [i * per, (i + 1) * per) -- neighbouring tiles share their halo rows in this XCD's L2 -- and, when the layer has a single
  // 32-channel chunk and one N tile, copies the 9 weight blobs to LDS once for all its tiles.
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int total = a.B * a.tiles * a.n_tiles;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int w_beg = blockIdx.x * per, w_end = w_beg + per < total ? w_beg + per : total;
  const int nchunks = a.nsteps / 9;
  const bool keep_b = nchunks == 1 && a.n_tiles == 1;
  for (int wid = w_beg; wid < w_end; ++wid) {
  const int ntile = wid % a.n_tiles;
  int mblk = wid / a.n_tiles;
  const int b = mblk / a.tiles;
  mblk -= b * a.tiles;
  const int tyi = mblk / a.tiles_x, txi = mblk - tyi * a.tiles_x;
  const int oy0 = tyi * 16 + wave * 2, ox = txi * 16 + r16;
  const unsigned char* xb = (const unsigned char*)a.x + (int64_t)b * a.in_h * a.in_w * a.in_ld * ES;
  const unsigned char* wblob = a.wpack + (int64_t)ntile * a.nsteps * BLOB;
  if (wid > w_beg) __syncthreads();            // `red` and (unless kept) the weight blobs of the previous tile are free
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) int4* StepPtr;
#else
  typedef const int4* StepPtr;
#endif
  StepPtr steps = (StepPtr)a.steps;
  const float in_slope = (INSS && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  auto in_act = [&](float t) -> float {
    if (a.in_act == PPST_ACT_LRELU) return (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
    if (a.in_act == PPST_ACT_PRELU) return t >= 0.f ? t : t * in_slope;
    return t;
  };
  // column of the centre pixel and of this lane's edge pixel (lane 0: x0 - 1, lane 15: x0 + 16; other lanes: unused copy)
  const int xe_raw = r16 == 0 ? ox - 1 : (r16 == 15 ? ox + 1 : ox);
  const bool cx_ok = ox < a.in_w, ex_ok = xe_raw >= 0 && xe_raw < a.in_w;
  const int cx = c1_pad(ox, a.in_w, a.pad_mode), ex = c1_pad(xe_raw, a.in_w, a.pad_mode);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 rawc[2], rawe[2], ss[4];
  bool okc = false, oke = false;
  auto row_load = [&](int chan, int r) {     // input row oy0 - 1 + r of the chunk at channel `chan`
    int iy = oy0 - 1 + r;
    const bool yok = iy >= 0 && iy < a.in_h;
    iy = c1_pad(iy, a.in_h, a.pad_mode);
    okc = a.pad_mode != PPST_PAD_ZERO || (yok && cx_ok);
    oke = a.pad_mode != PPST_PAD_ZERO || (yok && ex_ok);
    const float4* pc = (const float4*)(xb + (((int64_t)iy * a.in_w + cx) * a.in_ld + chan + g * 8) * ES);
    const float4* pe = (const float4*)(xb + (((int64_t)iy * a.in_w + ex) * a.in_ld + chan + g * 8) * ES);
    rawc[0] = pc[0];
    rawe[0] = pe[0];
    if (IOS == PPST_ST_F32) { rawc[1] = pc[1]; rawe[1] = pe[1]; }      // (predicating this load to the 8 lanes that need it measured SLOWER: the branch costs
                                           //  more than re-reading the row from L1)
  };
  auto ss_load = [&](int chan) {
    if (INSS) {
      const float4* q = (const float4*)(a.in_ss + ((int64_t)b * a.in_c + chan + g * 8) * 2);
      ss[0] = q[0]; ss[1] = q[1]; ss[2] = q[2]; ss[3] = q[3];
    }
  };
  auto convert = [&](const float4 (&raw)[2], bool ok, bf16x8& hi, bf16x8& lo) {
    if (IOS != PPST_ST_F32 && !INSS) {          // stored in the operand type already
      hi = __builtin_bit_cast(bf16x8, ok ? raw[0] : make_float4(0.f, 0.f, 0.f, 0.f));
      lo = hi;
      return;
    }
    float v[8] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
    if (IOS != PPST_ST_F32) c1_unpack8<IOS>(raw[0], v);
    if (INSS) {
      const float sc[8] = {ss[0].x, ss[0].z, ss[1].x, ss[1].z, ss[2].x, ss[2].z, ss[3].x, ss[3].z};
      const float sh[8] = {ss[0].y, ss[0].w, ss[1].y, ss[1].w, ss[2].y, ss[2].w, ss[3].y, ss[3].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in_act(sc[j] * v[j] + sh[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!ok) v[j] = 0.f;
    c1_cvt8<PREC>(v, hi, lo);
  };

  int chan = steps[0].x;
  ss_load(chan);
  row_load(chan, 0);
  for (int c = 0; c < nchunks; ++c) {
    if (!keep_b || wid == w_beg) {
      if (c) __syncthreads();                  // every wave has finished reading the previous chunk's blobs
      for (int i = tid; i < C3_GROUP * (BLOB / 16); i += 512) sB[i] = ((const uint4*)(wblob + (int64_t)c * C3_GROUP * BLOB))[i];
      __syncthreads();
    }
    const int chan_next = c + 1 < nchunks ? steps[(c + 1) * 9].x : chan;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bf16x8 ch, cl, eh, el;
      convert(rawc, okc, ch, cl);
      convert(rawe, oke, eh, el);
      if (r < 3) row_load(chan, r + 1);
      else if (c + 1 < nchunks) { ss_load(chan_next); row_load(chan_next, 0); }
      // (ss of the NEXT chunk replaces this chunk's only after row 3 has been converted: r == 3 branch above)
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        // dx = -1: lane i <- lane i-1, lane 0 <- edge;  dx = +1: lane i <- lane i+1, lane 15 <- edge
        const bf16x8 fh = dxi == 0 ? c3_shift<0x111>(eh, ch) : (dxi == 1 ? ch : c3_shift<0x101>(eh, ch));
        const bf16x8 fl = dxi == 0 ? c3_shift<0x111>(el, cl) : (dxi == 1 ? cl : c3_shift<0x101>(el, cl));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int dyi = r - mt;                                            // dy + 1
          if (dyi < 0 || dyi > 2) continue;
          const unsigned char* bs = (const unsigned char*)sB + (dyi * 3 + dxi) * BLOB + g * PSTR + r16 * 16;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 bh = *(const bf16x8*)(bs + nt * 256);
            bf16x8 bl = bh;
            if (X3) bl = *(const bf16x8*)(bs + nt * 256 + 4 * PSTR);
            acc[mt][nt] = c1_mfma<PREC>(bh, bl, fh, fl, acc[mt][nt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the weight-fragment reads of later taps from being hoisted (registers)
      }
    }
    chan = chan_next;
  }
  c1_epilogue<MT, NT, true, (X3 && NT <= 4), IOS>(a, acc, red, b, mblk, ntile, 0, oy0, ox, tid, wave, r16, g);
  }
}

// Entry used by ppst_conv2d_mfma (conv_mfma.hip) for variant 4.  tiles = statistics rows per image the caller allocated
// (ppst_conv_tiles); block i of an image covers pixels [256 i, 256 i + 256) -- rows past the last pixel come out zero.
static inline int64_t blocks_of(const ppst_conv_args* a, int tiles, int n_tiles) { return (int64_t)a->B * tiles * n_tiles; }
int ppst_conv1x1_stream_launch(const ppst_conv_args* a, int n_tiles, int tiles, hipStream_t st) {
  C1Args k;
  k.x = (const float*)a->x; k.wpack = (const unsigned char*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.hw = a->out_h * a->out_w; k.in_ld = a->in_ld; k.out_ld = a->out_ld; k.cout = a->cout; k.nsteps = a->nsteps;
  k.act = a->act; k.res_ld = a->res_ld; k.n_tiles = n_tiles; k.tiles = tiles;
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu; k.in_c = a->in_c; k.in_act = a->in_act;
  k.in_res = (const float*)a->in_res; k.in_res_ld = a->in_res_ld;
  if ((int64_t)tiles * 256 < k.hw) return PPST_EINVAL;
  if (k.in_res) {          // (ppst_conv2d_mfma has checked: in_scale_shift given, precision 0, fp32 storage)
    PPST_LAUNCH((conv1x1_stream_kernel<2, false, 4, 0, PPST_ST_F32>), dim3((unsigned)blocks_of(a, tiles, n_tiles)), dim3(512), 0, st, k);
    return PPST_LAUNCH_CHECK();
  }
  const int64_t blocks = (int64_t)a->B * tiles * n_tiles;
  if (blocks > 0x7fffffff) return PPST_EINVAL;
#define LS(PREC_, IOS_)                                                                                               \
  do {                                                                                                                \
    if (k.in_ss) PPST_LAUNCH((conv1x1_stream_kernel<1, false, 4, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k);   \
    else PPST_LAUNCH((conv1x1_stream_kernel<0, false, 4, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k);          \
  } while (0)
  if (a->precision == 1) { if (a->io_st) LS(1, PPST_ST_BF16); else LS(1, PPST_ST_F32); }
  else if (a->precision == 3) { if (a->io_st) LS(3, PPST_ST_F16); else LS(3, PPST_ST_F32); }
  else LS(0, PPST_ST_F32);
#undef LS
  return PPST_LAUNCH_CHECK();
}

// variant 5: the direct form for thin layers with taps (tiles_y x tiles_x 16x16-pixel tiles per image)
int ppst_conv_direct_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st) {
  C1Args k;
  k.x = (const float*)a->x; k.wpack = (const unsigned char*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.hw = a->out_h * a->out_w; k.in_ld = a->in_ld; k.out_ld = a->out_ld; k.cout = a->cout; k.nsteps = a->nsteps;
  k.act = a->act; k.res_ld = a->res_ld; k.n_tiles = n_tiles; k.tiles = tiles_y * tiles_x;
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu; k.in_c = a->in_c; k.in_act = a->in_act;
  k.in_h = a->in_h; k.in_w = a->in_w; k.out_h = a->out_h; k.out_w = a->out_w; k.tiles_x = tiles_x; k.pad_mode = a->pad_mode;
  const int64_t blocks = (int64_t)a->B * k.tiles * n_tiles;
  if (blocks > 0x7fffffff) return PPST_EINVAL;
  const bool nt2 = a->cout <= 32;
#define LD(INSS_, PREC_, IOS_)                                                                                       \
  do {                                                                                                               \
    if (nt2) PPST_LAUNCH((conv1x1_stream_kernel<INSS_, true, 2, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k);       \
    else PPST_LAUNCH((conv1x1_stream_kernel<INSS_, true, 4, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k);           \
  } while (0)
#define LDP(PREC_, IOS_) do { if (k.in_ss) LD(1, PREC_, IOS_); else LD(0, PREC_, IOS_); } while (0)
  if (a->precision == 1) { if (a->io_st) LDP(1, PPST_ST_BF16); else LDP(1, PPST_ST_F32); }
  else if (a->precision == 3) { if (a->io_st) LDP(3, PPST_ST_F16); else LDP(3, PPST_ST_F32); }
  else LDP(0, PPST_ST_F32);
#undef LDP
#undef LD
  return PPST_LAUNCH_CHECK();
}

// variant 6: plain 3x3 stride-1 plans (table order (chunk, dy, dx), nsteps = 9 * chunks, taps in [-1,1]^2)
int ppst_conv3x3_direct_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st) {
  if (a->nsteps % 9) return PPST_EINVAL;
  C1Args k;
  k.x = (const float*)a->x; k.wpack = (const unsigned char*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.hw = a->out_h * a->out_w; k.in_ld = a->in_ld; k.out_ld = a->out_ld; k.cout = a->cout; k.nsteps = a->nsteps;
  k.act = a->act; k.res_ld = a->res_ld; k.n_tiles = n_tiles; k.tiles = tiles_y * tiles_x;
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu; k.in_c = a->in_c; k.in_act = a->in_act;
  k.in_h = a->in_h; k.in_w = a->in_w; k.out_h = a->out_h; k.out_w = a->out_w; k.tiles_x = tiles_x; k.pad_mode = a->pad_mode;
  int64_t blocks = (int64_t)a->B * k.tiles * n_tiles;
  if (blocks > 0x7fffffff) return PPST_EINVAL;
  // single-chunk layers: persistent blocks (two 8-wave blocks per CU, 72 KB of LDS each) that keep the weights resident
  if (a->nsteps == 9 && n_tiles == 1 && blocks > 512) blocks = 512;
  const bool nt2 = a->cout <= 32, nt8 = a->bn == 128;
#define LD3(INSS_, PREC_, IOS_)                                                                                      \
  do {                                                                                                               \
    if (nt8 && PREC_ == 0) PPST_LAUNCH((conv3x3_direct_kernel<INSS_, 8, 0>), dim3((unsigned)blocks), dim3(512), 0, st, k);   \
    else if (nt8) return PPST_EUNSUPPORTED;                                                                          \
    else if (nt2) PPST_LAUNCH((conv3x3_direct_kernel<INSS_, 2, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k); \
    else PPST_LAUNCH((conv3x3_direct_kernel<INSS_, 4, PREC_, IOS_>), dim3((unsigned)blocks), dim3(512), 0, st, k);          \
  } while (0)
#define LD3P(PREC_, IOS_) do { if (k.in_ss) LD3(true, PREC_, IOS_); else LD3(false, PREC_, IOS_); } while (0)
  if (a->precision == 1) { if (a->io_st) LD3P(1, PPST_ST_BF16); else LD3P(1, PPST_ST_F32); }
  else if (a->precision == 3) { if (a->io_st) LD3P(3, PPST_ST_F16); else LD3P(3, PPST_ST_F32); }
  else LD3P(0, PPST_ST_F32);
#undef LD3P
#undef LD3
  return PPST_LAUNCH_CHECK();
}
