// 3x3 stride-1 convolution with FEWER matrix passes per algorithmic MAC (round 4): Winograd F(2,3) along x, direct along y.
//
// Same operation, padding modes, normalise-on-load and epilogue as conv_mfma.hip (the StyledConv / EqualConv2d conv of
// stylegan2_layers.py:184-193, 275-348, 439-475) for plain 3x3 stride-1 step tables -- ppst_conv_args.variant 10.
//
//   y[2p]   = d0 g0 + d1 g1 + d2 g2          with  d0..d3 = x[2p-1 .. 2p+2] of one input row, g0..g2 the row's three taps:
//   y[2p+1] = d1 g0 + d2 g1 + d3 g2
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d3 - d1          (input transform, while the tile is staged)
//     U0 = g0   U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2   (weight transform, in the pack: ppst_conv_pack_wino)
//     m_i = sum over (ky, cin) of V_i U_i                                   (4 GEMMs instead of 6 products per output pair)
//     y[2p] = m0 + m1 + m2   y[2p+1] = m1 - m2 + m3                         (output transform, in front of the epilogue)
// (position 3 carries the opposite sign of the textbook form, V3 and the output transform: see the staging.)  12 K-steps of 32 channels per chunk
// and PAIR of pixels instead of 9 per pixel: 1.5x fewer MFMAs, operands still bf16 hi + lo (three MFMAs per product, fp32
// accumulation): fp32-class like the other conv kernels (4-7e-6 against float64), NOT bit-identical to them.
//
// Why this shape.  The 2-D form F(2x2, 3x3) needs 4x the accumulators per output and 16/9 of the weight
// bytes per chunk: a block that fits the register file (T tiles x N channels = 4096) would have to pull ~100 B/cycle/CU of
// transformed weights from L2 and 128 KB of transformed activations per 32-channel chunk through LDS -- neither exists.  The 1-D
// form doubles the accumulators only, and the four transform positions i map onto WAVES: block tile 16 x 16 px x 128 ch, 8 waves
// = 4 (i) x 2 (channel halves), wave tile 128 pixel-pairs x 64 channels (8 x 4 MFMA tiles, 128 accumulator registers -- the
// wave tile of conv_mfma2.hip).  Consequences:
//   * a wave's weight fragments belong to that wave alone: they come straight from global memory / L2 into registers in fragment
//     order (pre-packed: one wave-instruction = 1 KB contiguous), no LDS, no DMA, no weight ring, no per-step barrier --
//     ONE block barrier per 32-channel chunk (288 MFMAs per wave) instead of one per tap (48);
//   * LDS holds only the transformed activation tile V: [i][hi|lo][k-group][18 rows][8 pairs][8 ch] bf16 = 72 KB per chunk,
//     two slots; an A fragment (2 tile rows x 8 pairs x 32 k) is one conflict-free ds_read_b128 per plane, the three ky taps are
//     row shifts of the same image;
//   * registers: weight fragments live in two half sets (n-tiles 0-1 / 2-3, 16 registers each); each half is reloaded for the
//     next K-step while the other half's 48 MFMAs run, so the L2 latency hides under the wave's own MFMAs and the partner wave's;
//   * the four m_i of a pixel pair sit in four waves: the output transform runs through LDS (which the epilogue's transposition
//     needs anyway), 32 channels per pass, double-buffered, one barrier per pass.
#include "common.h"

struct WinoKArgs {
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
  int B, in_h, in_w, in_ld, out_ld, cout, nchunk, pad_mode, act, res_ld;
  int tiles_y, tiles_x, n_tiles;
  const float* in_ss;
  const float* in_prelu;
  int in_c, in_act;
  KSplitDev ks;          // across-block K split (common.h): grid row y runs chunks [y * nchunk / S, (y + 1) * nchunk / S)
};

__device__ __forceinline__ int wino_pad_index(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// Timing ablations (tests/build_wino_variant.sh; results WRONG on purpose): -DWINO_ABL_NOB no weight-fragment loads in the loop,
// -DWINO_ABL_NOSTAGE no activation staging in the loop, -DWINO_ABL_NOSTSTORE its loads only, -DWINO_ABL_NOLDSW its arithmetic
// without the LDS stores, -DWINO_ABL_NOA no LDS fragment reads, -DWINO_ABL_NOMFMA no MFMAs, -DWINO_ABL_NOEPI no output transform /
// epilogue, -DWINO_ABL_NOBAR no chunk barrier (races).  Their numbers: DESIGN.md section 4 (j).
#ifdef WINO_ABL_NOA
#define WINO_LDA(dst, off, keep) dst = keep
#else
#define WINO_LDA(dst, off, keep) dst = *(const bf16x8*)(smem + (off))
#endif
#ifdef WINO_ABL_NOMFMA
#define WINO_MFMA(a_, b_, c_) asm volatile("" ::"v"(a_), "v"(b_))
#else
#define WINO_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
#endif
#define WINO_STEP_BYTES 8192   // one wave's weight fragments of one K-step: 4 n-tiles x (hi | lo) x 64 lanes x 16 B
#define WINO_OOB ((int)0x80000000)   // buffer-load offset of a padding item: out of range of every image (< 2^31 bytes) -> zeros

// split_bf16x4 (common.h) with the residual formed by scalar subtractions: packed fp32 arithmetic beside MFMAs costs more than two
// plain instructions (microarch guide), and this kernel's staging runs in the MFMA stream.  Same conversions, same subtraction:
// bit-identical to split_bf16x4.
__device__ __forceinline__ void wino_split4(float4 v, uint2& hi, uint2& lo) {
#ifdef WINO_PK_SPLIT
  split_bf16x4(v, hi, lo);
#else
  const unsigned h0 = f2bf_pk((ppst_f2){v.x, v.y}), h1 = f2bf_pk((ppst_f2){v.z, v.w});
  const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xffff0000u);
  const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xffff0000u);
  hi = make_uint2(h0, h1);
  lo = make_uint2(f2bf_pk((ppst_f2){rx, ry}), f2bf_pk((ppst_f2){rz, rw}));
#endif
}

// the value of lane ^ 8 (row_ror:8 -- a rotation by 8 inside each row of 16 lanes); hipcc folds the move into the consuming VALU
// instruction's DPP operand
__device__ __forceinline__ float wino_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}

// The lane id, computed afresh (2 VALU) and opaque to hipcc: what is derived from it at a point of use -- LDS store address, table
// address, weight-fragment offset, sign of the pair exchange -- is then not a loop-invariant register.  The kernel sits AT the 256
// registers of two waves per SIMD; every invariant hipcc hoisted was one more spill, and a scratch reload is a vector-memory
// operation: the in-order vmcnt makes its wait drain the weight fragments requested for the next step.
__device__ __forceinline__ int wino_lane() {
  int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}

// INMODE: normalise on load -- 0 off, 1 affine only (in_act none: the StyledConv conv2 case), 2 affine + activation
// FAT: four waves (ONE per SIMD, up to 512 registers) instead of eight: a wave takes a transform position with all 128 channels
// (wave tile 128 pairs x 128 ch, 256 accumulator registers), reads every A fragment ONCE per K-step (16 ds_read_b128 per 192 MFMAs
// instead of 32 per 96 ... reloaded half a step ahead) and has no SIMD
// partner running the same program phase.
template <int INMODE, bool FAT, bool KS = false>
__global__ __launch_bounds__(FAT ? 256 : 512, FAT ? 1 : 2) void conv_wino_kernel(WinoKArgs a) {
  constexpr bool INSS = INMODE != 0;
  constexpr int NWV = FAT ? 4 : 8, NTH = 64 * NWV, NTW = FAT ? 8 : 4;      // waves, threads, n-tiles per wave
  constexpr int NRND = FAT ? 5 : 3;                                        // halo rows a wave stages per chunk: wave + NWV * round
  constexpr int NP = 8, HH = 18;
  constexpr int PLANE = HH * NP * 16;      // one k-group plane of one transform position: 2304 B (a multiple of 256)
  constexpr int XIB = 8 * PLANE;           // hi g0..3 | lo g0..3
  constexpr int ABUF = 4 * XIB + 128;      // 73856 B per chunk slot
  constexpr int TROW = 36;                 // padded row of the output-transform tile (floats)
  constexpr int TXI = 128 * TROW;          // floats per transform position
  constexpr int TBUF = 4 * TXI;            // floats per pass buffer (73728 B: the two buffers overlay the two chunk slots)
  // [2 chunk slots | (a, s) table of normalise-on-load, up to 1024 channels]; the epilogue's pass buffers overlay the chunk slots
  __shared__ __attribute__((aligned(256))) unsigned char smem[2 * ABUF + 1024 * 2 * 4];
  float* const ss_lds = (float*)(smem + 2 * ABUF);

  // XCD-aware block -> (n tile, image tile) map: as conv_mfma.hip (bijective remap, N-major order)
  const int nwg = gridDim.x;
  int wid;
  {
    int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int m_count = a.B * a.tiles_y * a.tiles_x;
  const int ntile = wid / m_count;
  int midx = wid - ntile * m_count;
  const int b = midx / (a.tiles_y * a.tiles_x);
  midx -= b * a.tiles_y * a.tiles_x;
  const int tyi = midx / a.tiles_x, txi = midx - tyi * a.tiles_x;
  const int ty0 = tyi * 16, tx0 = txi * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = FAT ? wave : wave >> 1, nh = FAT ? 0 : wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) int4* StepPtr;
#else
  typedef const int4* StepPtr;
#endif
  StepPtr steps = (StepPtr)a.steps;
  // across-block K split (ppst_conv_args.ksplit, conv_mfma.hip): this block's chunks; its weight stream starts 3 c0 steps in
  int nchunk = a.nchunk, c0 = 0;
  if (KS && !FAT && a.ks.S > 1) {      // (instances of their own: the plain ones -- the swap path's -- keep their register allocation)
    int s0, s1;
    ks_range(a.ks, (int)blockIdx.y, s0, s1);
    c0 = s0 / 9;
    nchunk = s1 / 9 - c0;
    steps += c0 * 9;
  }
  const int nsteps = nchunk * 3;
  const int64_t xoff_ = (int64_t)b * a.in_h * a.in_w * a.in_ld;
  const float* xb = a.x + (((int64_t)__builtin_amdgcn_readfirstlane((int)(xoff_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)xoff_));

  // ---- activation staging.  A wave-instruction covers ONE halo row: lane = (k-group sg: 8 channels) x (se) x (pixel pair sp, lowest bits).
  // The two lanes of a pair split the four PIXELS, not the channels: the even lane loads d0 and d2, the odd lane d3 and d1 (8
  // channels each, "A" and "B").  Then
  //     A - B                  = d0 - d2 = V0 (even)   |   d3 - d1 = V3 (odd)
  //     B(partner) + sgn * B   = d1 + d2 = V1 (even)   |   d2 - d1 = V2 (odd),  sgn = +1 / -1
  // -- one lane-pair exchange (a DPP operand of the add), and every lane ends up with ALL 8 channels of a k-group for two
  // positions: its LDS stores are 16-byte ds_write_b128 (hi and lo of two positions = 4 per row).  Measured against the first
  // form of this kernel (4 channels x 4 positions per lane, 8-byte stores merged by hipcc into ds_write2st64_b64): the LDS
  // stores cost 20 % of the kernel there, and the same bytes as 16-byte stores half of that (timing ablations).
  const int sp = lane & 7, se = (lane >> 3) & 1, sg = lane >> 4;
  int colA, colB;
  bool okA, okB;
  {
    auto col = [&](int k, bool& ok) __attribute__((always_inline)) {
      int ix = tx0 + 2 * sp - 1 + k;
      ok = (ix >= 0 && ix < a.in_w) || a.pad_mode != PPST_PAD_ZERO;
      ix = wino_pad_index(ix, a.in_w, a.pad_mode);
      return ok ? (ix * a.in_ld + sg * 8) * 4 : WINO_OOB;
    };
    colA = col(se ? 3 : 0, okA);
    colB = col(se ? 1 : 2, okB);
  }
  const bool interior = a.pad_mode != PPST_PAD_ZERO || (ty0 >= 1 && ty0 + 16 < a.in_h && tx0 >= 1 && tx0 + 16 < a.in_w);
  // the halo rows this wave stages per chunk: wave + NWV * round (eight waves: three rounds, four waves: five); waves >= 2 have no
  // last row: they request row 0 (no branch around a load: see the loop) and skip the arithmetic and the stores
  int rowoff[NRND];
#pragma unroll
  for (int r = 0; r < NRND; ++r) {
    const int hrow = wave + NWV * r;
    int iy = ty0 + hrow - 1;
    const bool inb = iy >= 0 && iy < a.in_h;
    int o = -1;
    if (hrow < HH && (inb || a.pad_mode != PPST_PAD_ZERO)) {
      iy = wino_pad_index(iy, a.in_h, a.pad_mode);
      o = iy * a.in_w * a.in_ld * 4;          // bytes, < 2^31: the entry point rejects larger images
    }
    rowoff[r] = __builtin_amdgcn_readfirstlane(o);
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, a.in_h * a.in_w * a.in_ld * 4, 0x00020000);
  const float in_slope = (INSS && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  // normalise-on-load activation, branch-free: act(t) = (t > 0 ? t : t * in_neg) * in_pos with (in_neg, in_pos) = (1, 1) none,
  // (0.2, sqrt 2) leaky ReLU, (slope, 1) PReLU -- the same operations as conv_mfma.hip's three-way form, element for element
  const float in_neg = !INSS ? 1.f : a.in_act == PPST_ACT_LRELU ? 0.2f : a.in_act == PPST_ACT_PRELU ? in_slope : 1.f;
  const float in_pos = (INSS && a.in_act == PPST_ACT_LRELU) ? 1.41421356237309515f : 1.f;
  auto in_act = [&](float t) __attribute__((always_inline)) -> float { return (t > 0.f ? t : t * in_neg) * in_pos; };
  float4 rd[4];                                       // A ch 0-3, A ch 4-7, B ch 0-3, B ch 4-7
  auto stage_load = [&](float4 (&q)[4], int r, int chan) __attribute__((always_inline)) {      // r compile-time after unrolling; chan wave-uniform
    // a halo row outside the image under zero padding (rowoff < 0; wave-uniform) is requested at row 0 -- in bounds, ONE address
    // register pair for all rows -- and zeroed in stage_prep
    const int so = (rowoff[r] >= 0 ? rowoff[r] : 0) + chan * 4;
    // (without normalise-on-load the kernel has the two registers per row to spare: the row's requests go out of range and come back
    //  as zeros; with it they are requested at row 0 and zeroed by the prep's mask)
    const int va = (INMODE == 0 && rowoff[r] < 0) ? WINO_OOB : colA, vb = (INMODE == 0 && rowoff[r] < 0) ? WINO_OOB : colB;
    // (channels 4-7: +16 bytes on the SCALAR offset -- two address registers per lane instead of four; the four spilled in
    //  the normalise-on-load build, and every scratch reload is a vector-memory operation the in-order vmcnt has to drain)
    q[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, va, so, 0));
    q[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, va, so + 16, 0));
    q[2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vb, so, 0));
    q[3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vb, so + 16, 0));
  };
  auto stage_skip = [&](int r) __attribute__((always_inline)) { return r == NRND - 1 && wave >= 2; };     // (wave-uniform; LDS stores only behind it)
  // normalise on load: (a, s) of the lane's 8 channels from the block's LDS copy of the image's table (a global load here sat in
  // front of its own use: one L2 round trip per row in the middle of the MFMA stream, 3.14 instead of 2.4 ms on 128 -> 128 @512^2)
  const int chan_first = steps[0].x;
  auto stage_prep = [&](float4 (&q)[4], int r, int chan, int part) __attribute__((always_inline)) {   // part 0: channels 0-3 of A and B, part 1: channels 4-7
    const bool rowok = rowoff[r] >= 0;
    if (INSS) {
      const float4* p = (const float4*)((const unsigned char*)ss_lds + (chan - chan_first) * 8 + (wino_lane() >> 4) * 64) + part * 2;
      const float4 s0 = p[0], s1 = p[1];
      auto nact = [&](float t) __attribute__((always_inline)) -> float { return INMODE == 2 ? in_act(t) : t; };
      if (interior) {      // (wave-uniform: no padding item in this block's halo -- or a padding mode that has none: no masks)
#pragma unroll
        for (int h = part; h < 4; h += 2) {
          float4 v = q[h];
          v.x = nact(s0.x * v.x + s0.y); v.y = nact(s0.z * v.y + s0.w);
          v.z = nact(s1.x * v.z + s1.y); v.w = nact(s1.z * v.w + s1.w);
          q[h] = v;
        }
      } else {
#pragma unroll
        for (int h = part; h < 4; h += 2) {            // (padding items: zeros from the out-of-range buffer load, and they stay zero)
          float4 v = q[h];
          if (rowok && (h < 2 ? okA : okB)) {
            v.x = nact(s0.x * v.x + s0.y); v.y = nact(s0.z * v.y + s0.w);
            v.z = nact(s1.x * v.z + s1.y); v.w = nact(s1.z * v.w + s1.w);
          } else v = make_float4(0.f, 0.f, 0.f, 0.f);
          q[h] = v;
        }
      }
    }
  };
  // piece 0: A - B -> position 0 (se = 0) / 3 (se = 1); piece 1: B(partner) + sgn B -> position 1 / 2.  The 8 lanes of a
  // ds_write_b128 group are the 8 pairs of one (se, sg): 128 contiguous bytes of one plane.
  auto stage_put = [&](float4 (&q)[4], int r, int slot, int piece) __attribute__((always_inline)) {
    const int hrow = wave + NWV * r;
    const float4 A0 = q[0], A1 = q[1], B0_ = q[2], B1_ = q[3];
    float4 V0, V1;
    int i;
    const int ln = wino_lane(), se = (ln >> 3) & 1, stb = (ln >> 4) * PLANE + (ln & 7) * 16;
    const float sgn = se ? -1.f : 1.f;
    if (piece == 0) {
      i = se ? 3 : 0;
      V0 = make_float4(A0.x - B0_.x, A0.y - B0_.y, A0.z - B0_.z, A0.w - B0_.w);
      V1 = make_float4(A1.x - B1_.x, A1.y - B1_.y, A1.z - B1_.z, A1.w - B1_.w);
    } else {
      i = se ? 2 : 1;
      V0 = make_float4(wino_swap1(B0_.x) + sgn * B0_.x, wino_swap1(B0_.y) + sgn * B0_.y, wino_swap1(B0_.z) + sgn * B0_.z, wino_swap1(B0_.w) + sgn * B0_.w);
      V1 = make_float4(wino_swap1(B1_.x) + sgn * B1_.x, wino_swap1(B1_.y) + sgn * B1_.y, wino_swap1(B1_.z) + sgn * B1_.z, wino_swap1(B1_.w) + sgn * B1_.w);
    }
#ifdef WINO_ABL_NOSTSTORE
    asm volatile("" ::"v"(V0.x), "v"(V0.y), "v"(V0.z), "v"(V0.w), "v"(V1.x), "v"(V1.y), "v"(V1.z), "v"(V1.w));
    return;
#endif
    uint2 h0, l0, h1, l1;
    wino_split4(V0, h0, l0);
    wino_split4(V1, h1, l1);
    unsigned char* dst = smem + (slot * ABUF + hrow * NP * 16) + i * XIB + stb;
#ifdef WINO_ABL_NOLDSW       /* the arithmetic without the LDS stores */
    asm volatile("" ::"v"(h0.x), "v"(h0.y), "v"(l0.x), "v"(l0.y), "v"(h1.x), "v"(h1.y), "v"(l1.x), "v"(l1.y));
#else
    *(uint4*)dst = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *(uint4*)(dst + 4 * PLANE) = make_uint4(l0.x, l0.y, l1.x, l1.y);
#endif
  };

  // ---- weight fragments: this wave's stream [step][n-tile 0..3][hi | lo][lane][16 B], straight into registers
  // (block- and wave-uniform, but computed with vector divisions: say so, or every load through it becomes a waterfall loop)
  const int64_t woff_ = ((((int64_t)ntile * 4 + xi) * 2 + nh) * (a.nchunk * 3) + c0 * 3) * WINO_STEP_BYTES;
  const unsigned char* wbase = a.wpack + (((int64_t)__builtin_amdgcn_readfirstlane((int)(woff_ >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)woff_));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (FAT ? 2 : 1) * nsteps * WINO_STEP_BYTES, 0x00020000);
  bf16x8 B0[2][2], B1[2][2];      // [n-tile of the half][hi, lo]
  auto load_b = [&](bf16x8 (&dst)[2][2], int s, int half) __attribute__((always_inline)) {
#ifdef WINO_ABL_NOB
    if (s > 0) return;
#endif
    const int so = s * WINO_STEP_BYTES + half * 4096;
    const int lo16 = wino_lane() * 16;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        dst[n][h] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lo16 + (n * 2 + h) * 1024, so, 0));
  };

  // (FAT) both halves of a step as one set of eight n-tiles; two sets: the next step's is requested at the top of this one
  bf16x8 BxA[FAT ? 4 : 1][2], BxB[FAT ? 4 : 1][2];       // (FAT) n-tiles 0-3 / 4-7: each reloaded for the next step behind its half
  auto load_b4 = [&](bf16x8 (&dst)[FAT ? 4 : 1][2], int st_, int half) __attribute__((always_inline)) {
    if (!FAT) return;
#ifdef WINO_ABL_NOB
    if (st_ > 0) return;
#endif
    // (the pack keeps the two channel halves of a position as two streams of steps: n-tiles 4..7 come from the second)
    const int so = st_ * WINO_STEP_BYTES + half * nsteps * WINO_STEP_BYTES;
    const int lo16 = wino_lane() * 16;
#pragma unroll
    for (int n = 0; n < (FAT ? 4 : 1); ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        dst[n][h] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lo16 + (n * 2 + h) * 1024, so, 0));
  };

  // ---- prologue: every request of the block's first chunk goes out at once (weight fragments of step 0, the halo rows, the
  // (a, s) table) -- the accumulators are not live yet, so every row has its register set -- then chunk 0 is staged into slot 0.
  // (Row by row with the weight request last: +-1.5 %, and the plain build then spills in its loop.)
  if (FAT) { load_b4(BxA, 0, 0); load_b4(BxB, 0, 1); }
  else { load_b(B0, 0, 0); load_b(B1, 0, 1); }
  {
    float4 q[NRND][4];
#pragma unroll
    for (int r = 0; r < NRND; ++r) stage_load(q[r], r, chan_first);
    if (INSS) {
      const float* src = a.in_ss + ((int64_t)b * a.in_c + chan_first) * 2;
      for (int i = tid * 4; i < nchunk * 64; i += NTH * 4) *(float4*)(ss_lds + i) = *(const float4*)(src + i);
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < NRND; ++r) {
      if (!stage_skip(r)) {
        stage_prep(q[r], r, chan_first, 0); stage_prep(q[r], r, chan_first, 1);
        stage_put(q[r], r, 0, 0); stage_put(q[r], r, 0, 1);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  f32x4 acc[8][NTW];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define WA_OFF(slot, dy, mt) ((slot) * ABUF + xi * XIB + g * PLANE + ((2 * (mt) + (dy)) * NP + r16) * 16)
  int s = 0;
  if constexpr (FAT) {
    // One wave per SIMD: the step is ONE stream of 192 MFMAs (8 m-tiles x 8 n-tiles x 3) with everything else in its gaps -- per
    // m-tile one A fragment request (pinned one m-tile ahead), and the two halo rows of the step transformed piecewise between
    // the m-tiles.  The next step's weight fragments (16 KB) go out at the top of the step into the other register set.
    float4 rd2[4];
    bf16x8 ah, al;
    int slot = 0, chan_next = 0, aoff0 = 0;
    // one K-step = two halves (n-tiles 0-3 with set A, 4-7 with set B): 96 MFMAs each; a set is reloaded for the next step right
    // behind its half, i.e. half a step (~1.5 k cycles) ahead of its use.  The step's two halo rows are transformed piecewise
    // between the m-tiles: round r0 in the first half, r1 in the second.
    auto step_fat = [&](auto dy_c) __attribute__((always_inline)) {
      constexpr int dy = decltype(dy_c)::value;
      const int sn = s + 1 < nsteps ? s + 1 : s;
      constexpr int r0 = 2 * dy, r1 = dy < 2 ? 2 * dy + 1 : 0;
      stage_load(rd, r0, chan_next);
      if (dy < 2) stage_load(rd2, r1, chan_next);
      __builtin_amdgcn_sched_barrier(0);
      const bool skip0 = stage_skip(r0);
      int aoff1 = aoff0;
      asm volatile("" : "+v"(aoff1));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          bf16x8 nh_ = ah, nl_ = al;
          if (mt < 7) {
            WINO_LDA(nh_, (half ? aoff1 : aoff0) + (2 * (mt + 1) + dy) * NP * 16, ah);
            WINO_LDA(nl_, (half ? aoff1 : aoff0) + (2 * (mt + 1) + dy) * NP * 16 + 4 * PLANE, al);
          } else if (half == 0) {
            WINO_LDA(nh_, aoff1 + dy * NP * 16, ah);
            WINO_LDA(nl_, aoff1 + dy * NP * 16 + 4 * PLANE, al);
          } else if (dy < 2) {
            WINO_LDA(nh_, aoff0 + (dy + 1) * NP * 16, ah);
            WINO_LDA(nl_, aoff0 + (dy + 1) * NP * 16 + 4 * PLANE, al);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (half == 0 && !skip0) {
            if (mt == 3) stage_prep(rd, r0, chan_next, 0);
            if (mt == 4) stage_prep(rd, r0, chan_next, 1);
            if (mt == 5) stage_put(rd, r0, slot ^ 1, 0);
            if (mt == 6) stage_put(rd, r0, slot ^ 1, 1);
          }
          if (half == 1 && dy < 2) {                        // (rounds 1 and 3 exist for every wave)
            if (mt == 1) stage_prep(rd2, r1, chan_next, 0);
            if (mt == 2) stage_prep(rd2, r1, chan_next, 1);
            if (mt == 3) stage_put(rd2, r1, slot ^ 1, 0);
            if (mt == 4) stage_put(rd2, r1, slot ^ 1, 1);
          }
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (half == 0) {
              WINO_MFMA(al, BxA[n][0], acc[mt][n]);
              WINO_MFMA(ah, BxA[n][1], acc[mt][n]);
              WINO_MFMA(ah, BxA[n][0], acc[mt][n]);
            } else {
              WINO_MFMA(al, BxB[n][0], acc[mt][4 + n]);
              WINO_MFMA(ah, BxB[n][1], acc[mt][4 + n]);
              WINO_MFMA(ah, BxB[n][0], acc[mt][4 + n]);
            }
          }
          ah = nh_;
          al = nl_;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (half == 0) load_b4(BxA, sn, 0); else load_b4(BxB, sn, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++s;
    };
    auto chunk_fat = [&](int c) __attribute__((always_inline)) {
      slot = c & 1;
      chan_next = steps[(c + 1 < nchunk ? c + 1 : c) * 9].x;
      aoff0 = WA_OFF(slot, 0, 0);
      ah = *(const bf16x8*)(smem + aoff0);
      al = *(const bf16x8*)(smem + aoff0 + 4 * PLANE);
      step_fat(EpiC<0>{}); step_fat(EpiC<1>{}); step_fat(EpiC<2>{});
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int c = 0; c < nchunk; ++c) chunk_fat(c);
  } else {
  // A fragment of m-tile mt (tile rows 2 mt, 2 mt + 1; 8 pairs each) for tap row dy: halo rows 2 mt + dy, 2 mt + dy + 1.
  //
  // Every vector-memory instruction of the loop is UNCONDITIONAL (the last step re-requests its own weight fragments, the last
  // chunk re-requests its own activation rows, waves without a third halo row request zeros): vmcnt retires in order, and with a
  // load behind a branch hipcc can only wait for vmcnt(0) -- i.e. for the weight fragments it has just requested.
  // Every A fragment request is PINNED (scheduling fence) in front of the previous m-tile's MFMAs: hipcc otherwise sinks each
  // request to just before its first use and waits at once -- zero prefetch distance (~1.6 k cycles of s_waitcnt per step and
  // wave by counter).  The staged row's two transform pieces sit between m-tiles of the second half, free to interleave with
  // that m-tile's six MFMAs.
  for (int c = 0; c < nchunk; ++c) {
    const int slot = c & 1;
    const bool more = c + 1 < nchunk;
    const int chan_next = steps[(more ? c + 1 : c) * 9].x;
    const int aoff0 = WA_OFF(slot, 0, 0);
    int aoff1 = aoff0;
    asm volatile("" : "+v"(aoff1));        // the second half RE-READS its A fragments: kept in registers they cost 64 VGPRs (spills)
    bf16x8 ah = *(const bf16x8*)(smem + aoff0);
    bf16x8 al = *(const bf16x8*)(smem + aoff0 + 4 * PLANE);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int sn = s + 1 < nsteps ? s + 1 : s;
#ifndef WINO_ABL_NOSTAGE
      stage_load(rd, dy, chan_next);
#endif
      __builtin_amdgcn_sched_barrier(0);
      // half 0: n-tiles 0, 1
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        bf16x8 nh_, nl_;
        if (mt < 7) {
          WINO_LDA(nh_, aoff0 + (2 * (mt + 1) + dy) * NP * 16, ah);
          WINO_LDA(nl_, aoff0 + (2 * (mt + 1) + dy) * NP * 16 + 4 * PLANE, al);
        } else {                                            // (the second half starts over at m-tile 0)
          WINO_LDA(nh_, aoff1 + dy * NP * 16, ah);
          WINO_LDA(nl_, aoff1 + dy * NP * 16 + 4 * PLANE, al);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          WINO_MFMA(al, B0[n][0], acc[mt][n]);
          WINO_MFMA(ah, B0[n][1], acc[mt][n]);
          WINO_MFMA(ah, B0[n][0], acc[mt][n]);
        }
        ah = nh_;
        al = nl_;
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(B0, sn, 0);
      __builtin_amdgcn_sched_barrier(0);
      // half 1: n-tiles 2, 3
#ifndef WINO_ABL_NOSTAGE
      const bool skip = stage_skip(dy);
#endif
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        bf16x8 nh_ = ah, nl_ = al;
        if (mt < 7) {
          WINO_LDA(nh_, aoff1 + (2 * (mt + 1) + dy) * NP * 16, ah);
          WINO_LDA(nl_, aoff1 + (2 * (mt + 1) + dy) * NP * 16 + 4 * PLANE, al);
        } else if (dy < 2) {
          WINO_LDA(nh_, aoff0 + (dy + 1) * NP * 16, ah);
          WINO_LDA(nl_, aoff0 + (dy + 1) * NP * 16 + 4 * PLANE, al);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_ABL_NOSTAGE
        if (!skip) {           // (the last chunk writes the dead slot once more: harmless, and no branch around the loads)
          if (mt == 2) stage_prep(rd, dy, chan_next, 0);
          if (mt == 3) stage_prep(rd, dy, chan_next, 1);
          if (mt == 4) stage_put(rd, dy, slot ^ 1, 0);
          if (mt == 6) stage_put(rd, dy, slot ^ 1, 1);
        }
#endif
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          WINO_MFMA(al, B1[n][0], acc[mt][2 + n]);
          WINO_MFMA(ah, B1[n][1], acc[mt][2 + n]);
          WINO_MFMA(ah, B1[n][0], acc[mt][2 + n]);
        }
        ah = nh_;
        al = nl_;
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(B1, sn, 1);
      __builtin_amdgcn_sched_barrier(0);
      ++s;
    }
    // chunk c + 1 is complete in its slot; every wave has left chunk c's.  (Raw barrier: the fence inside __syncthreads() would
    // drain vmcnt, i.e. wait for the weight fragments just requested for the next step.)
#ifdef WINO_ABL_NOBAR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  }
  }
#undef WA_OFF

  // ---- across-block K split: the partial sums are exchanged in the TRANSFORM domain (the output transform is linear): rows
  // 0 .. S-2 hand their 128 accumulator registers over and leave, row S-1 adds them (row order) and goes on
  if (KS && !FAT && a.ks.S > 1) {
    const int S = a.ks.S, y = (int)blockIdx.y;
    float* const slot0 = a.ks.scratch + (int64_t)wid * (S - 1) * (8 * NTW * 4 * NTH) + tid;
    if (y < S - 1) {
      float* const dst = slot0 + (int64_t)y * (8 * NTW * 4 * NTH);
      KS_SCATTER(acc, 8, NTW, NTH, dst)
      ks_publish(a.ks, wid, y, tid);
      return;
    }
    ks_wait(a.ks, wid, tid);
    KS_GATHER(acc, 8, NTW, NTH, S, slot0);
  }

  // ---- output transform + epilogue: four passes of 4 tile rows (two m-tiles) x all 128 channels.  In a pass EVERY wave puts
  // its accumulators of those two m-tiles into LDS ([position][32 pairs][128 channels + 4]: 32 ds_write_b32 per wave and pass,
  // conflict-free), then all eight waves form y[2p] = m0 + m1 + m2, y[2p+1] = m1 - m2 + m3: a thread takes 4 consecutive
  // channels of one pair, a wave-instruction stores two full 512-byte pixel rows.  Two buffers: one barrier per pass.
  float* const T = (float*)smem;
  float* const red = (float*)smem;                   // [8 waves][128][2]: in pass buffer 0, free once every wave is in / past pass 3
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  constexpr int NML = NTH / 32, NIT = 32 / NML;      // pass-local pair rows covered per item round; item rounds per pass
  const int f32_ = tid & 31, mloc = tid >> 5;        // mloc 0..NML-1 (+NML it): pass-local pair row; channels 4 f32_ .. +3
  const int n0 = ntile * 128 + f32_ * 4;
  const bool nok = n0 < a.cout;
  const int64_t img = (int64_t)b * a.in_h * a.in_w;
  float* const yb = a.y + img * a.out_ld;
  const float* const nzb = a.noise ? a.noise + img : nullptr;
  const float* const rb = a.residual ? a.residual + img * a.res_ld : nullptr;
  // pass ph, item it: pair row m = 32 ph + 16 it + mloc -> tile row m >> 3, pair m & 7
  // bias and noise fetched before the first store (vmcnt counts stores and retires in order: conv_mfma.hip)
  const float4 bv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.y), 0, a.bias ? a.cout * 4 : 0, 0x00020000), n0 * 4, 0, 0));     // (out of range -> zeros)
  // (buffer loads, every request unconditional: a pixel outside the image -- and a launch without noise -- reads out of range and gets
  //  zero.  As conditional global loads hipcc issued them one by one, each with its own wait: found in the nine-product upscale's
  //  epilogue, conv_mfma2.hip, where 32 of them cost 0.3 ms of a 2.6-ms launch)
  float nzv[8 * NIT];
  {
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(nzb ? nzb : (const float*)yb), 0,
                                                                         nzb ? a.in_h * a.in_w * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8 * NIT; ++i) {
      const int m = 32 * (i / (2 * NIT)) + NML * ((i >> 1) % NIT) + mloc;
      const int oy = ty0 + (m >> 3), ox = tx0 + 2 * (m & 7) + (i & 1);
      nzv[i] = a.noise_weight * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                    nrs, (oy < a.in_h && ox < a.in_w) ? (oy * a.in_w + ox) * 4 : WINO_OOB, 0, 0));
    }
  }
  float4 s1a = make_float4(0.f, 0.f, 0.f, 0.f), s2a = s1a;
  const int resm = a.residual ? (res_after ? 2 : 1) : 0;
  constexpr int TROW2 = 132, TXI2 = 32 * TROW2, TBUF2 = 4 * TXI2;     // floats: 67584 B per buffer
  auto epi_passes = [&](auto act_c) __attribute__((always_inline)) {
#pragma clang fp contract(off)   // no fused multiply-add here: every kernel family's epilogue rounds alike
    const int ACT = act_c.value;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      float* tw = T + (ph & 1) * TBUF2 + xi * TXI2 + nh * 64 + r16;
#pragma unroll
      for (int ml = 0; ml < 2; ++ml)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) tw[(ml * 16 + g * 4 + j) * TROW2 + nt * 16] = acc[ph * 2 + ml][nt][j];
      // (LDS only: __syncthreads() also waits for the previous pass's global stores)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float* tr = T + (ph & 1) * TBUF2 + f32_ * 4;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int ml_ = it * NML + mloc, m = ph * 32 + ml_;
        const float4 v0 = *(const float4*)(tr + 0 * TXI2 + ml_ * TROW2), v1 = *(const float4*)(tr + 1 * TXI2 + ml_ * TROW2);
        const float4 v2 = *(const float4*)(tr + 2 * TXI2 + ml_ * TROW2), v3 = *(const float4*)(tr + 3 * TXI2 + ml_ * TROW2);
        const int oy = ty0 + (m >> 3), ox0 = tx0 + 2 * (m & 7);
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          if (nok && oy < a.in_h && ox0 + px < a.in_w) {
            float o[4];
            if (px == 0) { o[0] = (v0.x + v1.x) + v2.x; o[1] = (v0.y + v1.y) + v2.y; o[2] = (v0.z + v1.z) + v2.z; o[3] = (v0.w + v1.w) + v2.w; }
            else         { o[0] = (v1.x - v2.x) + v3.x; o[1] = (v1.y - v2.y) + v3.y; o[2] = (v1.z - v2.z) + v3.z; o[3] = (v1.w - v2.w) + v3.w; }
            const int pix = oy * a.in_w + ox0 + px;
            const float nz = nzv[ph * 2 * NIT + it * 2 + px];
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (resm) rv = *(const float4*)(rb + (pix * a.res_ld + n0));
            const float r4[4] = {rv.x, rv.y, rv.z, rv.w};
            const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              float t = o[cc] + b4[cc] + nz;
              if (resm == 1) t += r4[cc];
              if (ACT == PPST_ACT_LRELU) t = (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
              else if (ACT == PPST_ACT_PRELU) t = t >= 0.f ? t : t * slope;
              if (resm == 2) t += r4[cc];
              o[cc] = t * a.out_scale;
            }
            PPST_EPI_STORE(yb + (pix * a.out_ld + n0), o);
            s1a.x += o[0]; s1a.y += o[1]; s1a.z += o[2]; s1a.w += o[3];
            s2a.x += o[0] * o[0]; s2a.y += o[1] * o[1]; s2a.z += o[2] * o[2]; s2a.w += o[3] * o[3];
          }
        }
      }
    }
  };
#ifdef WINO_ABL_NOEPI
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) asm volatile("" ::"v"(acc[i][j]));
  if (a.B < 0)
#endif
  if (act == PPST_ACT_LRELU) epi_passes(EpiC<PPST_ACT_LRELU>{});
  else if (act == PPST_ACT_PRELU) epi_passes(EpiC<PPST_ACT_PRELU>{});
  else epi_passes(EpiC<PPST_ACT_NONE>{});

  if (a.stats) {
    // a thread's sums cover its 4 channels over 16 pixels; the other lane of the wave with the same channels is lane ^ 32
    float4 s1 = s1a, s2 = s2a;
    s1.x += __shfl_xor(s1.x, 32, 64); s1.y += __shfl_xor(s1.y, 32, 64); s1.z += __shfl_xor(s1.z, 32, 64); s1.w += __shfl_xor(s1.w, 32, 64);
    s2.x += __shfl_xor(s2.x, 32, 64); s2.y += __shfl_xor(s2.y, 32, 64); s2.z += __shfl_xor(s2.z, 32, 64); s2.w += __shfl_xor(s2.w, 32, 64);
    __syncthreads();     // every wave has left its last pass (which read buffer 1; `red` lies in buffer 0)
    if (lane < 32) {
      float* r = red + (wave * 128 + f32_ * 4) * 2;
      r[0] = s1.x; r[1] = s2.x; r[2] = s1.y; r[3] = s2.y; r[4] = s1.z; r[5] = s2.z; r[6] = s1.w; r[7] = s2.w;
    }
    __syncthreads();
    if (tid < 128) {
      const int n = ntile * 128 + tid;
      if (n < a.cout) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { t0 += red[(w * 128 + tid) * 2]; t1 += red[(w * 128 + tid) * 2 + 1]; }
        float* o = a.stats + ((((int64_t)b * a.tiles_y + tyi) * a.tiles_x + txi) * a.cout + n) * 2;
        o[0] = t0;
        o[1] = t1;
      }
    }
  }
}

// Entry used by ppst_conv2d_mfma (conv_mfma.hip) for variant 10.  The caller's promises (as with variant 6): a plain 3x3
// stride-1 table -- nsteps = 9 * chunks, steps[9c].x = first channel of chunk c --, wpack from ppst_conv_pack_wino.
int ppst_conv_wino_launch(const ppst_conv_args* a, int n_tiles, int tiles_y, int tiles_x, hipStream_t st) {
  WinoKArgs k;
  k.x = (const float*)a->x; k.wpack = (const unsigned char*)a->wpack; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu;
  k.stats = (float*)a->stats; k.residual = (const float*)a->residual;
  k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.in_h = a->in_h; k.in_w = a->in_w; k.in_ld = a->in_ld; k.out_ld = a->out_ld; k.cout = a->cout;
  k.nchunk = a->nsteps / 9; k.pad_mode = a->pad_mode; k.act = a->act; k.res_ld = a->res_ld;
  k.tiles_y = tiles_y; k.tiles_x = tiles_x; k.n_tiles = n_tiles;
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu;
  k.in_c = a->in_c; k.in_act = a->in_act;
  const int blocks = n_tiles * a->B * tiles_y * tiles_x;
  k.ks.scratch = nullptr; k.ks.flags = nullptr; k.ks.epoch = 0; k.ks.S = 1;
#ifndef WINO_FAT
  if (a->ksplit > 1) {        // across-block K split: whole chunks per block (k.nchunk % S == 0 follows from nsteps % S with 9-step chunks
                              // only if the caller kept its promise; checked here)
    if ((!a->ksplit_starts && k.nchunk % a->ksplit) || a->ksplit > 4) return PPST_EINVAL;
    const int e0 = ppst_ksplit_prepare_(a->ksplit, a->ksplit_starts, blocks, a->nsteps, 128, 512, st, &k.ks);
    if (e0 != PPST_OK) return e0;
    for (int i = 1; i < a->ksplit; ++i)
      if (k.ks.start[i] % 9) return PPST_EINVAL;           // whole chunks per block
  }
#endif
  if (k.ks.S > 1) {
    const dim3 gridk(blocks, k.ks.S);
    if (k.in_ss && k.in_act != PPST_ACT_NONE) PPST_LAUNCH((conv_wino_kernel<2, false, true>), gridk, dim3(512), 0, st, k);
    else if (k.in_ss) PPST_LAUNCH((conv_wino_kernel<1, false, true>), gridk, dim3(512), 0, st, k);
    else PPST_LAUNCH((conv_wino_kernel<0, false, true>), gridk, dim3(512), 0, st, k);
    return PPST_LAUNCH_CHECK();
  }
  const dim3 grid(blocks);
#ifdef WINO_FAT
  if (k.in_ss && k.in_act != PPST_ACT_NONE) PPST_LAUNCH((conv_wino_kernel<2, true>), dim3(blocks), dim3(256), 0, st, k);
  else if (k.in_ss) PPST_LAUNCH((conv_wino_kernel<1, true>), dim3(blocks), dim3(256), 0, st, k);
  else PPST_LAUNCH((conv_wino_kernel<0, true>), dim3(blocks), dim3(256), 0, st, k);
#else
  if (k.in_ss && k.in_act != PPST_ACT_NONE) PPST_LAUNCH((conv_wino_kernel<2, false>), grid, dim3(512), 0, st, k);
  else if (k.in_ss) PPST_LAUNCH((conv_wino_kernel<1, false>), grid, dim3(512), 0, st, k);
  else PPST_LAUNCH((conv_wino_kernel<0, false>), grid, dim3(512), 0, st, k);
#endif
  return PPST_LAUNCH_CHECK();
}

// ---- weight transform + pack: [n tile of 128][i][channel half][chunk][ky][n-tile 0..3][hi | lo][lane][8 k] bf16
struct WinoPackJob {
  const float* w;
  int64_t sn, sc, sy, sx;
  unsigned short* out;
  int64_t total;
  float scale;
  int cout, nchunk;
};
__global__ __launch_bounds__(256) void conv_pack_wino_kernel(WinoPackJob j) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < j.total; t += (int64_t)gridDim.x * 256) {
    const int lane = (int)(t & 63);
    int64_t r = t >> 6;
    const int ntq = (int)(r & 3); r >>= 2;
    const int ky = (int)(r % 3); r /= 3;
    const int c = (int)(r % j.nchunk); r /= j.nchunk;
    const int nh = (int)(r & 1); r >>= 1;
    const int xi = (int)(r & 3);
    const int ntile = (int)(r >> 2);
    const int n = ntile * 128 + nh * 64 + ntq * 16 + (lane & 15);
    const int c0 = c * 32 + (lane >> 4) * 8;
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      double u = 0.0;
      if (n < j.cout) {
        const float* wp = j.w + n * j.sn + (int64_t)(c0 + q) * j.sc + ky * j.sy;
        const double g0 = (double)wp[0] * (double)j.scale, g1 = (double)wp[j.sx] * (double)j.scale, g2 = (double)wp[2 * j.sx] * (double)j.scale;
        u = xi == 0 ? g0 : xi == 1 ? 0.5 * (g0 + g1 + g2) : xi == 2 ? 0.5 * (g0 - g1 + g2) : g2;
      }
      const float uf = (float)u;
      hi[q] = f2bf(uf);
      lo[q] = f2bf((float)(u - (double)bf2f(hi[q])));
    }
    unsigned short* oh = j.out + (((t >> 6) * 2) * 64 + lane) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) oh[q] = hi[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) oh[64 * 8 + q] = lo[q];
  }
}
extern "C" int64_t ppst_conv_pack_wino_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cin % 32) return 0;
  return (int64_t)cdiv(cout, 128) * 8 * (cin / 32) * 3 * WINO_STEP_BYTES;
}
extern "C" int ppst_conv_pack_wino(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int cin,
                                   void* out, void* stream) {
  if (cout <= 0 || cin <= 0 || cin % 32) return PPST_EINVAL;
  if (!w || !out) return PPST_ENULL;
  WinoPackJob j;
  j.w = (const float*)w; j.sn = sn; j.sc = sc; j.sy = sy; j.sx = sx; j.out = (unsigned short*)out; j.scale = scale;
  j.cout = cout; j.nchunk = cin / 32;
  j.total = (int64_t)cdiv(cout, 128) * 8 * j.nchunk * 3 * 4 * 64;
  int64_t blocks = cdiv64(j.total, 256);
  if (blocks > 4096) blocks = 4096;
  PPST_LAUNCH(conv_pack_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), j);
  return PPST_LAUNCH_CHECK();
}
