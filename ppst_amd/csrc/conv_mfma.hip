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
//  * Pipeline: register-prefetch double buffering -- global loads for step s+1 are issued
//    before the MFMAs of step s and written to the other LDS buffer after them; one
//    __syncthreads per K-step (= per tap), >= 48 MFMAs per wave between barriers.
//  * Stride-2 convs run as stride-1 convs over a space-to-depth input, the 4x4 stride-2
//    transposed conv as 4 output-phase groups of 2x2 taps; both are just step tables.
//  * Block -> tile map is XCD-aware: the 8 XCDs each get a contiguous range of an N-major
//    ordering so that co-resident blocks of one XCD stream the same weight blobs from its L2.
#include "common.h"

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
};

__device__ __forceinline__ int pad_index(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  // replicate, and the safety clamp for reflect on far-out (masked) tile pixels
  i = i < 0 ? 0 : (i >= n ? n - 1 : i);
  return i;
}

template <int WM, int WN, int HALO, bool X3>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvKArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TH = 4 * WM, TW = 16;
  constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO, HP = HH * HW;
  constexpr int PLANE = ((HP * 16 + 255) / 256) * 256;  // bytes
  constexpr int NPL = X3 ? 8 : 4;                        // planes per A buffer (hi g0..3, lo g0..3)
  constexpr int ABUF = NPL * PLANE;
  constexpr int BN = 64 * WN;
  constexpr int BPLANE = BN * 16;
  constexpr int BBUF = NPL * BPLANE;
  constexpr int A_ITEMS = HP * 8;                        // float4 items per chunk
  constexpr int A_IT = (A_ITEMS + NT - 1) / NT;
  constexpr int B_ITEMS = BBUF / 16;
  constexpr int B_IT = (B_ITEMS + NT - 1) / NT;
  static_assert(B_ITEMS % NT == 0, "B blob must split evenly");

  __shared__ __attribute__((aligned(256))) unsigned char smem[2 * ABUF + 2 * BBUF];
  unsigned char* smA = smem;
  unsigned char* smB = smem + 2 * ABUF;

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

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int r16 = lane & 15, g = lane >> 4;

  const int4* steps = a.steps + (int64_t)group * a.nsteps;
  const unsigned char* wblob = (const unsigned char*)a.wpack + ((int64_t)nidx * a.nsteps) * BBUF;
  const float* xb = a.x + (int64_t)b * a.in_h * a.in_w * a.in_ld;

  static_assert(B_IT == 1 || B_IT == 2, "B staging assumes 1 or 2 16-B items per thread");
  float4 ra[A_IT];
  uint4 rb0, rb1;  // (named, not an array: hipcc promoted a 2-element array to LDS)

  auto a_load = [&](int chan_off) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int i = tid + it * NT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < A_ITEMS) {
        int pix = i >> 3, q4 = i & 7;
        int hy = pix / HW, hx = pix - hy * HW;
        int iy = ty0 + hy - HALO + a.in_off_y, ix = tx0 + hx - HALO + a.in_off_x;
        bool inb = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
        if (inb || a.pad_mode != PPST_PAD_ZERO) {
          iy = pad_index(iy, a.in_h, a.pad_mode);
          ix = pad_index(ix, a.in_w, a.pad_mode);
          v = *(const float4*)(xb + ((int64_t)iy * a.in_w + ix) * a.in_ld + chan_off + q4 * 4);
        }
      }
      ra[it] = v;
    }
  };
  auto a_store = [&](int buf) {
    unsigned char* base = smA + buf * ABUF;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int i = tid + it * NT;
      if (i < A_ITEMS) {
        int pix = i >> 3, q4 = i & 7;
        float4 v = ra[it];
        unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
        split_bf16(v.x, h0, l0); split_bf16(v.y, h1, l1); split_bf16(v.z, h2, l2); split_bf16(v.w, h3, l3);
        int off = (q4 >> 1) * PLANE + pix * 16 + (q4 & 1) * 8;
        uint2 hv = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        *(uint2*)(base + off) = hv;
        if (X3) {
          uint2 lv = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
          *(uint2*)(base + 4 * PLANE + off) = lv;
        }
      }
    }
  };
  auto b_load = [&](int s) {
    const uint4* src = (const uint4*)(wblob + (int64_t)s * BBUF);
    rb0 = src[tid];
    if (B_IT > 1) rb1 = src[tid + NT];
  };
  auto b_store = [&](int buf) {
    uint4* dst = (uint4*)(smB + buf * BBUF);
    dst[tid] = rb0;
    if (B_IT > 1) dst[tid + NT] = rb1;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int4 st = steps[0];
  a_load(st.x);
  b_load(0);
  a_store(0);
  b_store(0);
  __syncthreads();
  int curA = 0;

  for (int s = 0; s < a.nsteps; ++s) {
    const bool has_next = s + 1 < a.nsteps;
    int4 nx = st;
    if (has_next) nx = steps[s + 1];
    const bool nextA = has_next && nx.w != 0;
    if (has_next) b_load(s + 1);
    if (nextA) a_load(nx.x);

    // ---- MFMA over this tap: A window shifted by (dy, dx) inside the halo tile
    {
      const unsigned char* Ab = smA + curA * ABUF + g * PLANE;
      const unsigned char* Bb = smB + (s & 1) * BBUF + g * BPLANE + (wn * 64 + r16) * 16;
      const int pix0 = (wm * 4 + HALO + st.y) * HW + (HALO + st.z) + r16;
      bf16x8 bh[4], bl[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bh[nt] = *(const bf16x8*)(Bb + nt * 256);
        if (X3) bl[nt] = *(const bf16x8*)(Bb + 4 * BPLANE + nt * 256);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int off = (pix0 + mt * HW) * 16;
        bf16x8 ah = *(const bf16x8*)(Ab + off);
        bf16x8 al;
        if (X3) al = *(const bf16x8*)(Ab + 4 * PLANE + off);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (X3) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }

    if (has_next) b_store((s + 1) & 1);
    if (nextA) a_store(curA ^ 1);
    __syncthreads();
    if (nextA) curA ^= 1;
    st = nx;
  }

  // ---- epilogue: + bias + noise [+ residual] -> act -> * out_scale -> store, tile statistics
  const int gy = group >> 1, gx = group & 1;  // output phase of the transposed conv
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;  // residual joins after the activation (resnet skip)
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  float ssum[4], ssq[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = ntile * BN + wn * 64 + nt * 16 + r16;
    const bool nok = n < a.cout;
    const float bv = (nok && a.bias) ? a.bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int ty = ty0 + wm * 4 + mt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tx = tx0 + g * 4 + j;
        if (nok && ty < a.tile_h && tx < a.tile_w) {
          const int oy = ty * a.out_sy + (a.n_groups > 1 ? gy : 0), ox = tx * a.out_sx + (a.n_groups > 1 ? gx : 0);
          const int64_t opix = ((int64_t)b * a.out_h + oy) * a.out_w + ox;
          float v = acc[mt][nt][j] + bv;
          if (a.noise) v += a.noise_weight * a.noise[opix];
          float rv = a.residual ? a.residual[opix * a.res_ld + n] : 0.f;
          if (!res_after) v += rv;
          if (act == PPST_ACT_LRELU) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
          else if (act == PPST_ACT_PRELU) v = v >= 0.f ? v : v * slope;
          if (res_after) v += rv;
          v *= a.out_scale;
          a.y[opix * a.out_ld + n] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    ssum[nt] = s1;
    ssq[nt] = s2;
  }
  if (a.stats) {
    // reduce over the 4 lane groups (pixels), then over the WM waves through LDS
    float* red = (float*)smem;  // [WM][BN][2]  (main-loop buffers are dead: last barrier passed)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float s1 = ssum[nt], s2 = ssq[nt];
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) {
        int nl = wn * 64 + nt * 16 + r16;
        red[(wm * BN + nl) * 2] = s1;
        red[(wm * BN + nl) * 2 + 1] = s2;
      }
    }
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
}

// ------------------------------------------------------------ weight packing --
// out[group][ntile][step][hilo][g][n_local][j] = split_bf16(scale * w[n][src_c+8g+j][ky][kx])
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, int64_t sn, int64_t sc, int64_t sy,
                                                        int64_t sx, float scale, int cout, int bn,
                                                        const int* __restrict__ src_c, const int* __restrict__ src_ky,
                                                        const int* __restrict__ src_kx, int nsteps, int n_groups, int x3,
                                                        unsigned short* __restrict__ out, int64_t total) {
  const int n_tiles = (cout + bn - 1) / bn;
  const int npl = x3 ? 8 : 4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int nl = (int)(t % bn);
    int64_t r = t / bn;
    int g = (int)(r % 4); r /= 4;
    int s = (int)(r % nsteps); r /= nsteps;
    int ntile = (int)(r % n_tiles);
    int group = (int)(r / n_tiles);
    int n = ntile * bn + nl;
    int gs = group * nsteps + s;
    int c0 = src_c[gs] + 8 * g, ky = src_ky[gs], kx = src_kx[gs];
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (n < cout) v = w[n * sn + (int64_t)(c0 + j) * sc + ky * sy + kx * sx] * scale;
      split_bf16(v, hi[j], lo[j]);
    }
    int64_t blob = (((int64_t)group * n_tiles + ntile) * nsteps + s) * ((int64_t)npl * bn * 8);
    unsigned short* oh = out + blob + ((int64_t)g * bn + nl) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) oh[j] = hi[j];
    if (x3) {
      unsigned short* ol = oh + (int64_t)4 * bn * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) ol[j] = lo[j];
    }
  }
}

extern "C" int ppst_conv_pack(const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float scale, int cout, int bn,
                              const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, int nsteps, int n_groups,
                              int precision, void* out, void* stream) {
  if (cout <= 0 || (bn != 64 && bn != 128) || nsteps <= 0 || (n_groups != 1 && n_groups != 4) || precision < 0 || precision > 1)
    return PPST_EINVAL;
  if (!w || !src_c || !src_ky || !src_kx || !out) return PPST_ENULL;
  int n_tiles = (cout + bn - 1) / bn;
  int64_t total = (int64_t)n_groups * n_tiles * nsteps * 4 * bn;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)w, sn, sc, sy, sx,
                     scale, cout, bn, src_c, src_ky, src_kx, nsteps, n_groups, precision == 0 ? 1 : 0, (unsigned short*)out, total);
  return PPST_LAUNCH_CHECK();
}

// EqualizedConv2d fused-upscale weight (stylegan2_layers.py:314-319):
// w (Cout,Cin,3,3) -> out (Cin,Cout,4,4) = sum of the 4 unit shifts of the zero-padded kernel
__global__ __launch_bounds__(256) void upscale_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                                             float scale, int64_t total) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int kx = (int)(t & 3), ky = (int)((t >> 2) & 3);
    int64_t r = t >> 4;
    int n = (int)(r % cout), c = (int)(r / cout);
    const float* wp = w + ((int64_t)n * cin + c) * 9;
    auto at = [&](int y, int x) -> float { return (y >= 0 && y < 3 && x >= 0 && x < 3) ? wp[y * 3 + x] * scale : 0.f; };
    // padded p[y][x] = w[y-1][x-1]; out[ky][kx] = p[ky+1][kx+1] + p[ky][kx+1] + p[ky+1][kx] + p[ky][kx]
    out[t] = at(ky, kx) + at(ky - 1, kx) + at(ky, kx - 1) + at(ky - 1, kx - 1);
  }
}
extern "C" int ppst_upscale_weight(const void* w, void* out, int cout, int cin, float scale, void* stream) {
  if (cout <= 0 || cin <= 0) return PPST_EINVAL;
  if (!w || !out) return PPST_ENULL;
  int64_t total = (int64_t)cin * cout * 16;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(upscale_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)w, (float*)out,
                     cout, cin, scale, total);
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------------------- profiling --
#define PROF_MAX 4096
static int g_prof_on = 0;
static hipEvent_t g_ev[PROF_MAX][2];
static double g_flop[PROF_MAX];
static int g_ev_made = 0, g_ev_used = 0;

extern "C" int ppst_prof_enable(int on) {
  g_prof_on = on;
  g_ev_used = 0;
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

extern "C" int ppst_conv_tiles(int tile_h, int tile_w) { return cdiv(tile_h, 16) * cdiv(tile_w, 16); }

template <int WM, int WN, int HALO, bool X3>
static void launch_conv(const ConvKArgs& k, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, HALO, X3>), dim3(blocks), dim3(64 * WM * WN), 0, st, k);
}

extern "C" int ppst_conv2d_mfma(const ppst_conv_args* a, void* stream) {
  if (!a) return PPST_ENULL;
  if (a->B < 0 || a->in_h <= 0 || a->in_w <= 0 || a->in_ld <= 0 || a->in_ld % 4 || a->out_h <= 0 || a->out_w <= 0 ||
      a->out_ld < a->cout || a->cout <= 0 || a->nsteps <= 0 || (a->n_groups != 1 && a->n_groups != 4) || a->pad_mode < 0 ||
      a->pad_mode > 2 || a->tile_h <= 0 || a->tile_w <= 0 || a->out_sy <= 0 || a->out_sx <= 0 || a->precision < 0 ||
      a->precision > 1 || a->halo < 0 || a->halo > 1 || (a->bn != 64 && a->bn != 128) || (a->residual && a->res_ld < a->cout))
    return PPST_EINVAL;
  // the scattered output must stay inside the output tensor
  if ((a->tile_h - 1) * a->out_sy + (a->n_groups > 1 ? 1 : 0) >= a->out_h ||
      (a->tile_w - 1) * a->out_sx + (a->n_groups > 1 ? 1 : 0) >= a->out_w)
    return PPST_EINVAL;
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
  k.tiles_y = cdiv(a->tile_h, 16); k.tiles_x = cdiv(a->tile_w, 16);
  k.n_tiles = cdiv(a->cout, a->bn);
  int64_t blocks64 = (int64_t)a->n_groups * k.n_tiles * a->B * k.tiles_y * k.tiles_x;
  if (blocks64 > 0x7fffffff) return PPST_EINVAL;
  int blocks = (int)blocks64;
  hipStream_t st = as_stream(stream);
  int slot = -1;
  if (g_prof_on && g_ev_used < PROF_MAX) {
    while (g_ev_made <= g_ev_used) {
      if (hipEventCreate(&g_ev[g_ev_made][0]) != hipSuccess || hipEventCreate(&g_ev[g_ev_made][1]) != hipSuccess) return PPST_EINVAL;
      ++g_ev_made;
    }
    slot = g_ev_used++;
    g_flop[slot] = 2.0 * 32.0 * a->nsteps * (double)a->n_groups * a->cout * (double)a->B * a->tile_h * a->tile_w;
    (void)hipEventRecord(g_ev[slot][0], st);
  }
  const bool x3 = a->precision == 0;
  if (a->bn == 128) {
    if (a->halo) { if (x3) launch_conv<4, 2, 1, true>(k, blocks, st); else launch_conv<4, 2, 1, false>(k, blocks, st); }
    else         { if (x3) launch_conv<4, 2, 0, true>(k, blocks, st); else launch_conv<4, 2, 0, false>(k, blocks, st); }
  } else {
    if (a->halo) { if (x3) launch_conv<4, 1, 1, true>(k, blocks, st); else launch_conv<4, 1, 1, false>(k, blocks, st); }
    else         { if (x3) launch_conv<4, 1, 0, true>(k, blocks, st); else launch_conv<4, 1, 0, false>(k, blocks, st); }
  }
  int e = PPST_LAUNCH_CHECK();
  if (slot >= 0) (void)hipEventRecord(g_ev[slot][1], st);
  return e;
}
