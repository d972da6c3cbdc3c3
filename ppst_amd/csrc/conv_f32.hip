// Exact-fp32 variant of the fused implicit-GEMM convolution (precision = 2): the same step tables, padding modes,
// normalise-on-load, scattered output phases and epilogue (bias + noise [+ residual] -> act -> * out_scale, tile
// statistics) as conv_mfma.hip, computed with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf chain,
// MI355X_MICROARCH.md "Matrix cores") straight from the fp32 weight tensor.
//
// Purpose: a verification path.  The production kernel splits fp32 into bf16 hi + lo (16 mantissa bits, ~8e-6 per
// operand); gradients that are heavily cancelling sums (bias / noise-weight gradients in front of an instance norm,
// PReLU slopes) amplify that to ~1e-2.  Running the SAME train step on this kernel separates rounding from defects:
// with it the gradients meet the fp32 reference to its own float32 noise.  1/16 of the bf16 MFMA rate and no LDS
// staging (operands come from L1/L2): 10-30x slower than conv_mfma_kernel -- never the measured path.
#include "common.h"

struct ConvF32Args {
  const float* x; const float* w; const int4* steps; float* y;
  const float* bias; const float* noise; const float* prelu; float* stats; const float* residual;
  const int* src_c; const int* src_ky; const int* src_kx;
  int64_t sn, sc, sy, sx;
  float wscale, noise_weight, out_scale;
  int B, in_h, in_w, in_ld, out_h, out_w, out_ld, cout;
  int nsteps, n_groups, pad_mode, in_off_y, in_off_x, out_sy, out_sx, act, res_ld, tile_h, tile_w;
  int tiles_y, tiles_x, n_tiles;
  const float* in_ss; const float* in_prelu; int in_c, in_act;
};

__device__ __forceinline__ int f32_pad_index(int i, int n, int mode) {
  if (mode == PPST_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// block = 256 threads = 4 waves; block tile = 16 x 16 output pixels x 32 output channels; wave w owns rows 4w..4w+3
// as two 32-pixel M tiles (rows 4w+2m, 4w+2m+1).  MFMA 32x32x2: A lane l -> (pixel l%32, k l/32), B lane l -> (n l%32, k l/32).
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Args a) {
  const int m_count = a.B * a.tiles_y * a.tiles_x;
  const int nidx = blockIdx.x / m_count;
  int midx = blockIdx.x - nidx * m_count;
  const int group = nidx / a.n_tiles, ntile = nidx - group * a.n_tiles;
  const int b = midx / (a.tiles_y * a.tiles_x);
  midx -= b * a.tiles_y * a.tiles_x;
  const int tyi = midx / a.tiles_x, txi = midx - tyi * a.tiles_x;
  const int ty0 = tyi * 16, tx0 = txi * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int n = ntile * 32 + li;
  const bool nok = n < a.cout;
  const float* xb = a.x + (int64_t)b * a.in_h * a.in_w * a.in_ld;
  const float in_slope = (a.in_ss && a.in_act == PPST_ACT_PRELU && a.in_prelu) ? a.in_prelu[0] : 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  // this lane's A pixel in each M tile: row 4*wave + 2*m + (li >> 4), column li & 15
  const int prow[2] = {ty0 + wave * 4 + (li >> 4), ty0 + wave * 4 + 2 + (li >> 4)};
  const int pcol = tx0 + (li & 15);
  for (int s = 0; s < a.nsteps; ++s) {
    const int gs = group * a.nsteps + s;
    const int4 d = a.steps[gs];
    const int sc0 = a.src_c[gs];
    if (sc0 < 0) continue;                                   // zero-weight padding step
    const float* wp = a.w + (nok ? (int64_t)n * a.sn : 0) + (int64_t)sc0 * a.sc + a.src_ky[gs] * a.sy + a.src_kx[gs] * a.sx;
    const float* xp[2];
    bool ok[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      int iy = prow[m] + d.y + a.in_off_y, ix = pcol + d.z + a.in_off_x;
      const bool inb = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
      ok[m] = inb || a.pad_mode != PPST_PAD_ZERO;
      iy = f32_pad_index(iy, a.in_h, a.pad_mode);
      ix = f32_pad_index(ix, a.in_w, a.pad_mode);
      xp[m] = xb + ((int64_t)iy * a.in_w + ix) * a.in_ld + d.x;
    }
    for (int k = 0; k < 32; k += 2) {
      const int kk = k + lk;
      const float bv = nok ? wp[(int64_t)kk * a.sc] * a.wscale : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float av = ok[m] ? xp[m][kk] : 0.f;
        if (a.in_ss && ok[m]) {
          const float* q = a.in_ss + ((int64_t)b * a.in_c + d.x + kk) * 2;
          av = q[0] * av + q[1];
          if (a.in_act == PPST_ACT_LRELU) av = (av > 0.f ? av : av * 0.2f) * 1.41421356237309515f;
          else if (a.in_act == PPST_ACT_PRELU) av = av >= 0.f ? av : av * in_slope;
        }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
      }
    }
  }
  // epilogue: D tile col = li (channel), row = (reg&3) + 8*(reg>>2) + 4*lk (pixel of the M tile)
  const int gy = group >> 1, gx = group & 1;
  const int act = a.act & 0xff;
  const bool res_after = (a.act >> 8) & 1;
  const float slope = (act == PPST_ACT_PRELU && a.prelu) ? a.prelu[0] : 0.f;
  const float bv = (nok && a.bias) ? a.bias[n] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int p = (rg & 3) + 8 * (rg >> 2) + 4 * lk;
      const int ty = ty0 + wave * 4 + 2 * m + (p >> 4), tx = tx0 + (p & 15);
      if (!nok || ty >= a.tile_h || tx >= a.tile_w) continue;
      const int oy = ty * a.out_sy + (a.n_groups > 1 ? gy : 0), ox = tx * a.out_sx + (a.n_groups > 1 ? gx : 0);
      if (oy >= a.out_h || ox >= a.out_w) continue;
      const int64_t opix = ((int64_t)b * a.out_h + oy) * a.out_w + ox;
      float t = acc[m][rg] + bv + (a.noise ? a.noise_weight * a.noise[opix] : 0.f);
      const float r = a.residual ? a.residual[opix * a.res_ld + n] : 0.f;
      if (!res_after) t += r;
      if (act == PPST_ACT_LRELU) t = (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
      else if (act == PPST_ACT_PRELU) t = t >= 0.f ? t : t * slope;
      if (res_after) t += r;
      t *= a.out_scale;
      a.y[opix * a.out_ld + n] = t;
      s1 += t;
      s2 += t * t;
    }
  if (a.stats) {
    __shared__ float red[4][32][2];
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lk == 0) { red[wave][li][0] = s1; red[wave][li][1] = s2; }
    __syncthreads();
    if (threadIdx.x < 32 && ntile * 32 + threadIdx.x < a.cout) {
      const int t = threadIdx.x;
      float* o = a.stats + ((((int64_t)b * a.n_groups + group) * (a.tiles_y * a.tiles_x) + tyi * a.tiles_x + txi) * a.cout + ntile * 32 + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
    }
  }
}

extern "C" int ppst_conv2d_f32(const ppst_conv_args* a, const void* w, int64_t sn, int64_t sc, int64_t sy, int64_t sx, float wscale,
                               const int32_t* src_c, const int32_t* src_ky, const int32_t* src_kx, void* stream) {
  if (!a) return PPST_ENULL;
  if (a->B < 0 || a->in_h <= 0 || a->in_w <= 0 || a->in_ld <= 0 || a->out_h <= 0 || a->out_w <= 0 || a->out_ld < a->cout || a->cout <= 0 ||
      a->nsteps <= 0 || (a->n_groups != 1 && a->n_groups != 4) || a->pad_mode < 0 || a->pad_mode > 2 || a->tile_h <= 0 || a->tile_w <= 0 ||
      a->out_sy <= 0 || a->out_sx <= 0 || (a->residual && a->res_ld < a->cout) || (a->in_scale_shift && a->in_c <= 0))
    return PPST_EINVAL;
  if ((a->tile_h - 1) * a->out_sy >= a->out_h || (a->tile_w - 1) * a->out_sx >= a->out_w) return PPST_EINVAL;
  // same limits as ppst_conv2d_mfma: 16-row tiles (the statistics layout assumes them) and 32-bit offsets inside one image
  if (a->tile_rows != 16) return PPST_EINVAL;
  {
    const int64_t px = (int64_t)a->out_h * a->out_w + 64 * (int64_t)a->out_sx;
    if (px * a->out_ld > 0x7fffffff || (a->residual && px * a->res_ld > 0x7fffffff)) return PPST_EINVAL;
    if ((int64_t)a->in_h * a->in_w * a->in_ld * 4 > 0x7fffffff) return PPST_EINVAL;
  }
  if (a->B == 0) return PPST_OK;
  if (!a->x || !w || !a->steps || !a->y || !src_c || !src_ky || !src_kx) return PPST_ENULL;
  ConvF32Args k;
  k.x = (const float*)a->x; k.w = (const float*)w; k.steps = (const int4*)a->steps; k.y = (float*)a->y;
  k.bias = (const float*)a->bias; k.noise = (const float*)a->noise; k.prelu = (const float*)a->prelu; k.stats = (float*)a->stats;
  k.residual = (const float*)a->residual; k.src_c = src_c; k.src_ky = src_ky; k.src_kx = src_kx;
  k.sn = sn; k.sc = sc; k.sy = sy; k.sx = sx; k.wscale = wscale; k.noise_weight = a->noise_weight; k.out_scale = a->out_scale;
  k.B = a->B; k.in_h = a->in_h; k.in_w = a->in_w; k.in_ld = a->in_ld; k.out_h = a->out_h; k.out_w = a->out_w; k.out_ld = a->out_ld;
  k.cout = a->cout; k.nsteps = a->nsteps; k.n_groups = a->n_groups; k.pad_mode = a->pad_mode; k.in_off_y = a->in_off_y;
  k.in_off_x = a->in_off_x; k.out_sy = a->out_sy; k.out_sx = a->out_sx; k.act = a->act; k.res_ld = a->res_ld; k.tile_h = a->tile_h;
  k.tile_w = a->tile_w; k.tiles_y = cdiv(a->tile_h, 16); k.tiles_x = cdiv(a->tile_w, 16); k.n_tiles = cdiv(a->cout, 32);
  k.in_ss = (const float*)a->in_scale_shift; k.in_prelu = (const float*)a->in_prelu; k.in_c = a->in_c; k.in_act = a->in_act;
  const int64_t blocks = (int64_t)a->n_groups * k.n_tiles * a->B * k.tiles_y * k.tiles_x;
  if (blocks > 0x7fffffff) return PPST_EINVAL;
  PPST_LAUNCH(conv_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k);
  return PPST_LAUNCH_CHECK();
}
