// HBM-bound glue kernels of the PPST path on NHWC fp32 activations:
// layout transposes, instance-norm statistics / finalize / apply (+StyleMod, +residual,
// +activation), pooling, bilinear resize, small-channel 1x1 convs, lerp, tensor2im.
// Each is a single pass at 16 B per lane; algorithmic bytes are listed in DESIGN.md.
#include "common.h"

#define GRID_CAP (256 * 16)
static inline unsigned grid_for(int64_t work, int threads = 256) {
  int64_t b = cdiv64(work, threads);
  if (b > GRID_CAP) b = GRID_CAP;
  if (b < 1) b = 1;
  return (unsigned)b;
}

__device__ __forceinline__ float act_apply(float t, int act, float slope);

// residual that is a half-resolution tensor, bilinearly upsampled x2 on the fly
// (F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) of the resnet skip,
// generator.py:75): output pixel (oy, ox) of an H x W image, res is [B][H/2][W/2][res_ld]
template <int ST = PPST_ST_F32>
__device__ __forceinline__ float4 res_up2_sample(const void* __restrict__ res, int b, int oy, int ox, int H, int W, int res_ld,
                                                 int c) {
  const int h = H >> 1, w = W >> 1;
  float fy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.f);
  int y0 = (int)fy, x0 = (int)fx;
  int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
  float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const int64_t base = (int64_t)b * h * w * res_ld + c;
  float4 v00 = st_ld4<ST>(res, base + ((int64_t)y0 * w + x0) * res_ld);
  float4 v01 = st_ld4<ST>(res, base + ((int64_t)y0 * w + x1) * res_ld);
  float4 v10 = st_ld4<ST>(res, base + ((int64_t)y1 * w + x0) * res_ld);
  float4 v11 = st_ld4<ST>(res, base + ((int64_t)y1 * w + x1) * res_ld);
  float4 o;
  o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
  o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
  o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
  o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  return o;
}

// ------------------------------------------------------------------ layout --
// x[b][c][p] <-> y[b][p][c]; 32(c) x 64(p) tiles through LDS (pad 1: conflict-free)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t P) {
  __shared__ float sm[32][65];
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int b = blockIdx.z;
  const float* xb = x + (int64_t)b * C * P;
  float* yb = y + (int64_t)b * C * P;
  {
    int tp = threadIdx.x & 63, tc = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int c = c0 + tc + i * 4;
      int64_t p = p0 + tp;
      sm[tc + i * 4][tp] = (c < C && p < P) ? xb[(int64_t)c * P + p] : 0.f;
    }
  }
  __syncthreads();
  {
    int tc = threadIdx.x & 31, tp = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int c = c0 + tc;
      int64_t p = p0 + tp + i * 8;
      if (c < C && p < P) yb[p * C + c] = sm[tc][tp + i * 8];
    }
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t P) {
  __shared__ float sm[64][33];
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int b = blockIdx.z;
  const float* xb = x + (int64_t)b * C * P;
  float* yb = y + (int64_t)b * C * P;
  {
    int tc = threadIdx.x & 31, tp = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int c = c0 + tc;
      int64_t p = p0 + tp + i * 8;
      sm[tp + i * 8][tc] = (c < C && p < P) ? xb[p * C + c] : 0.f;
    }
  }
  __syncthreads();
  {
    int tp = threadIdx.x & 63, tc = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int c = c0 + tc + i * 4;
      int64_t p = p0 + tp;
      if (c < C && p < P) yb[(int64_t)c * P + p] = sm[tp][tc + i * 4];
    }
  }
}
extern "C" int ppst_nchw_to_nhwc(const void* x, void* y, int B, int C, int H, int W, void* stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t P = (int64_t)H * W;
  dim3 grid((unsigned)cdiv64(P, 64), cdiv(C, 32), B);
  PPST_LAUNCH(nchw_to_nhwc_kernel, grid, dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, C, P);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_nhwc_to_nchw(const void* x, void* y, int B, int C, int H, int W, void* stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t P = (int64_t)H * W;
  dim3 grid((unsigned)cdiv64(P, 64), cdiv(C, 32), B);
  PPST_LAUNCH(nhwc_to_nchw_kernel, grid, dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, C, P);
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------- per-(b,c) reductions --
// One block reduces PIX_CHUNK pixels of one image for all channels and writes a
// partial (v0, v1) per channel: MODE 0 = (sum w*x, sum w*x^2) [instance norm; w = border
// multiplicity when the statistics are those of the ReplicationPad2d(1)-padded tensor],
// MODE 1 = (sum m*x, max m*x) [GAP/GMP with optional mask].
// Pixels per block: 1024 for big images, fewer when that would leave CUs idle (>= 2048 blocks
// wanted, >= 64 pixels per block).  n_partials = ceil(H*W / pix_chunk(B, H*W)) everywhere.
// The chunk depends on the image size ONLY (not on the batch): the grouping of the partial sums -- and so every
// bit of the statistics -- is the same whether an image is processed alone or inside any batch (the image-sharded grid
// evaluator relies on it: N ranks reproduce one rank bit for bit).
static inline int pix_chunk(int B, int64_t hw) {
  (void)B;
  int chunk = 1024;
  while (chunk > 64 && cdiv64(hw, chunk) < 2048) chunk >>= 1;
  return chunk;
}
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          float* __restrict__ partial, int H, int W, int C, int ld,
                                                          int rep_pad, int nchunks, int PIX_CHUNK) {
  __shared__ float s0[256], s1[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int P = H * W;
  const int pbeg = chunk * PIX_CHUNK;
  const int pend = (pbeg + PIX_CHUNK < P) ? pbeg + PIX_CHUNK : P;
  int lanesC = 1;
  while (lanesC < C && lanesC < 256) lanesC <<= 1;
  const int rows = 256 / lanesC;
  const int cl = threadIdx.x % lanesC, pr = threadIdx.x / lanesC;
  const float* xb = x + (int64_t)b * P * ld;
  const float* mb = mask ? mask + (int64_t)b * P : nullptr;
  for (int cbase = 0; cbase < C; cbase += lanesC) {
    int c = cbase + cl;
    float a0 = 0.f, a1 = (MODE == 1) ? -INFINITY : 0.f;
    if (c < C) {
      for (int p = pbeg + pr; p < pend; p += rows) {
        float v = xb[(int64_t)p * ld + c];
        if (MODE == 0) {
          float w = 1.f;
          if (rep_pad) {
            int py = p / W, px = p - py * W;
            w = (float)((1 + (py == 0) + (py == H - 1)) * (1 + (px == 0) + (px == W - 1)));
          }
          a0 += w * v;
          a1 += w * v * v;
        } else {
          if (mb) v *= mb[p];
          a0 += v;
          a1 = fmaxf(a1, v);
        }
      }
    }
    s0[threadIdx.x] = a0;
    s1[threadIdx.x] = a1;
    __syncthreads();
    if (pr == 0 && c < C) {
      for (int r = 1; r < rows; ++r) {
        a0 += s0[r * lanesC + cl];
        a1 = (MODE == 1) ? fmaxf(a1, s1[r * lanesC + cl]) : a1 + s1[r * lanesC + cl];
      }
      float* o = partial + (((int64_t)b * nchunks + chunk) * C + c) * 2;
      o[0] = a0;
      o[1] = a1;
    }
    __syncthreads();
  }
}

// 16-B-per-lane version of the reduction (C % 4 == 0): lane = 4 channels, the block's other
// threads walk different pixel rows; optionally (APPLY) the elements are first transformed
// like ppst_affine_act and stored, so a producer's apply pass also yields the statistics of
// its output (no separate read pass for the instance norm that follows).
struct ApplyArgs {
  const float* ss; const float* res; const float* rss; float* y; const float* prelu;
  int res_ld, y_ld, actf; float out_scale;
  int res_up2;  // residual is half-resolution, bilinearly upsampled x2 on the fly
};
template <int MODE, bool APPLY, int XS = PPST_ST_F32>
__global__ __launch_bounds__(256) void chan_reduce4_kernel(const void* __restrict__ x, const float* __restrict__ mask,
                                                           float* __restrict__ partial, int H, int W, int C, int ld,
                                                           int rep_pad, int nchunks, ApplyArgs ap, int PIX_CHUNK, FastDiv d_w) {
  __shared__ float4 s0[256], s1[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int P = H * W;
  const int pbeg = chunk * PIX_CHUNK;
  const int pend = (pbeg + PIX_CHUNK < P) ? pbeg + PIX_CHUNK : P;
  const int c4n = C >> 2;
  int lanes = 1;
  while (lanes < c4n && lanes < 256) lanes <<= 1;
  const int rows = 256 / lanes;
  const int cl = threadIdx.x % lanes, pr = threadIdx.x / lanes;
  const float* mb = mask ? mask + (int64_t)b * P : nullptr;
  const int act = ap.actf & 0xff;
  const bool res_first = (ap.actf >> 8) & 1;
  const float slope = (APPLY && act == PPST_ACT_PRELU && ap.prelu) ? ap.prelu[0] : 0.f;
  for (int cbase = 0; cbase < c4n; cbase += lanes) {
    const int c = (cbase + cl) * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a1 = (MODE == 1) ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : a0;
    if (c < C) {
      float4 sa = make_float4(1.f, 1.f, 1.f, 1.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 ra = sa, rb = sb;
      if (APPLY && ap.ss) {
        const float4* q = (const float4*)(ap.ss + ((int64_t)b * C + c) * 2);
        float4 q0 = q[0], q1 = q[1];
        sa = make_float4(q0.x, q0.z, q1.x, q1.z); sb = make_float4(q0.y, q0.w, q1.y, q1.w);
      }
      if (APPLY && ap.res && ap.rss) {
        const float4* q = (const float4*)(ap.rss + ((int64_t)b * C + c) * 2);
        float4 q0 = q[0], q1 = q[1];
        ra = make_float4(q0.x, q0.z, q1.x, q1.z); rb = make_float4(q0.y, q0.w, q1.y, q1.w);
      }
      for (int p = pbeg + pr; p < pend; p += rows) {
        const int64_t bp = (int64_t)b * P + p;
        unsigned pxu = 0;
        const int py = (APPLY || MODE == 0) ? (int)fd_divmod((unsigned)p, d_w, pxu) : 0, px = (int)pxu;
        float4 v = st_ld4<XS>(x, bp * ld + c);
        if (APPLY) {
          float t[4] = {sa.x * v.x + sb.x, sa.y * v.y + sb.y, sa.z * v.z + sb.z, sa.w * v.w + sb.w};
          float r[4] = {0.f, 0.f, 0.f, 0.f};
          if (ap.res) {
            float4 rv;
            if (ap.res_up2) rv = res_up2_sample(ap.res, b, py, px, H, W, ap.res_ld, c);
            else rv = *(const float4*)(ap.res + bp * ap.res_ld + c);
            r[0] = ra.x * rv.x + rb.x; r[1] = ra.y * rv.y + rb.y; r[2] = ra.z * rv.z + rb.z; r[3] = ra.w * rv.w + rb.w;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float tt = t[i];
            if (res_first) tt += r[i];
            tt = act_apply(tt, act, slope);
            if (!res_first) tt += r[i];
            t[i] = tt * ap.out_scale;
          }
          v = make_float4(t[0], t[1], t[2], t[3]);
          *(float4*)(ap.y + bp * ap.y_ld + c) = v;
        }
        if (MODE == 0) {
          float w = 1.f;
          if (rep_pad) w = (float)((1 + (py == 0) + (py == H - 1)) * (1 + (px == 0) + (px == W - 1)));
          a0.x += w * v.x; a0.y += w * v.y; a0.z += w * v.z; a0.w += w * v.w;
          a1.x += w * v.x * v.x; a1.y += w * v.y * v.y; a1.z += w * v.z * v.z; a1.w += w * v.w * v.w;
        } else {
          if (mb) { float m = mb[p]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
          a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
          a1.x = fmaxf(a1.x, v.x); a1.y = fmaxf(a1.y, v.y); a1.z = fmaxf(a1.z, v.z); a1.w = fmaxf(a1.w, v.w);
        }
      }
    }
    s0[threadIdx.x] = a0;
    s1[threadIdx.x] = a1;
    __syncthreads();
    if (pr == 0 && c < C) {
      for (int r = 1; r < rows; ++r) {
        float4 u = s0[r * lanes + cl], w = s1[r * lanes + cl];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        if (MODE == 1) { a1.x = fmaxf(a1.x, w.x); a1.y = fmaxf(a1.y, w.y); a1.z = fmaxf(a1.z, w.z); a1.w = fmaxf(a1.w, w.w); }
        else { a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w; }
      }
      float4* o = (float4*)(partial + (((int64_t)b * nchunks + chunk) * C + c) * 2);
      o[0] = make_float4(a0.x, a1.x, a0.y, a1.y);
      o[1] = make_float4(a0.z, a1.z, a0.w, a1.w);
    }
    __syncthreads();
  }
}

extern "C" int ppst_in_stats(const void* x, void* partial, int B, int H, int W, int C, int ld, int rep_pad,
                             int* n_partials, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || ld < C || (int64_t)H * W > 0x7fffffffll) return PPST_EINVAL;
  const int chunk = pix_chunk(B, (int64_t)H * W);
  int nchunks = (int)cdiv64((int64_t)H * W, chunk);
  if (n_partials) *n_partials = nchunks;
  if (!x && !partial) return PPST_OK;  // size query
  if (B == 0) return PPST_OK;
  if (!x || !partial) return PPST_ENULL;
  if (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0) {
    ApplyArgs ap = {};
    PPST_LAUNCH((chan_reduce4_kernel<0, false>), dim3(nchunks, B), dim3(256), 0, as_stream(stream), x,
                (const float*)nullptr, (float*)partial, H, W, C, ld, rep_pad, nchunks, ap, chunk, make_fastdiv(W));
  } else {
    PPST_LAUNCH(chan_reduce_kernel<0>, dim3(nchunks, B), dim3(256), 0, as_stream(stream), (const float*)x,
                (const float*)nullptr, (float*)partial, H, W, C, ld, rep_pad, nchunks, chunk);
  }
  return PPST_LAUNCH_CHECK();
}

// One block per (image b, 8-channel group): 32 partial rows x 4 (unrolled) are in flight per step (64-B segments of
// 8 (sum, sumsq) pairs), double accumulation in a fixed order, LDS tree at the end.  (Round 1 used 32 channels x 8 rows:
// B*C/32 blocks -- 32 for a 128-channel map -- each walking up to 256 dependent-latency steps took 16 us per call, 118
// calls per swap step.)
#define FIN_CH 8
#define FIN_ROWS 32
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ partial, int n_partials,
                                                          const float* __restrict__ style, int style_ld,
                                                          const float* __restrict__ post_bias, float* __restrict__ ss, int B,
                                                          int C, double count, float eps, float* __restrict__ mr) {
  __shared__ double sm[FIN_ROWS][FIN_CH][2];
  const int cgroups = (C + FIN_CH - 1) / FIN_CH;
  const int b = blockIdx.x / cgroups, c0 = (blockIdx.x % cgroups) * FIN_CH;
  const int cl = threadIdx.x & (FIN_CH - 1), kk = threadIdx.x / FIN_CH;
  const int c = c0 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float2* p = (const float2*)partial + ((int64_t)b * n_partials * C + c);
    int k = kk;
    for (; k + 3 * FIN_ROWS < n_partials; k += 4 * FIN_ROWS) {
      float2 v0 = p[(int64_t)k * C], v1 = p[(int64_t)(k + FIN_ROWS) * C], v2 = p[(int64_t)(k + 2 * FIN_ROWS) * C],
             v3 = p[(int64_t)(k + 3 * FIN_ROWS) * C];
      s += (double)v0.x; q += (double)v0.y;
      s += (double)v1.x; q += (double)v1.y;
      s += (double)v2.x; q += (double)v2.y;
      s += (double)v3.x; q += (double)v3.y;
    }
    for (; k < n_partials; k += FIN_ROWS) {
      float2 v = p[(int64_t)k * C];
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sm[kk][cl][0] = s;
  sm[kk][cl][1] = q;
  __syncthreads();
  if (kk == 0 && c < C) {
    for (int r = 1; r < FIN_ROWS; ++r) { s += sm[r][cl][0]; q += sm[r][cl][1]; }
    double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    double rstd = 1.0 / sqrt(var + (double)eps);
    double a = rstd, sh = -mean * rstd;
    if (style) {
      double s0 = (double)style[(int64_t)b * style_ld + c] + 1.0;
      double s1 = (double)style[(int64_t)b * style_ld + C + c];
      a = rstd * s0;
      sh = s1 - mean * a;
    }
    if (post_bias) sh += (double)post_bias[c];  // FusedLeakyReLU bias that follows the norm (ConvLayer norm='in')
    ss[((int64_t)b * C + c) * 2] = (float)a;
    ss[((int64_t)b * C + c) * 2 + 1] = (float)sh;
    if (mr) {   // training: the backward of the norm needs (mean, rstd) themselves (ppst_in_bwd_finalize)
      mr[((int64_t)b * C + c) * 2] = (float)mean;
      mr[((int64_t)b * C + c) * 2 + 1] = (float)rstd;
    }
  }
}
extern "C" int ppst_in_finalize(const void* partial, int n_partials, const void* style, int style_ld, const void* post_bias,
                                void* scale_shift, int B, int C, double count, float eps, void* stream) {
  if (B < 0 || C <= 0 || n_partials <= 0 || count <= 0 || (style && style_ld < 2 * C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!partial || !scale_shift) return PPST_ENULL;
  PPST_LAUNCH(in_finalize_kernel, dim3(B * cdiv(C, FIN_CH)), dim3(256), 0, as_stream(stream),
                     (const float*)partial, n_partials, (const float*)style, style_ld, (const float*)post_bias,
                     (float*)scale_shift, B, C, count, eps, (float*)nullptr);
  return PPST_LAUNCH_CHECK();
}
// ppst_in_finalize that also returns mean_rstd [B][C][2] for the backward pass
extern "C" int ppst_in_finalize_train(const void* partial, int n_partials, const void* style, int style_ld, const void* post_bias,
                                      void* scale_shift, void* mean_rstd, int B, int C, double count, float eps, void* stream) {
  if (B < 0 || C <= 0 || n_partials <= 0 || count <= 0 || (style && style_ld < 2 * C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!partial || !scale_shift || !mean_rstd) return PPST_ENULL;
  PPST_LAUNCH(in_finalize_kernel, dim3(B * cdiv(C, FIN_CH)), dim3(256), 0, as_stream(stream),
                     (const float*)partial, n_partials, (const float*)style, style_ld, (const float*)post_bias,
                     (float*)scale_shift, B, C, count, eps, (float*)mean_rstd);
  return PPST_LAUNCH_CHECK();
}

// y = act(a*x + s [+ res (before act)]) ; then [+ res (after act)] ; * out_scale
// act flags: low byte = PPST_ACT_*, bit 8 = residual is added before the activation
__device__ __forceinline__ float act_apply(float t, int act, float slope) {
  if (act == PPST_ACT_LRELU) return (t > 0.f ? t : t * 0.2f) * 1.41421356237309515f;
  if (act == PPST_ACT_PRELU) return t >= 0.f ? t : t * slope;
  return t;
}
// XS: storage type of x and res, YS: of y (common.h; the scalar form is fp32 only).  V = channels per thread: 1, 4 (16-byte items
// of an fp32 tensor) or 8 (16-byte items of a half tensor -- with 4 a half launch moves 8 bytes per lane and instruction issue,
// not HBM, bounds it: 1024^2 x 128 ch fp16 apply pass 0.156 ms with V = 4).
template <int V, int XS = PPST_ST_F32, int YS = PPST_ST_F32>
__global__ __launch_bounds__(256) void affine_act_kernel(const void* __restrict__ xv_, const float* __restrict__ ss,
                                                         const void* __restrict__ res, const float* __restrict__ rss,
                                                         void* __restrict__ yv_, unsigned hw, int C, int x_ld, int res_ld,
                                                         int y_ld, int actf,
                                                         const float* __restrict__ prelu, float out_scale, unsigned total,
                                                         int up2_w, FastDiv d_cv, FastDiv d_hw, FastDiv d_w) {
  const int act = actf & 0xff;
  const bool res_first = (actf >> 8) & 1;
  const float slope = (act == PPST_ACT_PRELU && prelu) ? prelu[0] : 0.f;
  static_assert(V != 1 || (XS == PPST_ST_F32 && YS == PPST_ST_F32), "scalar form: fp32 storage only");
  const float* x = (const float*)xv_;
  float* y = (float*)yv_;
  // 32-bit indices + multiplier division (host guarantees total <= PPST_IDX32_MAX)
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    const unsigned t = (unsigned)t64;
    unsigned cq;
    const unsigned bpu = fd_divmod(t, d_cv, cq);  // b*hw + p
    const int c = (int)cq * V;
    const int64_t bp = bpu;
    const int b = (int)fd_div(bpu, d_hw);
    const float* sp = ss ? ss + ((int64_t)b * C + c) * 2 : nullptr;
    const float* rp = (res && rss) ? rss + ((int64_t)b * C + c) * 2 : nullptr;
    float xv[V], rv[V], o[V];
    if (V >= 4) {
      float4 v[2], r[2];
      if (V == 8) st_ld8<XS>(xv_, bp * x_ld + c, v[0], v[1]);
      else v[0] = st_ld4<XS>(xv_, bp * x_ld + c);
      if (res) {
        if (up2_w > 0) {
          unsigned oxu;
          const int oy = (int)fd_divmod(bpu - (unsigned)b * hw, d_w, oxu), ox = (int)oxu;
#pragma unroll
          for (int q = 0; q < V / 4; ++q)
            r[q] = res_up2_sample<XS>(res, b, oy, ox, (int)(hw / (unsigned)up2_w), up2_w, res_ld, c + 4 * q);
        } else if (V == 8) {
          st_ld8<XS>(res, bp * res_ld + c, r[0], r[1]);
        } else {
          r[0] = st_ld4<XS>(res, bp * res_ld + c);
        }
      }
#pragma unroll
      for (int q = 0; q < V / 4; ++q) {
        xv[4 * q] = v[q].x; xv[4 * q + 1] = v[q].y; xv[4 * q + 2] = v[q].z; xv[4 * q + 3] = v[q].w;
        if (res) { rv[4 * q] = r[q].x; rv[4 * q + 1] = r[q].y; rv[4 * q + 2] = r[q].z; rv[4 * q + 3] = r[q].w; }
      }
    } else {
      xv[0] = x[bp * x_ld + c];
      if (res) rv[0] = ((const float*)res)[bp * res_ld + c];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float tt = sp ? sp[i * 2] * xv[i] + sp[i * 2 + 1] : xv[i];
      if (rp) rv[i] = rp[i * 2] * rv[i] + rp[i * 2 + 1];
      if (res && res_first) tt += rv[i];
      tt = act_apply(tt, act, slope);
      if (res && !res_first) tt += rv[i];
      o[i] = tt * out_scale;
    }
    if (V == 8) st_st8<YS>(yv_, bp * y_ld + c, make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7]));
    else if (V == 4) st_st4<YS>(yv_, bp * y_ld + c, make_float4(o[0], o[1], o[2], o[3]));
    else y[bp * y_ld + c] = o[0];
  }
}
extern "C" int ppst_affine_act_st(const void* x, const void* scale_shift, const void* res, const void* res_scale_shift, void* y,
                                  int B, int64_t hw, int C, int x_ld, int res_ld, int y_ld, int act, const void* prelu,
                                  float out_scale, int res_up2_w, int x_st, int y_st, void* stream) {
  if ((unsigned)x_st > 2u || (unsigned)y_st > 2u || (x_st && y_st && x_st != y_st)) return PPST_EINVAL;
  if (res_up2_w < 0 || (res_up2_w > 0 && (!res || hw % res_up2_w || res_up2_w % 2 || (hw / res_up2_w) % 2 || C % 4))) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || C <= 0 || x_ld < C || y_ld < C || (res && res_ld < C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  if (res_up2_w > 0 && (x_ld % 4 || y_ld % 4 || res_ld % 4)) return PPST_EINVAL;
  bool vec = C % 4 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0 && (!res || res_ld % 4 == 0) &&
             (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) % (x_st || y_st ? 8 : 16) == 0);
  if ((x_st || y_st) && !vec) return PPST_EINVAL;     // half storage: the vector forms only
  // 8 channels per thread when a half tensor is involved and everything is 16-byte addressable
  const bool w8 = vec && (x_st || y_st) && C % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && (!res || res_ld % 8 == 0) &&
                  (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) % 16 == 0);
  const int V = w8 ? 8 : (vec ? 4 : 1);
  int64_t total = (int64_t)B * hw * (C / V);
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  const FastDiv d_cv = make_fastdiv((unsigned)(C / V)), d_hw = make_fastdiv((unsigned)hw),
                d_w = make_fastdiv(res_up2_w > 0 ? (unsigned)res_up2_w : 1u);
#define AA_GO(XS_, YS_)                                                                                                   \
  do {                                                                                                                    \
    if (w8) PPST_LAUNCH((affine_act_kernel<8, XS_, YS_>), dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,      \
                        (const float*)scale_shift, res, (const float*)res_scale_shift, y, (unsigned)hw, C, x_ld, res_ld, y_ld, act, \
                        (const float*)prelu, out_scale, (unsigned)total, res_up2_w, d_cv, d_hw, d_w);                      \
    else PPST_LAUNCH((affine_act_kernel<4, XS_, YS_>), dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,         \
                     (const float*)scale_shift, res, (const float*)res_scale_shift, y, (unsigned)hw, C, x_ld, res_ld, y_ld, act, \
                     (const float*)prelu, out_scale, (unsigned)total, res_up2_w, d_cv, d_hw, d_w);                         \
  } while (0)
  if (vec) {
    if (x_st == PPST_ST_F16 && y_st == PPST_ST_F16) AA_GO(PPST_ST_F16, PPST_ST_F16);
    else if (x_st == PPST_ST_F16) AA_GO(PPST_ST_F16, PPST_ST_F32);
    else if (x_st == PPST_ST_BF16 && y_st == PPST_ST_BF16) AA_GO(PPST_ST_BF16, PPST_ST_BF16);
    else if (x_st == PPST_ST_BF16) AA_GO(PPST_ST_BF16, PPST_ST_F32);
    else if (y_st == PPST_ST_F16) AA_GO(PPST_ST_F32, PPST_ST_F16);
    else if (y_st == PPST_ST_BF16) AA_GO(PPST_ST_F32, PPST_ST_BF16);
    else PPST_LAUNCH((affine_act_kernel<4>), dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,
                     (const float*)scale_shift, res, (const float*)res_scale_shift, y, (unsigned)hw, C, x_ld, res_ld, y_ld, act,
                     (const float*)prelu, out_scale, (unsigned)total, res_up2_w, d_cv, d_hw, d_w);
  } else
    PPST_LAUNCH(affine_act_kernel<1>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x,
                       (const float*)scale_shift, res, (const float*)res_scale_shift, y, (unsigned)hw, C, x_ld,
                       res_ld, y_ld, act, (const float*)prelu, out_scale, (unsigned)total, 0, d_cv, d_hw, d_w);
#undef AA_GO
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_affine_act(const void* x, const void* scale_shift, const void* res, const void* res_scale_shift, void* y,
                               int B, int64_t hw, int C, int x_ld, int res_ld, int y_ld, int act, const void* prelu,
                               float out_scale, int res_up2_w, void* stream) {
  return ppst_affine_act_st(x, scale_shift, res, res_scale_shift, y, B, hw, C, x_ld, res_ld, y_ld, act, prelu, out_scale, res_up2_w,
                            PPST_ST_F32, PPST_ST_F32, stream);
}

// ppst_affine_act that also emits the instance-norm partials of its OUTPUT
// (partial [B][n_partials][C][2], n_partials as ppst_in_stats reports for (H, W)).
extern "C" int ppst_affine_act_stats(const void* x, const void* scale_shift, const void* res, const void* res_scale_shift,
                                     void* y, void* partial, int B, int H, int W, int C, int x_ld, int res_ld, int y_ld,
                                     int act, const void* prelu, float out_scale, int rep_pad, int res_up2, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || x_ld % 4 || y_ld % 4 || x_ld < C || y_ld < C ||
      (res && (res_ld < C || res_ld % 4)))
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y || !partial) return PPST_ENULL;
  if ((int64_t)H * W > 0x7fffffffll) return PPST_EINVAL;
  const int chunk = pix_chunk(B, (int64_t)H * W);
  int nchunks = (int)cdiv64((int64_t)H * W, chunk);
  ApplyArgs ap;
  ap.ss = (const float*)scale_shift; ap.res = (const float*)res; ap.rss = (const float*)res_scale_shift; ap.y = (float*)y;
  ap.prelu = (const float*)prelu; ap.res_ld = res_ld; ap.y_ld = y_ld; ap.actf = act; ap.out_scale = out_scale;
  ap.res_up2 = (res && res_up2) ? 1 : 0;
  if (ap.res_up2 && (H % 2 || W % 2)) return PPST_EINVAL;
  PPST_LAUNCH((chan_reduce4_kernel<0, true>), dim3(nchunks, B), dim3(256), 0, as_stream(stream), x,
              (const float*)nullptr, (float*)partial, H, W, C, x_ld, rep_pad, nchunks, ap, chunk, make_fastdiv(W));
  return PPST_LAUNCH_CHECK();
}

// ---------------------------------------------------------------- GAP/GMP --
// One block per (image b, 8-channel group), 32 partial rows x 4 in flight (as in_finalize_kernel), double accumulation
// of the sums in a fixed order, LDS tree at the end.
__global__ __launch_bounds__(256) void gap_gmp_finalize_kernel(const float* __restrict__ partial, int n_partials,
                                                               float* __restrict__ out, int B, int C, double count) {
  __shared__ double ss[FIN_ROWS][FIN_CH];
  __shared__ float sm[FIN_ROWS][FIN_CH];
  const int cgroups = (C + FIN_CH - 1) / FIN_CH;
  const int b = blockIdx.x / cgroups, c0 = (blockIdx.x % cgroups) * FIN_CH;
  const int cl = threadIdx.x & (FIN_CH - 1), kk = threadIdx.x / FIN_CH;
  const int c = c0 + cl;
  double s = 0.0;
  float m = -INFINITY;
  if (c < C) {
    const float2* p = (const float2*)partial + ((int64_t)b * n_partials * C + c);
    int k = kk;
    for (; k + 3 * FIN_ROWS < n_partials; k += 4 * FIN_ROWS) {
      float2 v0 = p[(int64_t)k * C], v1 = p[(int64_t)(k + FIN_ROWS) * C], v2 = p[(int64_t)(k + 2 * FIN_ROWS) * C],
             v3 = p[(int64_t)(k + 3 * FIN_ROWS) * C];
      s += (double)v0.x; s += (double)v1.x; s += (double)v2.x; s += (double)v3.x;
      m = fmaxf(fmaxf(m, v0.y), fmaxf(v1.y, fmaxf(v2.y, v3.y)));
    }
    for (; k < n_partials; k += FIN_ROWS) {
      float2 v = p[(int64_t)k * C];
      s += (double)v.x;
      m = fmaxf(m, v.y);
    }
  }
  ss[kk][cl] = s;
  sm[kk][cl] = m;
  __syncthreads();
  if (kk == 0 && c < C) {
    for (int r = 1; r < FIN_ROWS; ++r) { s += ss[r][cl]; m = fmaxf(m, sm[r][cl]); }
    out[(int64_t)b * 2 * C + c] = (float)(s / count);
    out[(int64_t)b * 2 * C + C + c] = m;
  }
}
extern "C" int64_t ppst_gap_gmp_ws(int B, int64_t hw, int C) {
  if (B <= 0 || hw <= 0) return 0;
  return cdiv64(hw, pix_chunk(B, hw)) * C * 2 * B * (int64_t)sizeof(float);
}
extern "C" int ppst_gap_gmp_st(const void* x, const void* mask, void* out, void* ws, int B, int H, int W, int C, int ld,
                                int x_st, void* stream) {
  if ((unsigned)x_st > 2u || (x_st && (C % 4 || ld % 4 || (uintptr_t)x % 8))) return PPST_EINVAL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || ld < C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !out || !ws) return PPST_ENULL;
  if ((int64_t)H * W > 0x7fffffffll) return PPST_EINVAL;
  const int chunk = pix_chunk(B, (int64_t)H * W);
  int nchunks = (int)cdiv64((int64_t)H * W, chunk);
  if (x_st) {
    ApplyArgs ap = {};
    if (x_st == PPST_ST_F16)
      PPST_LAUNCH((chan_reduce4_kernel<1, false, PPST_ST_F16>), dim3(nchunks, B), dim3(256), 0, as_stream(stream), x,
                  (const float*)mask, (float*)ws, H, W, C, ld, 0, nchunks, ap, chunk, make_fastdiv(W));
    else
      PPST_LAUNCH((chan_reduce4_kernel<1, false, PPST_ST_BF16>), dim3(nchunks, B), dim3(256), 0, as_stream(stream), x,
                  (const float*)mask, (float*)ws, H, W, C, ld, 0, nchunks, ap, chunk, make_fastdiv(W));
  } else if (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0) {
    ApplyArgs ap = {};
    PPST_LAUNCH((chan_reduce4_kernel<1, false>), dim3(nchunks, B), dim3(256), 0, as_stream(stream), x,
                (const float*)mask, (float*)ws, H, W, C, ld, 0, nchunks, ap, chunk, make_fastdiv(W));
  } else {
    PPST_LAUNCH(chan_reduce_kernel<1>, dim3(nchunks, B), dim3(256), 0, as_stream(stream), (const float*)x,
                (const float*)mask, (float*)ws, H, W, C, ld, 0, nchunks, chunk);
  }
  int e = PPST_LAUNCH_CHECK();
  if (e) return e;
  PPST_LAUNCH(gap_gmp_finalize_kernel, dim3(B * cdiv(C, FIN_CH)), dim3(256), 0, as_stream(stream),
                     (const float*)ws, nchunks, (float*)out, B, C, (double)H * W);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_gap_gmp(const void* x, const void* mask, void* out, void* ws, int B, int H, int W, int C, int ld,
                             void* stream) {
  return ppst_gap_gmp_st(x, mask, out, ws, B, H, W, C, ld, PPST_ST_F32, stream);
}


// ---- GAP || GMP of x * mask for SEVERAL masks in one read of x (round 5; the masked E2 heads of encoder_col.py:217-245 pool every
// feature map four times -- unmasked and under each of the three class masks).  heads: h = 0 the plain pooling (with_plain), then one
// per mask channel; masks [B][hw][nm] (the NHWC planes of the one-hot pyramid), nm <= 3.  Same block geometry, lane map and summation
// order per head as chan_reduce4_kernel<1>: a head's values equal the single-head launch's bit for bit.
// partial [(h * B + b)][nchunks][C][2] -> gap_gmp_finalize_kernel over heads * B rows.
#define GGM_MAXH 4
template <int XS = PPST_ST_F32>
__global__ __launch_bounds__(256) void gap_gmp_multi_kernel(const void* __restrict__ x, const float* __restrict__ masks,
                                                            float* __restrict__ partial, int P, int C, int ld, int nm, int with_plain,
                                                            int B, int nchunks, int PIX_CHUNK) {
  __shared__ float4 s0[256], s1[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pbeg = chunk * PIX_CHUNK;
  const int pend = (pbeg + PIX_CHUNK < P) ? pbeg + PIX_CHUNK : P;
  const int c4n = C >> 2;
  int lanes = 1;
  while (lanes < c4n && lanes < 256) lanes <<= 1;
  const int rows = 256 / lanes;
  const int cl = threadIdx.x % lanes, pr = threadIdx.x / lanes;
  const int nh = nm + (with_plain ? 1 : 0);
  const float* mb = masks + (int64_t)b * P * nm;
  for (int cbase = 0; cbase < c4n; cbase += lanes) {
    const int c = (cbase + cl) * 4;
    float4 a0[GGM_MAXH], a1[GGM_MAXH];
#pragma unroll
    for (int h = 0; h < GGM_MAXH; ++h) { a0[h] = make_float4(0.f, 0.f, 0.f, 0.f); a1[h] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); }
    if (c < C) {
      for (int p = pbeg + pr; p < pend; p += rows) {
        const float4 v = st_ld4<XS>(x, ((int64_t)b * P + p) * ld + c);
        float m[GGM_MAXH];
#pragma unroll
        for (int h = 0; h < GGM_MAXH; ++h) {
          const int mi = h - (with_plain ? 1 : 0);
          m[h] = (mi < 0) ? 1.f : (mi < nm ? mb[(int64_t)p * nm + mi] : 0.f);
        }
#pragma unroll
        for (int h = 0; h < GGM_MAXH; ++h) {
          if (h < nh) {
            const float4 w = make_float4(v.x * m[h], v.y * m[h], v.z * m[h], v.w * m[h]);
            a0[h].x += w.x; a0[h].y += w.y; a0[h].z += w.z; a0[h].w += w.w;
            a1[h].x = fmaxf(a1[h].x, w.x); a1[h].y = fmaxf(a1[h].y, w.y); a1[h].z = fmaxf(a1[h].z, w.z); a1[h].w = fmaxf(a1[h].w, w.w);
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < GGM_MAXH; ++h) {
      if (h < nh) {
        s0[threadIdx.x] = a0[h];
        s1[threadIdx.x] = a1[h];
        __syncthreads();
        if (pr == 0 && c < C) {
          float4 t0 = a0[h], t1 = a1[h];
          for (int r = 1; r < rows; ++r) {
            const float4 u = s0[r * lanes + cl], w = s1[r * lanes + cl];
            t0.x += u.x; t0.y += u.y; t0.z += u.z; t0.w += u.w;
            t1.x = fmaxf(t1.x, w.x); t1.y = fmaxf(t1.y, w.y); t1.z = fmaxf(t1.z, w.z); t1.w = fmaxf(t1.w, w.w);
          }
          float4* o = (float4*)(partial + ((((int64_t)h * B + b) * nchunks + chunk) * C + c) * 2);
          o[0] = make_float4(t0.x, t1.x, t0.y, t1.y);
          o[1] = make_float4(t0.z, t1.z, t0.w, t1.w);
        }
        __syncthreads();
      }
    }
  }
}
extern "C" int64_t ppst_gap_gmp_multi_ws(int B, int64_t hw, int C, int heads) {
  if (B <= 0 || hw <= 0 || heads <= 0) return 0;
  return cdiv64(hw, pix_chunk(B, hw)) * C * 2 * B * heads * (int64_t)sizeof(float);
}
// x [B][hw][ld] fp32, masks [B][hw][nm] (nm in 1..3), with_plain 0 / 1 -> out [(nm + with_plain) * B][2C], head-major
extern "C" int ppst_gap_gmp_multi(const void* x, const void* masks, void* out, void* ws, int B, int H, int W, int C, int ld, int nm,
                                  int with_plain, int x_st, void* stream) {
  if ((unsigned)x_st > 2u) return PPST_EINVAL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || ld % 4 || ld < C || nm < 1 || nm > 3 || (with_plain != 0 && with_plain != 1))
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !masks || !out || !ws) return PPST_ENULL;
  if ((int64_t)H * W > 0x7fffffffll || ((uintptr_t)x % (x_st ? 8 : 16))) return PPST_EINVAL;
  const int chunk = pix_chunk(B, (int64_t)H * W);
  const int nchunks = (int)cdiv64((int64_t)H * W, chunk), heads = nm + with_plain;
  PPST_ST_SWITCH(x_st, PPST_LAUNCH(gap_gmp_multi_kernel<ST_>, dim3(nchunks, B), dim3(256), 0, as_stream(stream), x, (const float*)masks,
                                   (float*)ws, H * W, C, ld, nm, with_plain, B, nchunks, chunk));
  int e = PPST_LAUNCH_CHECK();
  if (e) return e;
  PPST_LAUNCH(gap_gmp_finalize_kernel, dim3(heads * B * cdiv(C, FIN_CH)), dim3(256), 0, as_stream(stream), (const float*)ws, nchunks,
              (float*)out, heads * B, C, (double)H * W);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------ pooling / resize ---
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                      int C, int x_ld, int f, int y_ld, unsigned total, FastDiv d_c,
                                                      FastDiv d_ow, FastDiv d_oh) {
  const int OH = H / f, OW = W / f;
  const float inv = 1.f / (float)(f * f);
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq, oxu, oyu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cq);
    r = fd_divmod(r, d_ow, oxu);
    const int b = (int)fd_divmod(r, d_oh, oyu);
    const int c = (int)cq * 4, ox = (int)oxu, oy = (int)oyu;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        float4 v = *(const float4*)(x + (((int64_t)b * H + oy * f + dy) * W + ox * f + dx) * x_ld + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    *(float4*)(y + (((int64_t)b * OH + oy) * OW + ox) * y_ld + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}
extern "C" int ppst_avgpool(const void* x, void* y, int B, int H, int W, int C, int x_ld, int f, int y_ld, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || x_ld % 4 || y_ld % 4 || f <= 0 || H % f || W % f || x_ld < C || y_ld < C)
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * (H / f) * (W / f) * (C / 4);
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y,
                     H, W, C, x_ld, f, y_ld, (unsigned)total, make_fastdiv(C / 4), make_fastdiv(W / f), make_fastdiv(H / f));
  return PPST_LAUNCH_CHECK();
}

// F.interpolate(mode='bilinear', align_corners=False): src = (dst+0.5)*in/out - 0.5, clamped at 0
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                       int C, int x_ld, int OH, int OW, int y_ld, unsigned total,
                                                       FastDiv d_c, FastDiv d_ow, FastDiv d_oh) {
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq, oxu, oyu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cq);
    r = fd_divmod(r, d_ow, oxu);
    const int b = (int)fd_divmod(r, d_oh, oyu);
    const int c = (int)cq * 4, ox = (int)oxu, oy = (int)oyu;
    float fy = fmaxf(((float)oy + 0.5f) * sh - 0.5f, 0.f);
    float fx = fmaxf(((float)ox + 0.5f) * sw - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    float ly = fy - (float)y0, lx = fx - (float)x0;
    float hy = 1.f - ly, hx = 1.f - lx;
    const float* base = x + (int64_t)b * H * W * x_ld + c;
    float4 v00 = *(const float4*)(base + ((int64_t)y0 * W + x0) * x_ld);
    float4 v01 = *(const float4*)(base + ((int64_t)y0 * W + x1) * x_ld);
    float4 v10 = *(const float4*)(base + ((int64_t)y1 * W + x0) * x_ld);
    float4 v11 = *(const float4*)(base + ((int64_t)y1 * W + x1) * x_ld);
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *(float4*)(y + (((int64_t)b * OH + oy) * OW + ox) * y_ld + c) = o;
  }
}
extern "C" int ppst_bilinear(const void* x, void* y, int B, int H, int W, int C, int x_ld, int OH, int OW, int y_ld,
                             void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || x_ld % 4 || y_ld % 4 || OH <= 0 || OW <= 0 || x_ld < C || y_ld < C)
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * OH * OW * (C / 4);
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y,
                     H, W, C, x_ld, OH, OW, y_ld, (unsigned)total, make_fastdiv(C / 4), make_fastdiv(OW), make_fastdiv(OH));
  return PPST_LAUNCH_CHECK();
}

// Tail of a correspondence feature head (generator.py:174-238): f = act(a*x + s) of the head's last conv is never
// needed at full resolution -- only its PxP average (into ``feat``) and its bilinear resize to the feat1 grid, which
// for D = 1 is f itself and for D = 2 (align_corners=False, exact factor 2) the 2x2 mean.  One read of x instead of
// an apply pass + two reads.  thread = (b, oy, ox) of the pooled grid x 4 channels; P in {2,4,8}, D in {1,2}, D | P.
__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ x, const float* __restrict__ ss,
                                                        const float* __restrict__ prelu, float* __restrict__ feat,
                                                        float* __restrict__ feat1, int H, int W, int C, int x_ld, int f_ld,
                                                        int f1_ld, int P, int D, int actf, unsigned total, FastDiv d_c,
                                                        FastDiv d_w, FastDiv d_h) {
  const int act = actf & 0xff;
  const float slope = (act == PPST_ACT_PRELU && prelu) ? prelu[0] : 0.f;
  const int OW = W / P, OH = H / P, W1 = W / D, H1 = H / D;
  const float inv = 1.f / (float)(P * P);
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq, oxu, oyu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cq);
    r = fd_divmod(r, d_w, oxu);
    const int b = (int)fd_divmod(r, d_h, oyu);
    const int c = (int)cq * 4, ox = (int)oxu, oy = (int)oyu;
    const float4* q = (const float4*)(ss + ((int64_t)b * C + c) * 2);
    const float4 q0 = q[0], q1 = q[1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < P; dy += D) {
      for (int dx = 0; dx < P; dx += D) {
        float4 v[2][2];
#pragma unroll
        for (int ey = 0; ey < 2; ++ey)
#pragma unroll
          for (int ex = 0; ex < 2; ++ex) {
            if (ey < D && ex < D) {
              float4 t = *(const float4*)(x + (((int64_t)b * H + oy * P + dy + ey) * W + ox * P + dx + ex) * x_ld + c);
              t.x = act_apply(q0.x * t.x + q0.y, act, slope); t.y = act_apply(q0.z * t.y + q0.w, act, slope);
              t.z = act_apply(q1.x * t.z + q1.y, act, slope); t.w = act_apply(q1.z * t.w + q1.w, act, slope);
              v[ey][ex] = t;
            }
          }
        float4 o;
        if (D == 1) {
          o = v[0][0];
        } else {  // the bilinear kernel's expression with hx = lx = hy = ly = 0.5
          o.x = 0.5f * (0.5f * v[0][0].x + 0.5f * v[0][1].x) + 0.5f * (0.5f * v[1][0].x + 0.5f * v[1][1].x);
          o.y = 0.5f * (0.5f * v[0][0].y + 0.5f * v[0][1].y) + 0.5f * (0.5f * v[1][0].y + 0.5f * v[1][1].y);
          o.z = 0.5f * (0.5f * v[0][0].z + 0.5f * v[0][1].z) + 0.5f * (0.5f * v[1][0].z + 0.5f * v[1][1].z);
          o.w = 0.5f * (0.5f * v[0][0].w + 0.5f * v[0][1].w) + 0.5f * (0.5f * v[1][0].w + 0.5f * v[1][1].w);
        }
        *(float4*)(feat1 + (((int64_t)b * H1 + (oy * P + dy) / D) * W1 + (ox * P + dx) / D) * f1_ld + c) = o;
        // average pool: plain sum of f over the block in row-major order would need f at the (dy+ey, dx+ex) order of
        // ppst_avgpool; D = 2 visits 2x2 sub-blocks instead -- the sums differ in the last bit only
#pragma unroll
        for (int ey = 0; ey < 2; ++ey)
#pragma unroll
          for (int ex = 0; ex < 2; ++ex)
            if (ey < D && ex < D) { acc.x += v[ey][ex].x; acc.y += v[ey][ex].y; acc.z += v[ey][ex].z; acc.w += v[ey][ex].w; }
      }
    }
    *(float4*)(feat + (((int64_t)b * OH + oy) * OW + ox) * f_ld + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}
extern "C" int ppst_head_tail(const void* x, const void* scale_shift, const void* prelu, void* feat, void* feat1, int B, int H, int W,
                              int C, int x_ld, int feat_ld, int feat1_ld, int P, int D, int act, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || x_ld % 4 || feat_ld % 4 || feat1_ld % 4 || x_ld < C || feat_ld < C || feat1_ld < C ||
      (D != 1 && D != 2) || P <= 0 || P % D || H % P || W % P)
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !scale_shift || !feat || !feat1) return PPST_ENULL;
  int64_t total = (int64_t)B * (H / P) * (W / P) * (C / 4);
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(head_tail_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)scale_shift,
              (const float*)prelu, (float*)feat, (float*)feat1, H, W, C, x_ld, feat_ld, feat1_ld, P, D, act, (unsigned)total,
              make_fastdiv(C / 4), make_fastdiv(W / P), make_fastdiv(H / P));
  return PPST_LAUNCH_CHECK();
}

// nearest x2 upsample (Upscale2d, stylegan2_layers.py:86-97; the <128 px branch of
// EqualizedConv2d :322-323)
__global__ __launch_bounds__(256) void upsample_nearest2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int H, int W,
                                                                int c4n, unsigned total, FastDiv d_c, FastDiv d_ow,
                                                                FastDiv d_oh) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    const unsigned t = (unsigned)t64;
    unsigned cq, oxu, oyu;
    unsigned r = fd_divmod(t, d_c, cq);
    r = fd_divmod(r, d_ow, oxu);
    const int64_t b = fd_divmod(r, d_oh, oyu);
    const int c = (int)cq, ox = (int)oxu, oy = (int)oyu;
    y[t] = x[((b * H + (oy >> 1)) * W + (ox >> 1)) * c4n + c];
  }
}
extern "C" int ppst_upsample_nearest2_st(const void* x, void* y, int B, int H, int W, int C, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (st) {                      // a copy: 8 half channels are one 16-byte item
    if (C % 8) return PPST_EINVAL;
    C /= 2;
  }
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * 4 * H * W * (C / 4);
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(upsample_nearest2_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float4*)x,
                     (float4*)y, H, W, C / 4, (unsigned)total, make_fastdiv(C / 4), make_fastdiv(2 * W), make_fastdiv(2 * H));
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_upsample_nearest2(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  return ppst_upsample_nearest2_st(x, y, B, H, W, C, PPST_ST_F32, stream);
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                       unsigned total, FastDiv d_c, FastDiv d_ow, FastDiv d_oh) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    const unsigned t = (unsigned)t64;
    unsigned cq, oxu, oyu;
    unsigned r = fd_divmod(t, d_c, cq);
    r = fd_divmod(r, d_ow, oxu);
    const int b = (int)fd_divmod(r, d_oh, oyu);
    const int c = (int)cq, ox = (int)oxu, oy = (int)oyu;
    const float* p = x + (((int64_t)b * H + oy * 2) * W + ox * 2) * C + c;
    y[t] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(int64_t)W * C], p[(int64_t)W * C + C]));
  }
}
extern "C" int ppst_maxpool2(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  if (B < 0 || H <= 1 || W <= 1 || C <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(maxpool2_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, H, W, C,
              (unsigned)total, make_fastdiv(C), make_fastdiv(W / 2), make_fastdiv(H / 2));
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------------- small 1x1 convs --
// Cin <= 4 (FromRGB): thread = (pixel, 4 output channels).  The grid stride is a multiple of
// cout/4, so a thread keeps its output-channel group: its 4 x cin weights and biases are loaded
// once, the loop is one pixel read + one 16-B store.
template <int YS = PPST_ST_F32>
__global__ __launch_bounds__(256) void conv1x1_small_cin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, void* __restrict__ y,
                                                                int64_t npix, int cin, int in_ld, int cout, float wscale,
                                                                int act, FastDiv d_c) {
  const unsigned t0 = blockIdx.x * 256 + threadIdx.x;
  unsigned cq;
  int64_t p = fd_divmod(t0, d_c, cq);
  const int64_t pstep = ((int64_t)gridDim.x * 256) / d_c.d;  // exact: the host rounds the grid
  const int co = (int)cq * 4;
  float wr[4][4], br[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    br[j] = bias ? bias[co + j] : 0.f;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) wr[j][ci] = ci < cin ? w[(co + j) * cin + ci] * wscale : 0.f;
  }
  for (; p < npix; p += pstep) {
    float xv[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) xv[ci] = ci < cin ? x[p * in_ld + ci] : 0.f;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) acc += xv[ci] * wr[j][ci];  // ci >= cin adds +0 (same sum as the cin-term loop)
      o[j] = act_apply(acc + br[j], act, 0.f);
    }
    st_st4<YS>(y, p * cout + co, make_float4(o[0], o[1], o[2], o[3]));
  }
}
extern "C" int ppst_conv1x1_small_cin_st(const void* x, const void* w, const void* bias, void* y, int64_t npix, int cin,
                                         int in_ld, int cout, float wscale, int act, int y_st, void* stream) {
  if ((unsigned)y_st > 2u) return PPST_EINVAL;
  if (npix < 0 || cin <= 0 || cin > 4 || in_ld < cin || cout <= 0 || cout % 4) return PPST_EINVAL;
  if (npix == 0) return PPST_OK;
  if (!x || !w || !y) return PPST_ENULL;
  const int c4n = cout / 4;
  int64_t total = npix * c4n;
  // grid stride (blocks * 256) must be a multiple of c4n: blocks a multiple of c4n / gcd(c4n, 256)
  int g = c4n, t256 = 256;
  while (t256) { int r = g % t256; g = t256; t256 = r; }
  const int unit = c4n / g;
  int64_t blocks = (int64_t)grid_for(total);
  blocks = ((blocks + unit - 1) / unit) * unit;
  PPST_ST_SWITCH(y_st, PPST_LAUNCH(conv1x1_small_cin_kernel<ST_>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)x,
                                   (const float*)w, (const float*)bias, y, npix, cin, in_ld, cout, wscale, act, make_fastdiv(c4n)));
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_conv1x1_small_cin(const void* x, const void* w, const void* bias, void* y, int64_t npix, int cin,
                                      int in_ld, int cout, float wscale, int act, void* stream) {
  return ppst_conv1x1_small_cin_st(x, w, bias, y, npix, cin, in_ld, cout, wscale, act, PPST_ST_F32, stream);
}

// Cout <= 4 (ToRGB): 32 lanes cooperate on one pixel (float4 of channels each, strided over
// Cin), xor-shuffle reduction inside the half wave.
template <int XS = PPST_ST_F32>
__global__ __launch_bounds__(256) void conv1x1_small_cout_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int64_t npix, int cin, int cout, float wscale) {
  const int hl = threadIdx.x & 31;
  const int64_t half_id = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t nhalf = ((int64_t)gridDim.x * 256) >> 5;
  for (int64_t p = half_id; p < npix; p += nhalf) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = hl * 4; c < cin; c += 128) {
      float4 v = st_ld4<XS>(x, p * cin + c);
      for (int j = 0; j < cout; ++j) {
        float4 ww = *(const float4*)(w + (int64_t)j * cin + c);
        acc[j] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    if (hl < cout) {
      float r = hl == 0 ? acc[0] : hl == 1 ? acc[1] : hl == 2 ? acc[2] : acc[3];
      y[p * cout + hl] = r * wscale + (bias ? bias[hl] : 0.f);
    }
  }
}
// Cout == 3 (ToRGB): 32 lanes x 8 pixels per iteration.  Each lane keeps 8 x 3 partial dot products over its
// channel slice; the cross-lane sum is a transposing butterfly that halves the value count at offsets 16, 8, 4
// (24 -> 12 -> 6 -> 3 values) and finishes with a 2-stage butterfly: 27 shuffles per 8 pixels instead of 15 per pixel.
template <int XS = PPST_ST_F32>
__global__ __launch_bounds__(256) void conv1x1_cout3_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int64_t npix, int cin, float wscale) {
  const int hl = threadIdx.x & 31;
  const int64_t half_id = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t nhalf = ((int64_t)gridDim.x * 256) >> 5;
  const int b4 = (hl >> 4) & 1, b3 = (hl >> 3) & 1, b2 = (hl >> 2) & 1;
  for (int64_t p0 = half_id * 8; p0 < npix; p0 += nhalf * 8) {
    float v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = 0.f;
    for (int c = hl * 4; c < cin; c += 128) {
      const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + (int64_t)cin + c), w2 = *(const float4*)(w + 2 * (int64_t)cin + c);
      float4 xv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t p = p0 + k < npix ? p0 + k : npix - 1;
        xv[k] = st_ld4<XS>(x, p * cin + c);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k * 3 + 0] += xv[k].x * w0.x + xv[k].y * w0.y + xv[k].z * w0.z + xv[k].w * w0.w;
        v[k * 3 + 1] += xv[k].x * w1.x + xv[k].y * w1.y + xv[k].z * w1.z + xv[k].w * w1.w;
        v[k * 3 + 2] += xv[k].x * w2.x + xv[k].y * w2.y + xv[k].z * w2.z + xv[k].w * w2.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {   // offset 16: pixels 0..3 stay on lanes with bit4 = 0, pixels 4..7 on bit4 = 1
      const float keep = b4 ? v[12 + i] : v[i], send = b4 ? v[i] : v[12 + i];
      v[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float keep = b3 ? v[6 + i] : v[i], send = b3 ? v[i] : v[6 + i];
      v[i] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float keep = b2 ? v[3 + i] : v[i], send = b2 ? v[i] : v[3 + i];
      v[i] = keep + __shfl_xor(send, 4, 64);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] += __shfl_xor(v[i], 2, 64);
      v[i] += __shfl_xor(v[i], 1, 64);
    }
    const int k = b4 * 4 + b3 * 2 + b2, j = hl & 3;
    if (j < 3 && p0 + k < npix) {
      const float r = j == 0 ? v[0] : (j == 1 ? v[1] : v[2]);
      y[(p0 + k) * 3 + j] = r * wscale + (bias ? bias[j] : 0.f);
    }
  }
}

extern "C" int ppst_conv1x1_small_cout_st(const void* x, const void* w, const void* bias, void* y, int64_t npix, int cin,
                                          int cout, float wscale, int x_st, void* stream) {
  if ((unsigned)x_st > 2u || (x_st && (uintptr_t)x % 8)) return PPST_EINVAL;
  if (npix < 0 || cin <= 0 || cin % 4 || cout <= 0 || cout > 4) return PPST_EINVAL;
  if (npix == 0) return PPST_OK;
  if (!x || !w || !y) return PPST_ENULL;
  if (cout == 3 && ((uintptr_t)x % (x_st ? 8 : 16)) == 0 && ((uintptr_t)w % 16) == 0) {
    PPST_ST_SWITCH(x_st, PPST_LAUNCH(conv1x1_cout3_kernel<ST_>, dim3(grid_for(cdiv64(npix, 8) * 32)), dim3(256), 0, as_stream(stream), x,
                                     (const float*)w, (const float*)bias, (float*)y, npix, cin, wscale));
    return PPST_LAUNCH_CHECK();
  }
  PPST_ST_SWITCH(x_st, PPST_LAUNCH(conv1x1_small_cout_kernel<ST_>, dim3(grid_for(npix * 32)), dim3(256), 0, as_stream(stream),
                                   x, (const float*)w, (const float*)bias, (float*)y, npix, cin, cout, wscale));
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_conv1x1_small_cout(const void* x, const void* w, const void* bias, void* y, int64_t npix, int cin,
                                       int cout, float wscale, void* stream) {
  return ppst_conv1x1_small_cout_st(x, w, bias, y, npix, cin, cout, wscale, PPST_ST_F32, stream);
}


// ToRGB's 1x1 conv (Cout = 3) with the producer's merge pass applied ON LOAD (round 5): the input is read as
//   x_eff = (a[b][c] * x + s[b][c] + bilinear_up2(res)) * out_scale
// -- the (IN + StyleMod of conv2 + x2-upsampled skip) / sqrt2 of the last UpsamplingResnetBlock (generator.py:63-78), whose only
// consumer in the image pass is this conv: the 128-channel 512^2 tensor is neither written nor read back.  Same lane / butterfly
// scheme as conv1x1_cout3_kernel; the apply arithmetic is affine_act_kernel's, in its order.
template <int XS = PPST_ST_F32>
__global__ __launch_bounds__(256) void conv1x1_cout3_apply_kernel(const void* __restrict__ x, const float* __restrict__ ss,
                                                                  const void* __restrict__ res, int res_ld, float out_scale,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ y, int64_t npix, int H, int W, int cin, float wscale,
                                                                  FastDiv d_hw, FastDiv d_w) {
  const int hl = threadIdx.x & 31;
  const int64_t half_id = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t nhalf = ((int64_t)gridDim.x * 256) >> 5;
  const int b4 = (hl >> 4) & 1, b3 = (hl >> 3) & 1, b2 = (hl >> 2) & 1;
  const unsigned hw = (unsigned)H * (unsigned)W;
  for (int64_t p0 = half_id * 8; p0 < npix; p0 += nhalf * 8) {
    float v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = 0.f;
    for (int c = hl * 4; c < cin; c += 128) {
      const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + (int64_t)cin + c), w2 = *(const float4*)(w + 2 * (int64_t)cin + c);
      float4 xv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t p = p0 + k < npix ? p0 + k : npix - 1;
        unsigned pix, oxu;
        const int b = (int)fd_divmod((unsigned)p, d_hw, pix);
        const int oy = (int)fd_divmod(pix, d_w, oxu);
        const float4 xr = st_ld4<XS>(x, p * cin + c);
        const float4* q = (const float4*)(ss + ((int64_t)b * cin + c) * 2);
        const float4 q0 = q[0], q1 = q[1];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = res_up2_sample<XS>(res, b, oy, (int)oxu, H, W, res_ld, c);
        float t0 = q0.x * xr.x + q0.y, t1 = q0.z * xr.y + q0.w, t2 = q1.x * xr.z + q1.y, t3 = q1.z * xr.w + q1.w;
        t0 += r.x; t1 += r.y; t2 += r.z; t3 += r.w;
        xv[k] = make_float4(t0 * out_scale, t1 * out_scale, t2 * out_scale, t3 * out_scale);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k * 3 + 0] += xv[k].x * w0.x + xv[k].y * w0.y + xv[k].z * w0.z + xv[k].w * w0.w;
        v[k * 3 + 1] += xv[k].x * w1.x + xv[k].y * w1.y + xv[k].z * w1.z + xv[k].w * w1.w;
        v[k * 3 + 2] += xv[k].x * w2.x + xv[k].y * w2.y + xv[k].z * w2.z + xv[k].w * w2.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float keep = b4 ? v[12 + i] : v[i], send = b4 ? v[i] : v[12 + i];
      v[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float keep = b3 ? v[6 + i] : v[i], send = b3 ? v[i] : v[6 + i];
      v[i] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float keep = b2 ? v[3 + i] : v[i], send = b2 ? v[i] : v[3 + i];
      v[i] = keep + __shfl_xor(send, 4, 64);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] += __shfl_xor(v[i], 2, 64);
      v[i] += __shfl_xor(v[i], 1, 64);
    }
    const int k = b4 * 4 + b3 * 2 + b2, j = hl & 3;
    if (j < 3 && p0 + k < npix) {
      const float r = j == 0 ? v[0] : (j == 1 ? v[1] : v[2]);
      y[(p0 + k) * 3 + j] = r * wscale + (bias ? bias[j] : 0.f);
    }
  }
}
// x [B][H][W][cin] (storage x_st, dense), scale_shift [B][cin][2], res [B][H/2][W/2][res_ld] or NULL (same storage type),
// w [3][cin], bias [3] or NULL -> y [B][H][W][3] fp32
extern "C" int ppst_torgb_apply_st(const void* x, const void* scale_shift, const void* res, int res_ld, float out_scale, const void* w,
                                   const void* bias, void* y, int B, int H, int W, int cin, float wscale, int x_st, void* stream) {
  if ((unsigned)x_st > 2u) return PPST_EINVAL;
  if (B < 0 || H <= 0 || W <= 0 || cin <= 0 || cin % 4 || (res && (H % 2 || W % 2 || res_ld < cin || res_ld % 4))) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !scale_shift || !w || !y) return PPST_ENULL;
  const int64_t npix = (int64_t)B * H * W;
  if (npix > PPST_IDX32_MAX) return PPST_EINVAL;
  if (((uintptr_t)x | (uintptr_t)res) % (x_st ? 8 : 16) || ((uintptr_t)w | (uintptr_t)scale_shift) % 16) return PPST_EINVAL;
  PPST_ST_SWITCH(x_st, PPST_LAUNCH(conv1x1_cout3_apply_kernel<ST_>, dim3(grid_for(cdiv64(npix, 8) * 32)), dim3(256), 0, as_stream(stream), x,
                                   (const float*)scale_shift, res, res_ld, out_scale, (const float*)w, (const float*)bias, (float*)y, npix,
                                   H, W, cin, wscale, make_fastdiv((unsigned)H * (unsigned)W), make_fastdiv((unsigned)W)));
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------------------- misc glue --
__global__ __launch_bounds__(256) void lerp_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                   int64_t n, float r) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = a[i] * (1.f - r) + b[i] * r;  // util/util.py:35 (same operation order)
}
extern "C" int ppst_lerp(const void* a, const void* b, void* y, int64_t n, float r, void* stream) {
  if (n < 0) return PPST_EINVAL;
  if (n == 0) return PPST_OK;
  if (!a || !b || !y) return PPST_ENULL;
  PPST_LAUNCH(lerp_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), (const float*)a, (const float*)b, (float*)y, n, r);
  return PPST_LAUNCH_CHECK();
}

template <int YS = PPST_ST_F32>
__global__ __launch_bounds__(256) void spatial_mod_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                          const float* __restrict__ bi, void* __restrict__ y,
                                                          int C, unsigned total, FastDiv d_c, FastDiv d_hw) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq;
    const int64_t bp = fd_divmod((unsigned)t64, d_c, cq);
    const int c = (int)cq * 4;
    const int b = (int)fd_div((unsigned)bp, d_hw);
    float4 v = *(const float4*)(x + bp * C + c);
    float4 s = *(const float4*)(sc + (int64_t)b * C + c);
    float4 o = *(const float4*)(bi + (int64_t)b * C + c);
    st_st4<YS>(y, bp * C + c, make_float4(v.x * s.x + o.x, v.y * s.y + o.y, v.z * s.z + o.z, v.w * s.w + o.w));
  }
}
extern "C" int ppst_spatial_modulation_st(const void* x, const void* scale, const void* bias, void* y, int B, int64_t hw, int C,
                                          int y_st, void* stream) {
  if ((unsigned)y_st > 2u) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || C <= 0 || C % 4) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !scale || !bias || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * hw * (C / 4);
  if (total > PPST_IDX32_MAX || hw > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_ST_SWITCH(y_st, PPST_LAUNCH(spatial_mod_kernel<ST_>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x,
                                   (const float*)scale, (const float*)bias, y, C, (unsigned)total, make_fastdiv(C / 4),
                                   make_fastdiv((unsigned)hw)));
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_spatial_modulation(const void* x, const void* scale, const void* bias, void* y, int B, int64_t hw, int C,
                                       void* stream) {
  return ppst_spatial_modulation_st(x, scale, bias, y, B, hw, C, PPST_ST_F32, stream);
}

// util.tensor2im (util/util.py:125-131): NCHW [-1,1] -> HWC uint8, clip then truncate.
// The reference computes (x + 1) / 2.0 * 255.0 in fp32 numpy: same operation order here.
__global__ __launch_bounds__(256) void tensor2im_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int C,
                                                        int64_t P, int64_t total) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int c = (int)(t % C);
    int64_t bp = t / C;
    int64_t b = bp / P, p = bp - b * P;
    float v = (x[(b * C + c) * P + p] + 1.f) / 2.0f * 255.0f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    y[t] = (unsigned char)v;
  }
}
extern "C" int ppst_tensor2im_u8(const void* x, void* y, int B, int C, int H, int W, void* stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t P = (int64_t)H * W, total = (int64_t)B * P * C;
  PPST_LAUNCH(tensor2im_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x,
                     (unsigned char*)y, C, P, total);
  return PPST_LAUNCH_CHECK();
}
