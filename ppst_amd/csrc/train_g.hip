// Backward kernels of the generator / encoder update of the GAN train step (SURVEY.md section 8 a14;
// reference: optimizers/ppst_optimizer.py:73-94 `g_loss.backward()` through
// models/ppst_model.py:161-235, i.e. torch autograd of InstanceNorm2d + StyleMod
// (stylegan2_layers.py:361-374, 414-437), ReflectionPad2d / ReplicationPad2d, F.interpolate(bilinear),
// adaptive avg / max pooling (encoder_col.py:150-251), F.normalize / util.normalize, L1Loss, softmax, PReLU).
// All HBM-bound, NHWC fp32, one pass each; the conv input / weight gradients are the MFMA kernels of
// conv_mfma.hip / train.hip driven by other step tables.
#include "common.h"

#define TG_GRID_CAP (256 * 16)
static inline unsigned tg_grid(int64_t work, int threads = 256) {
  int64_t b = cdiv64(work, threads);
  if (b > TG_GRID_CAP) b = TG_GRID_CAP;
  if (b < 1) b = 1;
  return (unsigned)b;
}
__device__ __forceinline__ float lrelu_gate(float ref) { return (ref > 0.f ? 1.f : 0.2f) * 1.41421356237309515f; }

static inline int tg_pix_chunk(int B, int64_t hw) {   // batch-independent, like pix_chunk (elementwise.hip)
  (void)B;
  int chunk = 1024;
  while (chunk > 64 && cdiv64(hw, chunk) < 2048) chunk >>= 1;
  return chunk;
}

// ---------------------------------------------------- instance norm backward --
// out = n * A + s1,  n = (y - mean) * rstd  (A = style s0 + 1 or 1).  With g = dL/dout:
//   dy = rstd*A * (g - mean(g) - n * mean(g*n)),  ds0 = sum g*n,  ds1 = sum g.
// Pass 1 (this kernel): per-(b, c) partial sums (sum g', sum g'*y) over pixel chunks, g' = g or, with `gate`
// (ConvLayer norm='in': the norm precedes the activation), g * lrelu'(gate).
__global__ __launch_bounds__(256) void dual_stats_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                         const float* __restrict__ gate, float* __restrict__ partial, int P, int C,
                                                         int g_ld, int y_ld, int gate_ld, int nchunks, int PIX_CHUNK) {
  __shared__ float s0[256], s1[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pbeg = chunk * PIX_CHUNK;
  const int pend = (pbeg + PIX_CHUNK < P) ? pbeg + PIX_CHUNK : P;
  int lanesC = 1;
  while (lanesC < C && lanesC < 256) lanesC <<= 1;
  const int rows = 256 / lanesC;
  const int cl = threadIdx.x % lanesC, pr = threadIdx.x / lanesC;
  for (int cbase = 0; cbase < C; cbase += lanesC) {
    const int c = cbase + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
      for (int p = pbeg + pr; p < pend; p += rows) {
        const int64_t bp = (int64_t)b * P + p;
        float gv = g[bp * g_ld + c];
        const float yv = y[bp * y_ld + c];
        if (gate) gv *= lrelu_gate(gate[bp * gate_ld + c]);
        a0 += gv;
        a1 += gv * yv;
      }
    }
    s0[threadIdx.x] = a0;
    s1[threadIdx.x] = a1;
    __syncthreads();
    if (pr == 0 && c < C) {
      for (int r = 1; r < rows; ++r) { a0 += s0[r * lanesC + cl]; a1 += s1[r * lanesC + cl]; }
      float* o = partial + (((int64_t)b * nchunks + chunk) * C + c) * 2;
      o[0] = a0;
      o[1] = a1;
    }
    __syncthreads();
  }
}
// four channels per thread (C and the pixel strides multiples of 4, 16-byte aligned tensors: every tensor of the train step)
// ST (round 5: half-precision storage of the TRAINING activations and their gradients in precision mode 1): storage type of g, y and
// gate; the sums are fp32 either way
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void dual_stats4_kernel(const void* __restrict__ g, const void* __restrict__ y,
                                                          const void* __restrict__ gate, float* __restrict__ partial, int P, int C4,
                                                          int g_ld4, int y_ld4, int gate_ld4, int nchunks, int PIX_CHUNK) {
  __shared__ float4 s0[256], s1[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pbeg = chunk * PIX_CHUNK;
  const int pend = (pbeg + PIX_CHUNK < P) ? pbeg + PIX_CHUNK : P;
  int lanesC = 1;
  while (lanesC < C4 && lanesC < 256) lanesC <<= 1;
  const int rows = 256 / lanesC;
  const int cl = threadIdx.x % lanesC, pr = threadIdx.x / lanesC;
  for (int cbase = 0; cbase < C4; cbase += lanesC) {
    const int c = cbase + cl;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (c < C4) {
      for (int p = pbeg + pr; p < pend; p += rows) {
        const int64_t bp = (int64_t)b * P + p;
        float4 gv = st_ld4<ST>(g, (bp * g_ld4 + c) * 4);
        const float4 yv = st_ld4<ST>(y, (bp * y_ld4 + c) * 4);
        if (gate) {
          const float4 t = st_ld4<ST>(gate, (bp * gate_ld4 + c) * 4);
          gv.x *= lrelu_gate(t.x); gv.y *= lrelu_gate(t.y); gv.z *= lrelu_gate(t.z); gv.w *= lrelu_gate(t.w);
        }
        a0.x += gv.x; a0.y += gv.y; a0.z += gv.z; a0.w += gv.w;
        a1.x += gv.x * yv.x; a1.y += gv.y * yv.y; a1.z += gv.z * yv.z; a1.w += gv.w * yv.w;
      }
    }
    s0[threadIdx.x] = a0;
    s1[threadIdx.x] = a1;
    __syncthreads();
    if (pr == 0 && c < C4) {
      for (int r = 1; r < rows; ++r) {
        const float4 u = s0[r * lanesC + cl], w = s1[r * lanesC + cl];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w;
      }
      float4* o = (float4*)(partial + (((int64_t)b * nchunks + chunk) * C4 * 4 + c * 4) * 2);
      o[0] = make_float4(a0.x, a1.x, a0.y, a1.y);
      o[1] = make_float4(a0.z, a1.z, a0.w, a1.w);
    }
    __syncthreads();
  }
}
extern "C" int ppst_dual_stats_st(const void* g, const void* y, const void* gate, void* partial, int B, int64_t hw, int C, int g_ld,
                                  int y_ld, int gate_ld, int* n_partials, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || hw > 0x7fffffffll || C <= 0 || g_ld < C || y_ld < C || (gate && gate_ld < C)) return PPST_EINVAL;
  const int chunk = tg_pix_chunk(B, hw);
  const int nchunks = (int)cdiv64(hw, chunk);
  if (n_partials) *n_partials = nchunks;
  if (!g && !partial) return PPST_OK;  // size query
  if (B == 0) return PPST_OK;
  if (!g || !y || !partial) return PPST_ENULL;
  if (C % 4 == 0 && g_ld % 4 == 0 && y_ld % 4 == 0 && (!gate || gate_ld % 4 == 0) &&
      ((uintptr_t)g | (uintptr_t)y | (uintptr_t)gate) % (st ? 8 : 16) == 0 && (uintptr_t)partial % 16 == 0) {
    PPST_ST_SWITCH(st, PPST_LAUNCH(dual_stats4_kernel<ST_>, dim3(nchunks, B), dim3(256), 0, as_stream(stream), g, y, gate, (float*)partial,
                                   (int)hw, C / 4, g_ld / 4, y_ld / 4, gate_ld / 4, nchunks, chunk));
    return PPST_LAUNCH_CHECK();
  }
  if (st) return PPST_EINVAL;      // half storage: the four-channel form only
  PPST_LAUNCH(dual_stats_kernel, dim3(nchunks, B), dim3(256), 0, as_stream(stream), (const float*)g, (const float*)y,
              (const float*)gate, (float*)partial, (int)hw, C, g_ld, y_ld, gate_ld, nchunks, chunk);
  return PPST_LAUNCH_CHECK();
}

extern "C" int ppst_dual_stats(const void* g, const void* y, const void* gate, void* partial, int B, int64_t hw, int C, int g_ld,
                               int y_ld, int gate_ld, int* n_partials, void* stream) {
  return ppst_dual_stats_st(g, y, gate, partial, B, hw, C, g_ld, y_ld, gate_ld, n_partials, PPST_ST_F32, stream);
}

// Pass 2: reduce the partials (double accumulation) -> coef[b][c] = (k0, k1, k2) with dy = k0*g' + k1*y + k2 and,
// with `style`, dstyle[b] = (ds0[0..C), ds1[0..C)) (gradient of the StyleMod linear's output, stylegan2_layers.py:368-374).
// mean_rstd[b][c] = (mean, rstd) of the forward (ppst_in_finalize_train).  Without a norm (mean_rstd == null) the
// partials are plain per-channel sums: coef is not written and dstyle = (sum g*y, sum g) -- the gradient of the
// SpatialCodeModulation scale / shift (generator.py:80-91).
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const float* __restrict__ partial, int n_partials,
                                                              const float* __restrict__ mean_rstd, const float* __restrict__ style,
                                                              int style_ld, float* __restrict__ coef, float* __restrict__ dstyle,
                                                              int B, int C, double count) {
  __shared__ double sm[8][32][2];
  const int cgroups = (C + 31) / 32;
  const int b = blockIdx.x / cgroups, c0 = (blockIdx.x % cgroups) * 32;
  const int cl = threadIdx.x & 31, kk = threadIdx.x >> 5;
  const int c = c0 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float2* p = (const float2*)partial + ((int64_t)b * n_partials * C + c);
    int k = kk;
    for (; k + 24 < n_partials; k += 32) {                  // four rows in flight per thread (as in_finalize_kernel), fixed order
      const float2 v0 = p[(int64_t)k * C], v1 = p[(int64_t)(k + 8) * C], v2 = p[(int64_t)(k + 16) * C], v3 = p[(int64_t)(k + 24) * C];
      s += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; k < n_partials; k += 8) {
      float2 v = p[(int64_t)k * C];
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sm[kk][cl][0] = s;
  sm[kk][cl][1] = q;
  __syncthreads();
  if (kk == 0 && c < C) {
    for (int r = 1; r < 8; ++r) { s += sm[r][cl][0]; q += sm[r][cl][1]; }
    if (!mean_rstd) {
      if (dstyle) {
        dstyle[(int64_t)b * 2 * C + c] = (float)q;
        dstyle[(int64_t)b * 2 * C + C + c] = (float)s;
      }
      return;
    }
    const double mean = (double)mean_rstd[((int64_t)b * C + c) * 2], rstd = (double)mean_rstd[((int64_t)b * C + c) * 2 + 1];
    const double A = style ? (double)style[(int64_t)b * style_ld + c] + 1.0 : 1.0;
    const double m1 = s / count, m2 = q / count;
    const double qn = rstd * (m2 - mean * m1);          // mean(g * n)
    const double k0 = rstd * A;
    float* o = coef + ((int64_t)b * C + c) * 4;
    o[0] = (float)k0;
    o[1] = (float)(-k0 * rstd * qn);
    o[2] = (float)(k0 * (mean * rstd * qn - m1));
    o[3] = 0.f;
    if (dstyle) {
      dstyle[(int64_t)b * 2 * C + c] = (float)(qn * count);
      dstyle[(int64_t)b * 2 * C + C + c] = (float)s;
    }
  }
}
extern "C" int ppst_in_bwd_finalize(const void* partial, int n_partials, const void* mean_rstd, const void* style, int style_ld,
                                    void* coef, void* dstyle, int B, int C, double count, void* stream) {
  if (B < 0 || C <= 0 || n_partials <= 0 || count <= 0 || (style && style_ld < C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!partial || (mean_rstd && !coef) || (!mean_rstd && !dstyle)) return PPST_ENULL;
  PPST_LAUNCH(in_bwd_finalize_kernel, dim3(B * cdiv(C, 32)), dim3(256), 0, as_stream(stream), (const float*)partial, n_partials,
              (const float*)mean_rstd, (const float*)style, style_ld, (float*)coef, (float*)dstyle, B, C, count);
  return PPST_LAUNCH_CHECK();
}

// Pass 3: dx = post * (k0 * g' + k1 * y + k2);  g' = g * lrelu'(gate) if gate (norm before activation);
// post = lrelu'(y) if post_gate (StyledConv: the activation precedes the norm, so y is its output) else 1.
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ gate, const float* __restrict__ coef,
                                                           float* __restrict__ dx, unsigned hw, int C, int g_ld, int y_ld, int gate_ld,
                                                           int dx_ld, int post_gate, unsigned total, FastDiv d_c, FastDiv d_hw) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c, cq);
    const int c = (int)cq;
    const int b = (int)fd_div(bpu, d_hw);
    const int64_t bp = bpu;
    const float* k = coef + ((int64_t)b * C + c) * 4;
    float gv = g[bp * g_ld + c];
    const float yv = y[bp * y_ld + c];
    if (gate) gv *= lrelu_gate(gate[bp * gate_ld + c]);
    float o = k[0] * gv + k[1] * yv + k[2];
    if (post_gate) o *= lrelu_gate(yv);
    dx[bp * dx_ld + c] = o;
  }
}
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void in_bwd_apply4_kernel(const void* __restrict__ g, const void* __restrict__ y,
                                                            const void* __restrict__ gate, const float4* __restrict__ coef,
                                                            void* __restrict__ dx, unsigned hw, int C4, int g_ld4, int y_ld4,
                                                            int gate_ld4, int dx_ld4, int post_gate, unsigned total, FastDiv d_c,
                                                            FastDiv d_hw) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c, cq);
    const int c = (int)cq;
    const int b = (int)fd_div(bpu, d_hw);
    const int64_t bp = bpu;
    const float4* k = coef + ((int64_t)b * C4 + c) * 4;      // (k0, k1, k2, -) of the thread's four channels
    const float4 k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
    float4 gv = st_ld4<ST>(g, (bp * g_ld4 + c) * 4);
    const float4 yv = st_ld4<ST>(y, (bp * y_ld4 + c) * 4);
    if (gate) {
      const float4 t = st_ld4<ST>(gate, (bp * gate_ld4 + c) * 4);
      gv.x *= lrelu_gate(t.x); gv.y *= lrelu_gate(t.y); gv.z *= lrelu_gate(t.z); gv.w *= lrelu_gate(t.w);
    }
    float4 o = make_float4(k0.x * gv.x + k0.y * yv.x + k0.z, k1.x * gv.y + k1.y * yv.y + k1.z, k2.x * gv.z + k2.y * yv.z + k2.z,
                           k3.x * gv.w + k3.y * yv.w + k3.z);
    if (post_gate) { o.x *= lrelu_gate(yv.x); o.y *= lrelu_gate(yv.y); o.z *= lrelu_gate(yv.z); o.w *= lrelu_gate(yv.w); }
    st_st4<ST>(dx, (bp * dx_ld4 + c) * 4, o);
  }
}
extern "C" int ppst_in_bwd_apply_st(const void* g, const void* y, const void* gate, const void* coef, void* dx, int B, int64_t hw, int C,
                                    int g_ld, int y_ld, int gate_ld, int dx_ld, int post_gate, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || C <= 0 || g_ld < C || y_ld < C || dx_ld < C || (gate && gate_ld < C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!g || !y || !coef || !dx) return PPST_ENULL;
  const int64_t total = (int64_t)B * hw * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  if (C % 4 == 0 && g_ld % 4 == 0 && y_ld % 4 == 0 && dx_ld % 4 == 0 && (!gate || gate_ld % 4 == 0) &&
      ((uintptr_t)g | (uintptr_t)y | (uintptr_t)gate | (uintptr_t)dx) % (st ? 8 : 16) == 0 && (uintptr_t)coef % 16 == 0) {
    PPST_ST_SWITCH(st, PPST_LAUNCH(in_bwd_apply4_kernel<ST_>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), g, y, gate,
                                   (const float4*)coef, dx, (unsigned)hw, C / 4, g_ld / 4, y_ld / 4, gate_ld / 4, dx_ld / 4, post_gate,
                                   (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)hw)));
    return PPST_LAUNCH_CHECK();
  }
  if (st) return PPST_EINVAL;
  PPST_LAUNCH(in_bwd_apply_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)g, (const float*)y,
              (const float*)gate, (const float*)coef, (float*)dx, (unsigned)hw, C, g_ld, y_ld, gate_ld, dx_ld, post_gate,
              (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)hw));
  return PPST_LAUNCH_CHECK();
}

extern "C" int ppst_in_bwd_apply(const void* g, const void* y, const void* gate, const void* coef, void* dx, int B, int64_t hw, int C,
                                 int g_ld, int y_ld, int gate_ld, int dx_ld, int post_gate, void* stream) {
  return ppst_in_bwd_apply_st(g, y, gate, coef, dx, B, hw, C, g_ld, y_ld, gate_ld, dx_ld, post_gate, PPST_ST_F32, stream);
}

// PReLU (single slope, nn.PReLU()) applied to a normalised tensor: out = prelu(a*y + s) (feature heads,
// generator.py:10-32,174-238).  Given g = dL/dout: gpre = g * (z >= 0 ? 1 : slope), z = a*y + s;
// dslope partial = sum g * z * [z < 0].  Writes gpre and per-block partial sums of dslope.
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                        const float* __restrict__ ss, const float* __restrict__ res,
                                                        const float* __restrict__ prelu, float* __restrict__ gpre,
                                                        float* __restrict__ dslope_partial, unsigned hw, int C, int g_ld, int y_ld,
                                                        int res_ld, unsigned total, FastDiv d_c, FastDiv d_hw) {
  __shared__ float red[4];
  const float slope = prelu[0];
  float acc = 0.f;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cq;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c, cq);
    const int c = (int)cq;
    const int b = (int)fd_div(bpu, d_hw);
    const int64_t bp = bpu;
    float z = y[bp * y_ld + c];
    if (ss) z = ss[((int64_t)b * C + c) * 2] * z + ss[((int64_t)b * C + c) * 2 + 1];
    if (res) z += res[bp * res_ld + c];       // residual joins before the activation (ResidualBlock, generator.py:28-31)
    const float gv = g[bp * g_ld + c];
    gpre[bp * C + c] = z >= 0.f ? gv : gv * slope;
    if (z < 0.f) acc += gv * z;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dslope_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int64_t ppst_prelu_bwd_ws(int64_t total) { return (int64_t)tg_grid(total) * (int64_t)sizeof(float); }
extern "C" int ppst_prelu_bwd(const void* g, const void* y, const void* scale_shift, const void* res, const void* prelu, void* gpre,
                              void* ws, int B, int64_t hw, int C, int g_ld, int y_ld, int res_ld, void* stream) {
  if (B < 0 || hw <= 0 || C <= 0 || g_ld < C || y_ld < C || (res && res_ld < C)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!g || !y || !prelu || !gpre || !ws) return PPST_ENULL;
  const int64_t total = (int64_t)B * hw * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(prelu_bwd_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)g, (const float*)y,
              (const float*)scale_shift, (const float*)res, (const float*)prelu, (float*)gpre, (float*)ws, (unsigned)hw, C, g_ld,
              y_ld, res_ld, (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)hw));
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------------ padding --
__device__ __forceinline__ int pad_src(int t, int n, int mode) {   // padded coordinate (origin at the unpadded tensor) -> source
  if (t >= 0 && t < n) return t;
  if (mode == PPST_PAD_ZERO) return -1;
  if (mode == PPST_PAD_REFLECT) {
    if (t < 0) t = -t;
    if (t >= n) t = 2 * (n - 1) - t;
    return (t >= 0 && t < n) ? t : -1;
  }
  return t < 0 ? 0 : n - 1;
}
// y[b][ty][tx][c] = x[b][src(ty - py0)][src(tx - px0)][c]  (F.pad zero / reflect / replicate).  VT = float4 when C, x_ld are
// multiples of 4 and the pointers 16-byte aligned (every tensor of the train step): C and x_ld are then counted in float4s.
template <typename VT>
__global__ __launch_bounds__(256) void pad2d_kernel(const VT* __restrict__ x, VT* __restrict__ y, int H, int W, int C, int x_ld,
                                                    int OH, int OW, int py0, int px0, int mode, unsigned total, FastDiv d_c,
                                                    FastDiv d_ow, FastDiv d_oh) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, tx, ty;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_ow, tx);
    const unsigned b = fd_divmod(r, d_oh, ty);
    const int sy = pad_src((int)ty - py0, H, mode), sx = pad_src((int)tx - px0, W, mode);
    VT v = {};
    if (sy >= 0 && sx >= 0) v = x[(((int64_t)b * H + sy) * W + sx) * x_ld + c];
    y[t64] = v;
  }
}
extern "C" int ppst_pad2d_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int py0, int py1, int px0, int px1, int mode,
                             int st, void* stream);
extern "C" int ppst_pad2d(const void* x, void* y, int B, int H, int W, int C, int x_ld, int py0, int py1, int px0, int px1, int mode,
                          void* stream) {
  return ppst_pad2d_st(x, y, B, H, W, C, x_ld, py0, py1, px0, px1, mode, PPST_ST_F32, stream);
}
// st: storage type of x and y -- pure data movement: a half tensor's four channels travel as one 8-byte element
extern "C" int ppst_pad2d_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int py0, int py1, int px0, int px1, int mode,
                             int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  const int OH = H + py0 + py1, OW = W + px0 + px1;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || x_ld < C || OH <= 0 || OW <= 0 || mode < 0 || mode > 2) return PPST_EINVAL;
  if (mode == PPST_PAD_REFLECT && (py0 >= H || py1 >= H || px0 >= W || px1 >= W)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  const int64_t total = (int64_t)B * OH * OW * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  if (st) {
    if (C % 4 || x_ld % 4 || ((uintptr_t)x | (uintptr_t)y) % 8) return PPST_EINVAL;
    PPST_LAUNCH(pad2d_kernel<uint2>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), (const uint2*)x, (uint2*)y, H, W, C / 4,
                x_ld / 4, OH, OW, py0, px0, mode, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)OW),
                make_fastdiv((unsigned)OH));
    return PPST_LAUNCH_CHECK();
  }
  if (C % 4 == 0 && x_ld % 4 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0) {
    PPST_LAUNCH(pad2d_kernel<float4>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), (const float4*)x, (float4*)y, H, W, C / 4,
                x_ld / 4, OH, OW, py0, px0, mode, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)OW),
                make_fastdiv((unsigned)OH));
    return PPST_LAUNCH_CHECK();
  }
  PPST_LAUNCH(pad2d_kernel<float>, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, H, W, C, x_ld, OH, OW,
              py0, px0, mode, (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)OW), make_fastdiv((unsigned)OH));
  return PPST_LAUNCH_CHECK();
}
// adjoint of ppst_pad2d: dx[b][y][x][c] = sum of dy over every padded position whose source is (y, x).
// Candidates per axis: the interior copy plus the (py0 + py1) border positions -- tested, not enumerated in closed form.
__device__ __forceinline__ void pad_acc(float& a, float v) { a += v; }
__device__ __forceinline__ void pad_acc(float4& a, float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
template <typename VT>
__global__ __launch_bounds__(256) void pad2d_bwd_kernel(const VT* __restrict__ dy, VT* __restrict__ dx, int H, int W, int C,
                                                        int OH, int OW, int py0, int py1, int px0, int px1, int mode, unsigned total,
                                                        FastDiv d_c, FastDiv d_w, FastDiv d_h) {
  const int ny = 1 + (py0 > 0 ? py0 : 0) + (py1 > 0 ? py1 : 0), nx = 1 + (px0 > 0 ? px0 : 0) + (px1 > 0 ? px1 : 0);
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, xx, yy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_w, xx);
    const unsigned b = fd_divmod(r, d_h, yy);
    const VT* base = dy + (int64_t)b * OH * OW * C + c;
    VT acc = {};
    for (int iy = 0; iy < ny; ++iy) {
      // candidate padded row (coordinates relative to the unpadded origin): itself, the top border rows, the bottom ones
      int ty = iy == 0 ? (int)yy : (iy <= (py0 > 0 ? py0 : 0) ? -iy : H - 1 + (iy - (py0 > 0 ? py0 : 0)));
      if (ty + py0 < 0 || ty + py0 >= OH) continue;       // cropped away (negative pad)
      if (pad_src(ty, H, mode) != (int)yy) continue;
      for (int ix = 0; ix < nx; ++ix) {
        int tx = ix == 0 ? (int)xx : (ix <= (px0 > 0 ? px0 : 0) ? -ix : W - 1 + (ix - (px0 > 0 ? px0 : 0)));
        if (tx + px0 < 0 || tx + px0 >= OW) continue;
        if (pad_src(tx, W, mode) != (int)xx) continue;
        pad_acc(acc, base[((int64_t)(ty + py0) * OW + (tx + px0)) * C]);
      }
    }
    dx[t64] = acc;
  }
}
// half storage: the same sums (fp32) over four channels per thread, one rounding at the store
template <int ST>
__global__ __launch_bounds__(256) void pad2d_bwd_st_kernel(const void* __restrict__ dy, void* __restrict__ dx, int H, int W, int C4,
                                                           int OH, int OW, int py0, int py1, int px0, int px1, int mode, unsigned total,
                                                           FastDiv d_c, FastDiv d_w, FastDiv d_h) {
  const int ny = 1 + (py0 > 0 ? py0 : 0) + (py1 > 0 ? py1 : 0), nx = 1 + (px0 > 0 ? px0 : 0) + (px1 > 0 ? px1 : 0);
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, xx, yy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_w, xx);
    const unsigned b = fd_divmod(r, d_h, yy);
    const int64_t base = (int64_t)b * OH * OW * C4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = 0; iy < ny; ++iy) {
      int ty = iy == 0 ? (int)yy : (iy <= (py0 > 0 ? py0 : 0) ? -iy : H - 1 + (iy - (py0 > 0 ? py0 : 0)));
      if (ty + py0 < 0 || ty + py0 >= OH) continue;
      if (pad_src(ty, H, mode) != (int)yy) continue;
      for (int ix = 0; ix < nx; ++ix) {
        int tx = ix == 0 ? (int)xx : (ix <= (px0 > 0 ? px0 : 0) ? -ix : W - 1 + (ix - (px0 > 0 ? px0 : 0)));
        if (tx + px0 < 0 || tx + px0 >= OW) continue;
        if (pad_src(tx, W, mode) != (int)xx) continue;
        pad_acc(acc, st_ld4<ST>(dy, (base + ((int64_t)(ty + py0) * OW + (tx + px0)) * C4) * 4));
      }
    }
    st_st4<ST>(dx, (int64_t)t64 * 4, acc);
  }
}
extern "C" int ppst_pad2d_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int py0, int py1, int px0, int px1, int mode,
                                 int st, void* stream);
extern "C" int ppst_pad2d_bwd(const void* dy, void* dx, int B, int H, int W, int C, int py0, int py1, int px0, int px1, int mode,
                              void* stream) {
  return ppst_pad2d_bwd_st(dy, dx, B, H, W, C, py0, py1, px0, px1, mode, PPST_ST_F32, stream);
}
extern "C" int ppst_pad2d_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int py0, int py1, int px0, int px1, int mode,
                                 int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  const int OH = H + py0 + py1, OW = W + px0 + px1;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || mode < 0 || mode > 2) return PPST_EINVAL;
  if (mode == PPST_PAD_REFLECT && (py0 >= H || py1 >= H || px0 >= W || px1 >= W)) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!dy || !dx) return PPST_ENULL;
  const int64_t total = (int64_t)B * H * W * C;
  if (total > PPST_IDX32_MAX || (int64_t)B * OH * OW * C > PPST_IDX32_MAX) return PPST_EINVAL;
  if (st) {
    if (C % 4 || ((uintptr_t)dy | (uintptr_t)dx) % 8) return PPST_EINVAL;
    PPST_ST_SWITCH(st, PPST_LAUNCH(pad2d_bwd_st_kernel<ST_>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), dy, dx, H, W, C / 4, OH,
                                   OW, py0, py1, px0, px1, mode, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)),
                                   make_fastdiv((unsigned)W), make_fastdiv((unsigned)H)));
    return PPST_LAUNCH_CHECK();
  }
  if (C % 4 == 0 && ((uintptr_t)dy | (uintptr_t)dx) % 16 == 0) {     // four channels per thread (same sums, same order)
    PPST_LAUNCH(pad2d_bwd_kernel<float4>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), (const float4*)dy, (float4*)dx, H, W,
                C / 4, OH, OW, py0, py1, px0, px1, mode, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)W),
                make_fastdiv((unsigned)H));
    return PPST_LAUNCH_CHECK();
  }
  PPST_LAUNCH(pad2d_bwd_kernel<float>, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)dy, (float*)dx, H, W, C, OH, OW,
              py0, py1, px0, px1, mode, (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)W),
              make_fastdiv((unsigned)H));
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------- resize / pooling adjoints --
// adjoint of ppst_bilinear (F.interpolate bilinear, align_corners=False): dx (zero-initialised by the caller)
// += weights * dy.  Scatter with float atomics (memory-side adds on gfx950; two contributions commute).
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                                           int OH, int OW, int dy_ld, int dx_ld, float sy, float sx, unsigned total,
                                                           FastDiv d_c, FastDiv d_ow, FastDiv d_oh) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, ox, oy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_ow, ox);
    const unsigned b = fd_divmod(r, d_oh, oy);
    const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float gv = dy[(((int64_t)b * OH + oy) * OW + ox) * dy_ld + c];
    float* base = dx + (int64_t)b * H * W * dx_ld + c;
    atomicAdd(base + ((int64_t)y0 * W + x0) * dx_ld, hy * hx * gv);
    atomicAdd(base + ((int64_t)y0 * W + x1) * dx_ld, hy * lx * gv);
    atomicAdd(base + ((int64_t)y1 * W + x0) * dx_ld, ly * hx * gv);
    atomicAdd(base + ((int64_t)y1 * W + x1) * dx_ld, ly * lx * gv);
  }
}
// Gather form (C % 4 == 0, 16-B aligned rows): one thread = 4 channels of one INPUT pixel; it walks the output pixels whose
// 2x2 footprint can touch it (a generous index window; each tap re-derives the forward's (y0, y1, ly) and takes the weight
// that lands on this pixel), in a fixed order: no atomics, bit-reproducible, and the 4 x over-read of dy stays in L2.
// (The scatter form above cost 0.3 ms per launch on the x8 / x4 upsamplings of the train step.)
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void bilinear_bwd_gather_kernel(const void* __restrict__ dy, void* __restrict__ dx, int H, int W,
                                                                  int C4, int OH, int OW, int dy_ld4, int dx_ld4, float sy, float sx,
                                                                  unsigned total, FastDiv d_c, FastDiv d_w, FastDiv d_h) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4, xx, yy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c4);
    r = fd_divmod(r, d_w, xx);
    const unsigned b = fd_divmod(r, d_h, yy);
    const int y = (int)yy, x = (int)xx;
    // output rows / columns whose source coordinate lies in (y - 1, y + 1): o in ((y - 0.5) / s - 0.5, (y + 1.5) / s - 0.5)
    int oy_lo = (int)floorf(((float)y - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)y + 1.5f) / sy - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)x - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)x + 1.5f) / sx - 0.5f) + 1;
    if (y == 0) oy_lo = 0;                       // the clamp fy = max(., 0) sends every row above to y0 = 0
    if (x == 0) ox_lo = 0;
    oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0); oy_hi = min(oy_hi, OH - 1); ox_hi = min(ox_hi, OW - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f);
      const int y0 = min((int)fy, H - 1), y1 = y0 + (y0 < H - 1);
      const float ly = fy - (float)y0;
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      const int64_t row = (((int64_t)b * OH + oy) * OW) * dy_ld4 + c4;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
        const int x0 = min((int)fx, W - 1), x1 = x0 + (x0 < W - 1);
        const float lx = fx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx == 0.f) continue;
        const float4 g = st_ld4<ST>(dy, (row + (int64_t)ox * dy_ld4) * 4);
        // the forward's four products hy*hx, hy*lx, ly*hx, ly*lx: when both corners of an axis clamp onto this pixel their
        // weights add (1 - l) + l; the scatter form adds the two products separately -- equal to rounding
        const float w = wy * wx;
        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
      }
    }
    const int64_t d = ((((int64_t)b * H + y) * W + x) * dx_ld4 + c4) * 4;
    const float4 p = st_ld4<ST>(dx, d);
    st_st4<ST>(dx, d, make_float4(p.x + acc.x, p.y + acc.y, p.z + acc.z, p.w + acc.w));
  }
}
extern "C" int ppst_bilinear_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int OH, int OW, int dy_ld, int st,
                                    void* stream);
extern "C" int ppst_bilinear_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int OH, int OW, int dy_ld,
                                 void* stream) {
  return ppst_bilinear_bwd_st(dy, dx, B, H, W, C, dx_ld, OH, OW, dy_ld, PPST_ST_F32, stream);
}
extern "C" int ppst_bilinear_bwd_st(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int OH, int OW, int dy_ld, int st,
                                    void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || dx_ld < C || dy_ld < C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!dy || !dx) return PPST_ENULL;
  if (C % 4 == 0 && dx_ld % 4 == 0 && dy_ld % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) % (st ? 8 : 16)) == 0 &&
      (int64_t)B * H * W * (C / 4) <= PPST_IDX32_MAX) {
    const int64_t t4 = (int64_t)B * H * W * (C / 4);
    PPST_ST_SWITCH(st, PPST_LAUNCH(bilinear_bwd_gather_kernel<ST_>, dim3(tg_grid(t4)), dim3(256), 0, as_stream(stream), dy, dx, H, W,
                                   C / 4, OH, OW, dy_ld / 4, dx_ld / 4, (float)H / (float)OH, (float)W / (float)OW, (unsigned)t4,
                                   make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)W), make_fastdiv((unsigned)H)));
    return PPST_LAUNCH_CHECK();
  }
  if (st) return PPST_EINVAL;
  const int64_t total = (int64_t)B * OH * OW * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(bilinear_bwd_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)dy, (float*)dx, H, W, C, OH, OW,
              dy_ld, dx_ld, (float)H / (float)OH, (float)W / (float)OW, (unsigned)total, make_fastdiv((unsigned)C),
              make_fastdiv((unsigned)OW), make_fastdiv((unsigned)OH));
  return PPST_LAUNCH_CHECK();
}

// adjoint of ppst_avgpool (adaptive_avg_pool2d with an integer factor f): dx[y][x] = dy[y/f][x/f] / f^2
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                                          int f, int dy_ld, int dx_ld, unsigned total, FastDiv d_c, FastDiv d_w,
                                                          FastDiv d_h, FastDiv d_f) {
  const float inv = 1.f / (float)(f * f);
  const int oh = H / f, ow = W / f;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, xx, yy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_w, xx);
    const unsigned b = fd_divmod(r, d_h, yy);
    const unsigned oy = fd_div(yy, d_f), ox = fd_div(xx, d_f);
    dx[(((int64_t)b * H + yy) * W + xx) * dx_ld + c] = dy[(((int64_t)b * oh + oy) * ow + ox) * dy_ld + c] * inv;
  }
}
extern "C" int ppst_avgpool_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dx_ld, int f, int dy_ld, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || f <= 0 || H % f || W % f || dx_ld < C || dy_ld < C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!dy || !dx) return PPST_ENULL;
  const int64_t total = (int64_t)B * H * W * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(avgpool_bwd_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)dy, (float*)dx, H, W, C, f, dy_ld,
              dx_ld, (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)W), make_fastdiv((unsigned)H),
              make_fastdiv((unsigned)f));
  return PPST_LAUNCH_CHECK();
}

// adjoint of ppst_gap_gmp: v = cat(mean_p(m*x), max_p(m*x));  dx[p][c] = m[p] * (g_mean[c] / P + g_max[c] * [p == argmax[c]])
// argmax = the FIRST pixel (row-major) attaining the maximum, as nn.AdaptiveMaxPool2d's backward routes it (ties are real:
// the x8 bilinear upsampling of the warped features has 4x4 plateaus at the image border).  Pass 1 finds it with an
// atomicMin over the pixels equal to the forward's maximum; pass 2 applies.
// (accumulate != 0: dx += ...; several heads pool the same feature map, encoder_col.py:162-245)
__global__ __launch_bounds__(256) void gmp_argmax_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                         const float* __restrict__ v, int* __restrict__ arg, unsigned hw, int C, int ld,
                                                         unsigned total, FastDiv d_c, FastDiv d_hw) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c, c);
    const unsigned b = fd_div(bpu, d_hw);
    const float m = mask ? mask[bpu] : 1.f;
    int* ap = arg + (int64_t)b * C + c;
    const int pix = (int)(bpu - b * hw);
    if (x[(int64_t)bpu * ld + c] * m == v[(int64_t)b * 2 * C + C + c] && pix < *(const volatile int*)ap) atomicMin(ap, pix);
  }
}
__global__ __launch_bounds__(256) void gap_gmp_bwd_kernel(const float* __restrict__ mask, const int* __restrict__ arg,
                                                          const float* __restrict__ g, float* __restrict__ dx, unsigned hw, int C,
                                                          int accumulate, unsigned total, FastDiv d_c, FastDiv d_hw) {
  const float invP = 1.f / (float)hw;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c, c);
    const unsigned b = fd_div(bpu, d_hw);
    const float m = mask ? mask[bpu] : 1.f;
    const bool top = (int)(bpu - b * hw) == arg[(int64_t)b * C + c];
    float o = m * (g[(int64_t)b * 2 * C + c] * invP + (top ? g[(int64_t)b * 2 * C + C + c] : 0.f));
    float* d = dx + (int64_t)bpu * C + c;
    *d = accumulate ? *d + o : o;
  }
}
// 16-B forms of the two passes (C % 4 == 0, 16-B aligned rows): one thread = 4 channels of one pixel
// One thread = 4 channels of GMP_NP consecutive pixels: the first matching pixel of its strip is found in registers, then ONE
// guarded atomicMin per channel (round 2 did the guard read + atomic per ELEMENT: four volatile loads per pixel, 140 us for a
// (4, 512, 512, 32) tensor that a plain read moves in 30).  Needs hw % GMP_NP == 0 (a strip stays inside one image).
#define GMP_NP 16
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void gmp_argmax4_kernel(const void* __restrict__ x, const float* __restrict__ mask,
                                                          const float* __restrict__ v, int* __restrict__ arg, unsigned hw, int C, int ld4,
                                                          unsigned total_strips, FastDiv d_c4, FastDiv d_hwn) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total_strips; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4;
    const unsigned strip = fd_divmod((unsigned)t64, d_c4, c4);     // strip index over all images (hw / GMP_NP strips per image)
    const unsigned b = fd_div(strip, d_hwn);
    const unsigned bp0 = strip * GMP_NP;                            // first (image-major) pixel of the strip
    const int pix0 = (int)(bp0 - b * hw);
    const float4 mv = *(const float4*)(v + (int64_t)b * 2 * C + C + c4 * 4);
    int f0 = -1, f1 = -1, f2 = -1, f3 = -1;
#pragma unroll
    for (int j = GMP_NP - 1; j >= 0; --j) {                         // descending: the last assignment is the first match
      const float m = mask ? mask[bp0 + j] : 1.f;
      const float4 xv = st_ld4<ST>(x, ((int64_t)(bp0 + j) * ld4 + c4) * 4);
      if (xv.x * m == mv.x) f0 = j;
      if (xv.y * m == mv.y) f1 = j;
      if (xv.z * m == mv.z) f2 = j;
      if (xv.w * m == mv.w) f3 = j;
    }
    int* ap = arg + (int64_t)b * C + c4 * 4;
    // masked pooling makes ties massive (every masked-out pixel is 0 = the maximum of an all-negative channel): read the current
    // winner first -- a stale value is only larger, so at worst an unnecessary atomic -- instead of 10^5 atomics on one address
    const volatile int* cur = ap;
    if (f0 >= 0 && pix0 + f0 < cur[0]) atomicMin(ap, pix0 + f0);
    if (f1 >= 0 && pix0 + f1 < cur[1]) atomicMin(ap + 1, pix0 + f1);
    if (f2 >= 0 && pix0 + f2 < cur[2]) atomicMin(ap + 2, pix0 + f2);
    if (f3 >= 0 && pix0 + f3 < cur[3]) atomicMin(ap + 3, pix0 + f3);
  }
}
// (the per-element form: any hw)
__global__ __launch_bounds__(256) void gmp_argmax4e_kernel(const float4* __restrict__ x, const float* __restrict__ mask,
                                                           const float* __restrict__ v, int* __restrict__ arg, unsigned hw, int C, int ld4,
                                                           unsigned total, FastDiv d_c4, FastDiv d_hw) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c4, c4);
    const unsigned b = fd_div(bpu, d_hw);
    const float m = mask ? mask[bpu] : 1.f;
    const float4 xv = x[(int64_t)bpu * ld4 + c4];
    const float4 mv = *(const float4*)(v + (int64_t)b * 2 * C + C + c4 * 4);
    int* ap = arg + (int64_t)b * C + c4 * 4;
    const int pix = (int)(bpu - b * hw);
    const volatile int* cur = ap;
    if (xv.x * m == mv.x && pix < cur[0]) atomicMin(ap, pix);
    if (xv.y * m == mv.y && pix < cur[1]) atomicMin(ap + 1, pix);
    if (xv.z * m == mv.z && pix < cur[2]) atomicMin(ap + 2, pix);
    if (xv.w * m == mv.w && pix < cur[3]) atomicMin(ap + 3, pix);
  }
}
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void gap_gmp_bwd4_kernel(const float* __restrict__ mask, const int* __restrict__ arg,
                                                           const float* __restrict__ g, void* __restrict__ dx, unsigned hw, int C,
                                                           int accumulate, unsigned total, FastDiv d_c4, FastDiv d_hw) {
  const float invP = 1.f / (float)hw;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c4, c4);
    const unsigned b = fd_div(bpu, d_hw);
    const float m = mask ? mask[bpu] : 1.f;
    const int pix = (int)(bpu - b * hw);
    const int4 ar = *(const int4*)(arg + (int64_t)b * C + c4 * 4);
    const float4 ga = *(const float4*)(g + (int64_t)b * 2 * C + c4 * 4), gm = *(const float4*)(g + (int64_t)b * 2 * C + C + c4 * 4);
    float4 o = make_float4(m * (ga.x * invP + (pix == ar.x ? gm.x : 0.f)), m * (ga.y * invP + (pix == ar.y ? gm.y : 0.f)),
                           m * (ga.z * invP + (pix == ar.z ? gm.z : 0.f)), m * (ga.w * invP + (pix == ar.w ? gm.w : 0.f)));
    const int64_t d = ((int64_t)bpu * (C >> 2) + c4) * 4;
    if (accumulate) { const float4 p = st_ld4<ST>(dx, d); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    st_st4<ST>(dx, d, o);
  }
}
extern "C" int ppst_gap_gmp_bwd_st(const void* x, const void* mask, const void* v, const void* g, void* dx, void* arg_ws, int B,
                                   int64_t hw, int C, int ld, int accumulate, int st, void* stream);
extern "C" int ppst_gap_gmp_bwd(const void* x, const void* mask, const void* v, const void* g, void* dx, void* arg_ws, int B,
                                int64_t hw, int C, int ld, int accumulate, void* stream) {
  return ppst_gap_gmp_bwd_st(x, mask, v, g, dx, arg_ws, B, hw, C, ld, accumulate, PPST_ST_F32, stream);
}
// st: storage type of x and dx (v, g, mask fp32)
extern "C" int ppst_gap_gmp_bwd_st(const void* x, const void* mask, const void* v, const void* g, void* dx, void* arg_ws, int B,
                                   int64_t hw, int C, int ld, int accumulate, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || hw > 0x7fffffffll || C <= 0 || ld < C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !v || !g || !dx || !arg_ws) return PPST_ENULL;
  const int64_t total = (int64_t)B * hw * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  hipError_t e = hipMemsetAsync(arg_ws, 0x7f, (size_t)B * C * sizeof(int), as_stream(stream));   // 0x7f7f7f7f: above any pixel index
  if (e != hipSuccess) return (int)e;
  if (st && (C % 4 || ld % 4 || hw % GMP_NP || ((uintptr_t)x | (uintptr_t)dx) % 8 || ((uintptr_t)v | (uintptr_t)g | (uintptr_t)arg_ws) % 16))
    return PPST_EINVAL;          // half storage: the strip form only
  if (C % 4 == 0 && ld % 4 == 0 && (((uintptr_t)v | (uintptr_t)g | (uintptr_t)arg_ws) % 16) == 0 &&
      (((uintptr_t)x | (uintptr_t)dx) % (st ? 8 : 16)) == 0) {
    const int64_t t4 = total / 4;
    if (hw % GMP_NP == 0) {
      const int64_t strips = t4 / GMP_NP;
      PPST_ST_SWITCH(st, PPST_LAUNCH(gmp_argmax4_kernel<ST_>, dim3(tg_grid(strips)), dim3(256), 0, as_stream(stream), x, (const float*)mask,
                                     (const float*)v, (int*)arg_ws, (unsigned)hw, C, ld / 4, (unsigned)strips,
                                     make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)(hw / GMP_NP))));
    } else {
      PPST_LAUNCH(gmp_argmax4e_kernel, dim3(tg_grid(t4)), dim3(256), 0, as_stream(stream), (const float4*)x, (const float*)mask,
                  (const float*)v, (int*)arg_ws, (unsigned)hw, C, ld / 4, (unsigned)t4, make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)hw));
    }
    PPST_ST_SWITCH(st, PPST_LAUNCH(gap_gmp_bwd4_kernel<ST_>, dim3(tg_grid(t4)), dim3(256), 0, as_stream(stream), (const float*)mask,
                                   (const int*)arg_ws, (const float*)g, dx, (unsigned)hw, C, accumulate, (unsigned)t4,
                                   make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)hw)));
    return PPST_LAUNCH_CHECK();
  }
  PPST_LAUNCH(gmp_argmax_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)mask,
              (const float*)v, (int*)arg_ws, (unsigned)hw, C, ld, (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv((unsigned)hw));
  PPST_LAUNCH(gap_gmp_bwd_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)mask, (const int*)arg_ws,
              (const float*)g, (float*)dx, (unsigned)hw, C, accumulate, (unsigned)total, make_fastdiv((unsigned)C),
              make_fastdiv((unsigned)hw));
  return PPST_LAUNCH_CHECK();
}


// ---- the same adjoint for SEVERAL pooling heads of one feature map in one pass (round 5; forward: ppst_gap_gmp_multi): heads h = 0
// plain (with_plain), then one per mask channel of masks [B][hw][nm]; v / g [(h * B + b)][2C].  Per head the arithmetic of the
// single-head kernels; dx = the SUM over heads, written once (eight dense per-head gradients and seven autograd adds per pyramid
// level before).  Strip form only (hw % GMP_NP == 0: every level of the 512 / 256 pyramids).
#define GGM_MAXH 4
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void gmp_argmax_multi4_kernel(const void* __restrict__ x, const float* __restrict__ masks,
                                                                const float* __restrict__ v, int* __restrict__ arg, unsigned hw, int C, int ld4,
                                                                int nm, int with_plain, int B, unsigned total_strips, FastDiv d_c4,
                                                                FastDiv d_hwn) {
  const int nh = nm + with_plain;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total_strips; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4;
    const unsigned strip = fd_divmod((unsigned)t64, d_c4, c4);
    const unsigned b = fd_div(strip, d_hwn);
    const unsigned bp0 = strip * GMP_NP;
    const int pix0 = (int)(bp0 - b * hw);
    float4 mv[GGM_MAXH];
    int f[GGM_MAXH][4];
#pragma unroll
    for (int h = 0; h < GGM_MAXH; ++h) {
      mv[h] = h < nh ? *(const float4*)(v + ((int64_t)h * B + b) * 2 * C + C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      f[h][0] = f[h][1] = f[h][2] = f[h][3] = -1;
    }
#pragma unroll
    for (int j = GMP_NP - 1; j >= 0; --j) {                         // descending: the last assignment is the first match
      const float4 xv = st_ld4<ST>(x, ((int64_t)(bp0 + j) * ld4 + c4) * 4);
#pragma unroll
      for (int h = 0; h < GGM_MAXH; ++h) {
        if (h < nh) {
          const int mi = h - with_plain;
          const float m = mi < 0 ? 1.f : masks[(int64_t)(bp0 + j) * nm + mi];
          if (xv.x * m == mv[h].x) f[h][0] = j;
          if (xv.y * m == mv[h].y) f[h][1] = j;
          if (xv.z * m == mv[h].z) f[h][2] = j;
          if (xv.w * m == mv[h].w) f[h][3] = j;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < GGM_MAXH; ++h) {
      if (h < nh) {
        int* ap = arg + ((int64_t)h * B + b) * C + c4 * 4;
        const volatile int* cur = ap;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f[h][q] >= 0 && pix0 + f[h][q] < cur[q]) atomicMin(ap + q, pix0 + f[h][q]);
      }
    }
  }
}
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void gap_gmp_bwd_multi4_kernel(const float* __restrict__ masks, const int* __restrict__ arg,
                                                                 const float* __restrict__ g, void* __restrict__ dx, unsigned hw, int C,
                                                                 int nm, int with_plain, int B, int accumulate, unsigned total, FastDiv d_c4,
                                                                 FastDiv d_hw) {
  const float invP = 1.f / (float)hw;
  const int nh = nm + with_plain;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4;
    const unsigned bpu = fd_divmod((unsigned)t64, d_c4, c4);
    const unsigned b = fd_div(bpu, d_hw);
    const int pix = (int)(bpu - b * hw);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < GGM_MAXH; ++h) {
      if (h < nh) {
        const int mi = h - with_plain;
        const float m = mi < 0 ? 1.f : masks[(int64_t)bpu * nm + mi];
        const int64_t row = (int64_t)h * B + b;
        const int4 ar = *(const int4*)(arg + row * C + c4 * 4);
        const float4 ga = *(const float4*)(g + row * 2 * C + c4 * 4), gm = *(const float4*)(g + row * 2 * C + C + c4 * 4);
        o.x += m * (ga.x * invP + (pix == ar.x ? gm.x : 0.f));
        o.y += m * (ga.y * invP + (pix == ar.y ? gm.y : 0.f));
        o.z += m * (ga.z * invP + (pix == ar.z ? gm.z : 0.f));
        o.w += m * (ga.w * invP + (pix == ar.w ? gm.w : 0.f));
      }
    }
    const int64_t d = ((int64_t)bpu * (C >> 2) + c4) * 4;
    if (accumulate) { const float4 p = st_ld4<ST>(dx, d); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    st_st4<ST>(dx, d, o);
  }
}
// x [B][hw][ld], masks [B][hw][nm], v / g [(nm + with_plain) * B][2C] head-major, dx [B][hw][C] dense, arg_ws >= heads * B * C ints;
// st: storage type of x and dx
extern "C" int ppst_gap_gmp_multi_bwd(const void* x, const void* masks, const void* v, const void* g, void* dx, void* arg_ws, int B,
                                      int64_t hw, int C, int ld, int nm, int with_plain, int accumulate, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || hw <= 0 || hw > 0x7fffffffll || C <= 0 || C % 4 || ld % 4 || ld < C || nm < 1 || nm > 3 ||
      (with_plain != 0 && with_plain != 1) || hw % GMP_NP)
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !masks || !v || !g || !dx || !arg_ws) return PPST_ENULL;
  const int64_t total = (int64_t)B * hw * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  if ((((uintptr_t)v | (uintptr_t)g | (uintptr_t)arg_ws) % 16) != 0 || (((uintptr_t)x | (uintptr_t)dx) % (st ? 8 : 16)) != 0) return PPST_EINVAL;
  const int heads = nm + with_plain;
  hipError_t e = hipMemsetAsync(arg_ws, 0x7f, (size_t)heads * B * C * sizeof(int), as_stream(stream));
  if (e != hipSuccess) return (int)e;
  const int64_t t4 = total / 4, strips = t4 / GMP_NP;
  PPST_ST_SWITCH(st, PPST_LAUNCH(gmp_argmax_multi4_kernel<ST_>, dim3(tg_grid(strips)), dim3(256), 0, as_stream(stream), x, (const float*)masks,
                                 (const float*)v, (int*)arg_ws, (unsigned)hw, C, ld / 4, nm, with_plain, B, (unsigned)strips,
                                 make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)(hw / GMP_NP))));
  PPST_ST_SWITCH(st, PPST_LAUNCH(gap_gmp_bwd_multi4_kernel<ST_>, dim3(tg_grid(t4)), dim3(256), 0, as_stream(stream), (const float*)masks,
                                 (const int*)arg_ws, (const float*)g, dx, (unsigned)hw, C, nm, with_plain, B, accumulate, (unsigned)t4,
                                 make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)hw)));
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------ row-wise ops ----
// y = x * s, s = rsqrt(sum x^2 + eps) (mode 0, util.normalize) or 1 / max(||x||, eps) (mode 1, F.normalize):
// dx = s * (g - y * sum(g*y))   (mode 1 with ||x|| < eps: dx = g / eps)
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              float* __restrict__ dx, int K, float eps, int mode) {
  __shared__ float red[2][4];
  const float* xr = x + (int64_t)blockIdx.x * K;
  const float* gr = g + (int64_t)blockIdx.x * K;
  float ss = 0.f, gx = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) { ss += xr[k] * xr[k]; gx += gr[k] * xr[k]; }
  ss = wave_sum(ss); gx = wave_sum(gx);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ss; red[1][threadIdx.x >> 6] = gx; }
  __syncthreads();
  ss = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  gx = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  float s, proj;
  if (mode == 0) { s = rsqrtf(ss + eps); proj = s * s * s * gx; }
  else {
    const float nrm = sqrtf(ss);
    if (nrm < eps) { s = 1.f / eps; proj = 0.f; } else { s = 1.f / nrm; proj = s * s * s * gx; }
  }
  for (int k = threadIdx.x; k < K; k += 256) dx[(int64_t)blockIdx.x * K + k] = s * gr[k] - proj * xr[k];
}
extern "C" int ppst_l2norm_rows_bwd(const void* g, const void* x, void* dx, int B, int K, float eps, int mode, void* stream) {
  if (B < 0 || K <= 0 || mode < 0 || mode > 1) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!g || !x || !dx) return PPST_ENULL;
  PPST_LAUNCH(l2norm_rows_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), (const float*)g, (const float*)x, (float*)dx, K, eps, mode);
  return PPST_LAUNCH_CHECK();
}

// softmax(x / div) backward, in place on g: g <- p * (g - sum(g*p)) / div   (rows of the correspondence matrix)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ g, int cols, float inv_div) {
  __shared__ float red[4];
  const float* pr = p + (int64_t)blockIdx.x * cols;
  float* gr = g + (int64_t)blockIdx.x * cols;
  float d = 0.f;
  for (int k = threadIdx.x; k < cols; k += 256) d += pr[k] * gr[k];
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  d = red[0] + red[1] + red[2] + red[3];
  for (int k = threadIdx.x; k < cols; k += 256) gr[k] = pr[k] * (gr[k] - d) * inv_div;
}
extern "C" int ppst_softmax_rows_bwd(const void* p, void* g, int64_t rows, int cols, float div, void* stream) {
  if (rows < 0 || cols <= 0 || div == 0.f || rows > 0x7fffffffll) return PPST_EINVAL;
  if (rows == 0) return PPST_OK;
  if (!p || !g) return PPST_ENULL;
  PPST_LAUNCH(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (const float*)p, (float*)g, cols, 1.f / div);
  return PPST_LAUNCH_CHECK();
}

// correspondence feature preparation backward (ppst_model.py:343-356): per row (pixel) x (C values)
//   z = x with the first `ncenter` channels mean-centred over those channels;  y = z / (||z||_2 + eps)
// dz = (g - y * sum(g*y)) / (||z|| + eps);  dx = dz with the first ncenter entries re-centred.
__global__ __launch_bounds__(256) void corr_prep_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            float* __restrict__ dx, int64_t rows, int C, int ncenter, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  const float* gr = g + row * C;
  float m = 0.f;
  for (int k = lane; k < ncenter; k += 64) m += xr[k];
  m = ncenter > 0 ? wave_sum(m) / (float)ncenter : 0.f;
  float ss = 0.f, gz = 0.f;
  for (int k = lane; k < C; k += 64) {
    const float z = xr[k] - (k < ncenter ? m : 0.f);
    ss += z * z;
    gz += gr[k] * z;
  }
  ss = wave_sum(ss); gz = wave_sum(gz);
  const float nrm = sqrtf(ss), inv = 1.f / (nrm + eps);
  // y = z*inv;  d||z||/dz = z/||z||  ->  dz = inv*g - inv^2 * (g.z) * z/||z||
  const float k2 = nrm > 0.f ? inv * inv * gz / nrm : 0.f;
  float dsum = 0.f;
  for (int k = lane; k < ncenter; k += 64) dsum += inv * gr[k] - k2 * (xr[k] - m);
  dsum = ncenter > 0 ? wave_sum(dsum) / (float)ncenter : 0.f;
  for (int k = lane; k < C; k += 64) {
    const float z = xr[k] - (k < ncenter ? m : 0.f);
    dx[row * C + k] = inv * gr[k] - k2 * z - (k < ncenter ? dsum : 0.f);
  }
}
extern "C" int ppst_corr_prep_bwd(const void* g, const void* x, void* dx, int64_t rows, int C, int ncenter, float eps, void* stream) {
  if (rows < 0 || C <= 0 || ncenter < 0 || ncenter > C) return PPST_EINVAL;
  if (rows == 0) return PPST_OK;
  if (!g || !x || !dx) return PPST_ENULL;
  PPST_LAUNCH(corr_prep_bwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, as_stream(stream), (const float*)g, (const float*)x,
              (float*)dx, rows, C, ncenter, eps);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------------- losses ---
// d(weight * mean|a - b|)/da = weight / n * sign(a - b)   (torch: sign(0) = 0)
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                                      int64_t n, float w) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    da[i] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
  }
}
extern "C" int ppst_l1_grad(const void* a, const void* b, void* da, int64_t n, float weight, void* stream) {
  if (n < 0) return PPST_EINVAL;
  if (n == 0) return PPST_OK;
  if (!a || !b || !da) return PPST_ENULL;
  PPST_LAUNCH(l1_grad_kernel, dim3(tg_grid(n)), dim3(256), 0, as_stream(stream), (const float*)a, (const float*)b, (float*)da, n,
              weight / (float)n);
  return PPST_LAUNCH_CHECK();
}

// NoiseInjection weight gradient (stylegan2_layers.py:376-399): d/dw sum over (b,p,c) of dpre[b][p][c] * noise[b][p]
__global__ __launch_bounds__(256) void noise_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ noise,
                                                          float* __restrict__ partial, int64_t npix, int C, int ld) {
  __shared__ float red[4];
  float acc = 0.f;
  // one wave per pixel at a time: lanes stride the channels (coalesced), then the pixel's noise value multiplies the row sum
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < npix; p += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += dpre[p * ld + c];
    acc += s * noise[p];
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// float4 form (C, ld multiples of 4, aligned): a wave covers 256 / C4 ... whole pixels per pass -- lane -> (pixel, channel quad)
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void noise_wgrad4_kernel(const void* __restrict__ dpre, const float* __restrict__ noise,
                                                           float* __restrict__ partial, int64_t npix, int C4, int ld4) {
  __shared__ float red[4];
  float acc = 0.f;
  const int lanes = C4 < 256 ? C4 : 256;                     // threads per pixel
  const int ppb = 256 / lanes;                                // pixels per block pass
  const int cl = threadIdx.x % lanes, pr = threadIdx.x / lanes;
  if (pr < ppb)
    for (int64_t p = (int64_t)blockIdx.x * ppb + pr; p < npix; p += (int64_t)gridDim.x * ppb) {
      float s = 0.f;
      for (int c = cl; c < C4; c += lanes) { const float4 v = st_ld4<ST>(dpre, (p * ld4 + c) * 4); s += (v.x + v.y) + (v.z + v.w); }
      acc += s * noise[p];
    }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, float scale,
                                                           int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) * scale;
    out[0] = accumulate ? out[0] + v : v;
  }
}
extern "C" int64_t ppst_noise_wgrad_ws(int64_t npix) {
  int64_t b = cdiv64(npix, 4);
  if (b > 2048) b = 2048;
  return (b < 1 ? 1 : b) * (int64_t)sizeof(float);
}
extern "C" int ppst_noise_wgrad_st(const void* dpre, const void* noise, void* out, void* ws, int64_t npix, int C, int ld, int accumulate,
                                   int st, void* stream);
extern "C" int ppst_noise_wgrad(const void* dpre, const void* noise, void* out, void* ws, int64_t npix, int C, int ld, int accumulate,
                                void* stream) {
  return ppst_noise_wgrad_st(dpre, noise, out, ws, npix, C, ld, accumulate, PPST_ST_F32, stream);
}
extern "C" int ppst_noise_wgrad_st(const void* dpre, const void* noise, void* out, void* ws, int64_t npix, int C, int ld, int accumulate,
                                   int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (npix < 0 || C <= 0 || ld < C) return PPST_EINVAL;
  if (!out) return PPST_ENULL;
  if (npix == 0) return accumulate ? PPST_OK : (int)hipMemsetAsync(out, 0, sizeof(float), as_stream(stream));
  if (!dpre || !noise || !ws) return PPST_ENULL;
  const int blocks = (int)(ppst_noise_wgrad_ws(npix) / (int64_t)sizeof(float));
  // (C / 4 not a divisor of 256, e.g. 96: the last 256 % (C / 4) threads of a block idle -- the fp32 path keeps its scalar form for
  //  those widths, bit for bit as before; a half tensor takes the four-channel form at every width)
  if (C % 4 == 0 && ld % 4 == 0 && (uintptr_t)dpre % (st ? 8 : 16) == 0 && (st || C / 4 >= 256 || 256 % (C / 4) == 0))
    PPST_ST_SWITCH(st, PPST_LAUNCH(noise_wgrad4_kernel<ST_>, dim3(blocks), dim3(256), 0, as_stream(stream), dpre, (const float*)noise,
                                   (float*)ws, npix, C / 4, ld / 4));
  else if (st)
    return PPST_EINVAL;
  else
    PPST_LAUNCH(noise_wgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)dpre, (const float*)noise, (float*)ws,
                npix, C, ld);
  PPST_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)ws, (float*)out, blocks, 1.f, accumulate);
  return PPST_LAUNCH_CHECK();
}
// sum of a small float vector times scale -> out[0] (PReLU slope gradient from ppst_prelu_bwd's partials)
extern "C" int ppst_sum_partials(const void* partial, void* out, int n, float scale, void* stream) {
  if (n <= 0) return PPST_EINVAL;
  if (!partial || !out) return PPST_ENULL;
  PPST_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)partial, (float*)out, n, scale, 0);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------- weight-space adjoint --
// adjoint of ppst_upscale_weight: w4[c][n][ky][kx] = scale * (w[n][c][ky][kx] + w[ky-1][kx] + w[ky][kx-1] + w[ky-1][kx-1])
// -> dw[n][c][y][x] = scale * (dw4[c][n][y][x] + dw4[y+1][x] + dw4[y][x+1] + dw4[y+1][x+1])
__global__ __launch_bounds__(256) void upscale_weight_bwd_kernel(const float* __restrict__ dw4, float* __restrict__ dw, int cout, int cin,
                                                                 float scale, int64_t total, int accumulate) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xk = (int)(t % 3), yk = (int)((t / 3) % 3);
    const int64_t r = t / 9;
    const int c = (int)(r % cin), n = (int)(r / cin);
    const float* p = dw4 + ((int64_t)c * cout + n) * 16;
    const float v = scale * (p[yk * 4 + xk] + p[(yk + 1) * 4 + xk] + p[yk * 4 + xk + 1] + p[(yk + 1) * 4 + xk + 1]);
    dw[t] = accumulate ? dw[t] + v : v;
  }
}
extern "C" int ppst_upscale_weight_bwd(const void* dw4, void* dw, int cout, int cin, float scale, int accumulate, void* stream) {
  if (cout <= 0 || cin <= 0) return PPST_EINVAL;
  if (!dw4 || !dw) return PPST_ENULL;
  const int64_t total = (int64_t)cout * cin * 9;
  PPST_LAUNCH(upscale_weight_bwd_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)dw4, (float*)dw, cout, cin,
              scale, total, accumulate);
  return PPST_LAUNCH_CHECK();
}

// space-to-depth copy: x [B][H][W][C] -> y [B][ceil(H/2)][ceil(W/2)][4C], channel block (py*2+px)*C (zeros beyond the edge):
// the layout the stride-2 step tables read (input gradient of the fused transposed conv = a stride-2 4x4 conv of dY)
template <typename VT>
__global__ __launch_bounds__(256) void s2d_kernel(const VT* __restrict__ x, VT* __restrict__ y, int H, int W, int C, int x_ld,
                                                  int H2, int W2, unsigned total, FastDiv d_c, FastDiv d_4, FastDiv d_w2, FastDiv d_h2) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, ph, qx, qy;
    unsigned r = fd_divmod((unsigned)t64, d_c, c);
    r = fd_divmod(r, d_4, ph);
    r = fd_divmod(r, d_w2, qx);
    const unsigned b = fd_divmod(r, d_h2, qy);
    const int iy = (int)qy * 2 + (int)(ph >> 1), ix = (int)qx * 2 + (int)(ph & 1);
    VT v = {};
    if (iy < H && ix < W) v = x[(((int64_t)b * H + iy) * W + ix) * x_ld + c];
    y[t64] = v;
  }
}
extern "C" int ppst_space_to_depth_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int st, void* stream);
extern "C" int ppst_space_to_depth(const void* x, void* y, int B, int H, int W, int C, int x_ld, void* stream) {
  return ppst_space_to_depth_st(x, y, B, H, W, C, x_ld, PPST_ST_F32, stream);
}
extern "C" int ppst_space_to_depth_st(const void* x, void* y, int B, int H, int W, int C, int x_ld, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || x_ld < C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const int64_t total = (int64_t)B * H2 * W2 * 4 * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  if (st) {       // half storage: four channels = one 8-byte element (pure data movement)
    if (C % 4 || x_ld % 4 || ((uintptr_t)x | (uintptr_t)y) % 8) return PPST_EINVAL;
    PPST_LAUNCH(s2d_kernel<uint2>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), (const uint2*)x, (uint2*)y, H, W, C / 4,
                x_ld / 4, H2, W2, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv(4u), make_fastdiv((unsigned)W2),
                make_fastdiv((unsigned)H2));
    return PPST_LAUNCH_CHECK();
  }
  if (C % 4 == 0 && x_ld % 4 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0)
    PPST_LAUNCH(s2d_kernel<float4>, dim3(tg_grid(total / 4)), dim3(256), 0, as_stream(stream), (const float4*)x, (float4*)y, H, W, C / 4,
                x_ld / 4, H2, W2, (unsigned)(total / 4), make_fastdiv((unsigned)(C / 4)), make_fastdiv(4u), make_fastdiv((unsigned)W2),
                make_fastdiv((unsigned)H2));
  else
    PPST_LAUNCH(s2d_kernel<float>, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, H, W, C, x_ld, H2, W2,
                (unsigned)total, make_fastdiv((unsigned)C), make_fastdiv(4u), make_fastdiv((unsigned)W2), make_fastdiv((unsigned)H2));
  return PPST_LAUNCH_CHECK();
}

// y = x * s[0] (s on the device): chain rule through a scalar loss without a host round trip
__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y,
                                                       int64_t n) {
  const float f = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * f;
}
extern "C" int ppst_scale_by(const void* x, const void* s, void* y, int64_t n, void* stream) {
  if (n < 0) return PPST_EINVAL;
  if (n == 0) return PPST_OK;
  if (!x || !s || !y) return PPST_ENULL;
  PPST_LAUNCH(scale_by_kernel, dim3(tg_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)s, (float*)y, n);
  return PPST_LAUNCH_CHECK();
}

// --------------------------------------------------------- rsclLoss backward --
// d(mean_i CE(logits_i, 0))/dq (networks/rscl.py:42-64; keys, queue detached; the current-batch logits are the constant
// -10 of the reference's eye(1) mask): dq_i = g/(n*T) * [ (p_pos - 1) k_i + sum_j p_j key_j ],  key_j = queue[:, j] | k0_j.
#include "rscl_common.h"
__global__ __launch_bounds__(RS_T) void rscl_rows_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ k0, const float* __restrict__ queue,
                                                             const float* __restrict__ gout, float* __restrict__ dq, int n, int n0,
                                                             int C, int K, float invT) {
  __shared__ RsclShared sh;
  const int i = blockIdx.x, t = threadIdx.x;
  float s_pos, m, ssum;
  rscl_logits(q + (int64_t)i * C, k + (int64_t)i * C, k0, queue, n, n0, C, K, invT, sh, s_pos, m, ssum);
  const float inv_sum = 1.f / ssum;
  const float ppos = expf(s_pos - m) * inv_sum;
  const float f = gout[0] * invT / (float)n;
  for (int c = t; c < C; c += RS_T) {           // thread = channel: its queue row is contiguous (K floats)
    const float* qr = queue + (int64_t)c * K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 3 < K; j += 4) {
      a0 += sh.prob[j] * qr[j]; a1 += sh.prob[j + 1] * qr[j + 1]; a2 += sh.prob[j + 2] * qr[j + 2]; a3 += sh.prob[j + 3] * qr[j + 3];
    }
    for (; j < K; ++j) a0 += sh.prob[j] * qr[j];
    for (int j2 = 0; j2 < n0; ++j2) a1 += sh.prob[K + j2] * k0[(int64_t)j2 * C + c];
    dq[(int64_t)i * C + c] = f * ((ppos - 1.f) * k[(int64_t)i * C + c] + ((a0 + a1) + (a2 + a3)) * inv_sum);
  }
}
extern "C" int ppst_rscl_loss_bwd(const void* q, const void* k, const void* k0, const void* queue, const void* gout, void* dq, int n,
                                  int n0, int C, int K, float nce_T, void* stream) {
  if (n <= 0 || n > 64 || n0 < 0 || C <= 0 || K <= 0 || K + n0 > 512 || nce_T <= 0.f) return PPST_EINVAL;
  if (!q || !k || !queue || !gout || !dq || (n0 > 0 && !k0)) return PPST_ENULL;
  PPST_LAUNCH(rscl_rows_bwd_kernel, dim3(n), dim3(RS_T), 0, as_stream(stream), (const float*)q, (const float*)k, (const float*)k0,
              (const float*)queue, (const float*)gout, (float*)dq, n, n0, C, K, 1.0f / nce_T);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------ Rselfcorr backward --
// forward (ppst_model.py:330-339, ppst_rselfcorr): per 4x4 patch X[c][i] (c < 64 channels, i < 16 positions):
//   d[:, i] = X[:, i] - mean_c;  z[:, i] = d[:, i] / (||d[:, i]|| + eps);  G[i][j] = sum_c z[c][i] z[c][j]  (out channel i*16+j).
// backward: dz[c][i] = sum_j (dG[i][j] + dG[j][i]) z[c][j];  dd = inv*dz - (inv^2/nrm) (dz . d) d;  dX = dd - mean_c(dd).
// One wave per patch, lane = channel.
__global__ __launch_bounds__(256) void rselfcorr_bwd_kernel(const float* __restrict__ fea, const float* __restrict__ dout,
                                                            float* __restrict__ dfea, int B, int H, int W, int dout_ld, float eps,
                                                            int64_t npatch) {
  __shared__ float S[4][16][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gy = H >> 2, gx = W >> 2;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wv; p < npatch; p += (int64_t)gridDim.x * 4) {
    const int px = (int)(p % gx);
    const int64_t r = p / gx;
    const int py = (int)(r % gy), b = (int)(r / gy);
    const float* go = dout + (((int64_t)b * gy + py) * gx + px) * dout_ld;
    // S = dG + dG^T
    for (int e = lane; e < 256; e += 64) {
      const int i = e >> 4, j = e & 15;
      S[wv][i][j] = go[i * 16 + j] + go[j * 16 + i];
    }
    float d[16], z[16], inv[16], nrm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int iy = py * 4 + (i >> 2), ix = px * 4 + (i & 3);
      const float v = fea[(((int64_t)b * H + iy) * W + ix) * 64 + lane];
      const float m = wave_sum(v) * (1.f / 64.f);
      d[i] = v - m;
      nrm[i] = sqrtf(wave_sum(d[i] * d[i]));
      inv[i] = 1.f / (nrm[i] + eps);
      z[i] = d[i] * inv[i];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float dz = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) dz += S[wv][i][j] * z[j];
      const float dot = wave_sum(dz * d[i]);
      float dd = inv[i] * dz - (nrm[i] > 0.f ? inv[i] * inv[i] / nrm[i] * dot * d[i] : 0.f);
      dd -= wave_sum(dd) * (1.f / 64.f);
      const int iy = py * 4 + (i >> 2), ix = px * 4 + (i & 3);
      dfea[(((int64_t)b * H + iy) * W + ix) * 64 + lane] = dd;
    }
    __builtin_amdgcn_wave_barrier();
  }
}
extern "C" int ppst_rselfcorr_bwd(const void* fea, const void* dout, void* dfea, int B, int H, int W, int C, int dout_ld, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || H % 4 || W % 4 || C != 64 || dout_ld < 256) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!fea || !dout || !dfea) return PPST_ENULL;
  const int64_t npatch = (int64_t)B * (H / 4) * (W / 4);
  int64_t blocks = cdiv64(npatch, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  PPST_LAUNCH(rselfcorr_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)fea, (const float*)dout,
              (float*)dfea, B, H, W, dout_ld, 2.220446049250313e-16f, npatch);
  return PPST_LAUNCH_CHECK();
}

// ---- depth to space (round 5): x [B][th][tw][4 C] (channel block g = py*2+px holds output phase (py, px)) -> y [B][oh][ow][C],
// y[b][2q+py][2p+px][c] = x[b][q][p][g*C + c], oh <= 2 th, ow <= 2 tw (an odd extent drops the last phase-1 row / column).  Pure data
// movement: 16-byte items of either storage type (C % 4 == 0 for fp32, % 8 for half tensors).
__global__ __launch_bounds__(256) void depth_to_space_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int th, int tw, int oh, int ow,
                                                             int cq, unsigned total, FastDiv d_cq, FastDiv d_ow, FastDiv d_oh) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c, ox, oy;
    unsigned r = fd_divmod((unsigned)t64, d_cq, c);
    r = fd_divmod(r, d_ow, ox);
    const unsigned b = fd_divmod(r, d_oh, oy);
    const unsigned g = (oy & 1) * 2 + (ox & 1);
    y[t64] = x[(((uint64_t)b * th + (oy >> 1)) * tw + (ox >> 1)) * (4u * cq) + g * cq + c];
  }
}
extern "C" int ppst_depth_to_space_st(const void* x, void* y, int B, int th, int tw, int oh, int ow, int C, int st, void* stream) {
  if ((unsigned)st > 2u) return PPST_EINVAL;
  const int per = st ? 8 : 4;                     // elements per 16-byte item
  if (B < 0 || th <= 0 || tw <= 0 || oh <= 0 || ow <= 0 || oh > 2 * th || ow > 2 * tw || C <= 0 || C % per) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  if (((uintptr_t)x | (uintptr_t)y) % 16) return PPST_EINVAL;
  const int cq = C / per;
  const int64_t total = (int64_t)B * oh * ow * cq;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  PPST_LAUNCH(depth_to_space_kernel, dim3(tg_grid(total)), dim3(256), 0, as_stream(stream), (const uint4*)x, (uint4*)y, th, tw, oh, ow, cq,
              (unsigned)total, make_fastdiv(cq), make_fastdiv(ow), make_fastdiv(oh));
  return PPST_LAUNCH_CHECK();
}
