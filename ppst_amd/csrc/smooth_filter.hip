// Local-affine photo smoothing: /root/reference/smooth_filter.py -- the three NVRTC kernels
// best_local_affine_kernel (:149-241), bilateral_smooth_kernel (:243-290), reconstruction_best_kernel (:293-321) and their
// driver smooth_local_affine (:332-378) -- as two launches:
//
//   local_affine_kernel          one thread per pixel: normal equations of the 3x4 affine map content patch -> stylised
//                                patch (float products accumulated in double, +1e-3 on the colour diagonal), 4x4 inverse
//                                by cofactors in double, model[pixel][12] fp32.   HBM: 24 B in (L2-served 3x3 window), 48 B out.
//   bilateral_reconstruct_kernel 16x16-pixel tile per block; the (16 + 2r)^2 window of the model (48 B / px) and of the
//                                guide (padded to 16 B / px) is staged ONCE in LDS (135 KB at r = 15) instead of being
//                                re-read (2r+1)^2 = 961 times per pixel from global memory as the reference kernel does;
//                                spatial weights from a 961-entry LDS table; double accumulation; the smoothed model is
//                                rounded to fp32 (the reference stores it in a float array) and applied to the content pixel
//                                in the same kernel -- the filtered model never travels through HBM unless asked for.
//                                HBM: 48 + 12 B in (x halo 2.1 at r = 15), 12 B out per pixel; bound by the fp64 adds
//                                (13 per tap), not by memory.
//
// Planar fp32 images [B][3][H][W] as the reference passes them (channel order as given: the reference feeds BGR and the
// kernels' channel-reversed indexing returns RGB; that indexing is kept).  Parity with the CUDA original is unpinned
// (cupy / pynvrtc / CUDA absent; oracle/smooth_filter_oracle.py header).
#include "common.h"

__device__ __forceinline__ bool inverse4x4(const double a[4][4], double inv[4][4]) {
  const double s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1], s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
  const double s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3], s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
  const double s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3], s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
  const double c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3], c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
  const double c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2], c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
  const double c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2], c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
  double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  if (fabs(det) < 1e-9) return false;        // smooth_filter.py:134-136: the caller keeps an all-zero inverse
  det = 1.0 / det;
  inv[0][0] = (a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * det;
  inv[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * det;
  inv[0][2] = (a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * det;
  inv[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * det;
  inv[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * det;
  inv[1][1] = (a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * det;
  inv[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * det;
  inv[1][3] = (a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * det;
  inv[2][0] = (a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * det;
  inv[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * det;
  inv[2][2] = (a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * det;
  inv[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * det;
  inv[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * det;
  inv[3][1] = (a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * det;
  inv[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * det;
  inv[3][3] = (a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * det;
  return true;
}

// smooth_filter.py:149-241.  f = (I[2], I[1], I[0], 1); row i of the model is fitted to output channel 2 - i.
__global__ __launch_bounds__(256) void local_affine_kernel(const float* __restrict__ output, const float* __restrict__ input,
                                                           float* __restrict__ model, int H, int W, int radius) {
  const int size = H * W;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= size) return;
  const float* in = input + (int64_t)blockIdx.y * 3 * size;
  const float* out = output + (int64_t)blockIdx.y * 3 * size;
  const int x = id % W, y = id / W;
  double M[4][4], S[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      M[i][j] = (i == j && i != 3) ? 1e-3 : 0.0;
      if (i != 3) S[i][j] = 0.0;
    }
  for (int dy = -radius; dy <= radius; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -radius; dx <= radius; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const int id2 = yy * W + xx;
      const float f[3] = {in[id2 + 2 * size], in[id2 + size], in[id2]};
      const float t[3] = {out[id2 + 2 * size], out[id2 + size], out[id2]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) M[a][b] += (double)(f[a] * f[b]);     // float product, double sum (as the kernel's `+=`)
        M[a][3] += (double)f[a];
        M[3][a] += (double)f[a];
#pragma unroll
        for (int i = 0; i < 3; ++i) S[i][a] += (double)(f[a] * t[i]);
      }
      M[3][3] += 1.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) S[i][3] += (double)t[i];
    }
  }
  double inv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[i][j] = 0.0;
  (void)inverse4x4(M, inv);
  float* mo = model + ((int64_t)blockIdx.y * size + id) * 12;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += inv[j][k] * S[i][k];
      r[j] = (float)a;
    }
    *(float4*)(mo + 4 * i) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// smooth_filter.py:243-290 + :293-321.  LDS_TILE: window staged in LDS (radius <= SF_RMAX); otherwise read from global.
#define SF_T 16
#define SF_RMAX 15
#define SF_TS (SF_T + 2 * SF_RMAX)
template <bool LDS_TILE>
__global__ __launch_bounds__(256) void bilateral_reconstruct_kernel(const float* __restrict__ model, const float* __restrict__ guide,
                                                                    float* __restrict__ result, float* __restrict__ filtered,
                                                                    int H, int W, int radius, float sigma1, float sigma2) {
  __shared__ float4 s_model[LDS_TILE ? SF_TS * SF_TS * 3 : 1];
  __shared__ float4 s_guide[LDS_TILE ? SF_TS * SF_TS : 1];
  __shared__ float s_w[LDS_TILE ? (2 * SF_RMAX + 1) * (2 * SF_RMAX + 1) : 1];
  const int size = H * W;
  const int b = blockIdx.z;
  const float* g = guide + (int64_t)b * 3 * size;
  const float4* m4 = (const float4*)(model + (int64_t)b * size * 12);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * SF_T, y0 = blockIdx.y * SF_T;
  const int x = x0 + tx, y = y0 + ty;
  const int ts = SF_T + 2 * radius, win = 2 * radius + 1;
  const float den1 = 2.f * sigma1 * sigma1, den2 = 2.f * sigma2 * sigma2;
  if (LDS_TILE) {
    for (int i = threadIdx.x; i < ts * ts; i += 256) {
      const int ly = i / ts, lx = i - ly * ts;
      const int gy = y0 - radius + ly, gx = x0 - radius + lx;
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), a0 = gv, a1 = gv, a2 = gv;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int p = gy * W + gx;
        gv = make_float4(g[p], g[p + size], g[p + 2 * size], 0.f);
        a0 = m4[(int64_t)p * 3]; a1 = m4[(int64_t)p * 3 + 1]; a2 = m4[(int64_t)p * 3 + 2];
      }
      s_guide[i] = gv;
      s_model[i * 3] = a0; s_model[i * 3 + 1] = a1; s_model[i * 3 + 2] = a2;
    }
    for (int i = threadIdx.x; i < win * win; i += 256) {
      const int dx = i / win - radius, dy = i % win - radius;          // dx outer, dy inner (the reference's loop order)
      s_w[i] = expf((float)(-(dx * dx + dy * dy)) / den1);
    }
    __syncthreads();
  }
  if (x >= W || y >= H) return;
  const int id = y * W + x;
  const float gc0 = g[id], gc1 = g[id + size], gc2 = g[id + 2 * size];
  double sum[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) sum[k] = 0.0;
  double sum_w = 0.0;
  for (int dx = -radius; dx <= radius; ++dx) {
    const int xx = x + dx;
    if (xx < 0 || xx >= W) continue;
    for (int dy = -radius; dy <= radius; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      float4 gv, a0, a1, a2;
      float v1;
      if (LDS_TILE) {
        const int li = (ty + dy + radius) * ts + tx + dx + radius;
        gv = s_guide[li];
        a0 = s_model[li * 3]; a1 = s_model[li * 3 + 1]; a2 = s_model[li * 3 + 2];
        v1 = s_w[(dx + radius) * win + dy + radius];
      } else {
        const int p = yy * W + xx;
        gv = make_float4(g[p], g[p + size], g[p + 2 * size], 0.f);
        a0 = m4[(int64_t)p * 3]; a1 = m4[(int64_t)p * 3 + 1]; a2 = m4[(int64_t)p * 3 + 2];
        v1 = expf((float)(-(dx * dx + dy * dy)) / den1);
      }
      const float d0 = gv.x - gc0, d1 = gv.y - gc1, d2 = gv.z - gc2;
      const float cds = (d0 * d0 + d1 * d1 + d2 * d2) / 3.f;
      const float wgt = v1 * expf(-cds / den2);
      sum[0] += (double)(wgt * a0.x); sum[1] += (double)(wgt * a0.y); sum[2] += (double)(wgt * a0.z); sum[3] += (double)(wgt * a0.w);
      sum[4] += (double)(wgt * a1.x); sum[5] += (double)(wgt * a1.y); sum[6] += (double)(wgt * a1.z); sum[7] += (double)(wgt * a1.w);
      sum[8] += (double)(wgt * a2.x); sum[9] += (double)(wgt * a2.y); sum[10] += (double)(wgt * a2.z); sum[11] += (double)(wgt * a2.w);
      sum_w += (double)wgt;
    }
  }
  float fm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) fm[k] = (float)(sum[k] / sum_w);
  if (filtered) {
    float4* fo = (float4*)(filtered + ((int64_t)b * size + id) * 12);
    fo[0] = make_float4(fm[0], fm[1], fm[2], fm[3]);
    fo[1] = make_float4(fm[4], fm[5], fm[6], fm[7]);
    fo[2] = make_float4(fm[8], fm[9], fm[10], fm[11]);
  }
  float* r = result + (int64_t)b * 3 * size;
#pragma unroll
  for (int c = 0; c < 3; ++c)   // I[2]*A[c][0] + I[1]*A[c][1] + I[0]*A[c][2] + A[c][3], float, contracted left to right
    r[id + c * size] = fmaf(gc0, fm[4 * c + 2], fmaf(gc1, fm[4 * c + 1], gc2 * fm[4 * c])) + fm[4 * c + 3];
}

extern "C" int64_t ppst_smooth_local_affine_ws(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)B * H * W * 12 * (int64_t)sizeof(float);
}

extern "C" int ppst_smooth_local_affine(const void* output, const void* input, void* result, void* model_ws, void* filtered_model,
                                        int B, int H, int W, int patch_radius, int filter_radius, float sigma1, float sigma2,
                                        void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || patch_radius < 0 || filter_radius < 0 || !(sigma1 > 0.f) || !(sigma2 > 0.f) ||
      (int64_t)H * W * 12 > 0x7fffffffll || B > 65535)
    return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!output || !input || !result || !model_ws) return PPST_ENULL;
  if (((uintptr_t)model_ws | (uintptr_t)filtered_model) % 16) return PPST_EINVAL;
  hipStream_t st = as_stream(stream);
  PPST_LAUNCH(local_affine_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, st, (const float*)output, (const float*)input,
              (float*)model_ws, H, W, patch_radius);
  int e = PPST_LAUNCH_CHECK();
  if (e) return e;
  dim3 grid(cdiv(W, SF_T), cdiv(H, SF_T), B);
  if (filter_radius <= SF_RMAX)
    PPST_LAUNCH(bilateral_reconstruct_kernel<true>, grid, dim3(256), 0, st, (const float*)model_ws, (const float*)input,
                (float*)result, (float*)filtered_model, H, W, filter_radius, sigma1, sigma2);
  else
    PPST_LAUNCH(bilateral_reconstruct_kernel<false>, grid, dim3(256), 0, st, (const float*)model_ws, (const float*)input,
                (float*)result, (float*)filtered_model, H, W, filter_radius, sigma1, sigma2);
  return PPST_LAUNCH_CHECK();
}
