// Small-batch linear layers (weight-read bound GEMV batches) and row normalisation.
// y[b][n] = act( sum_k f(x[b][k]) * w[n][k] * wscale + bias[n] * bscale )
// Covers EqualLinear (stylegan2_layers.py:222-242), EqualizedLinear/StyleMod (:268-273,
// :364-374), GeneratorModulation (generator.py:80-91) and the nn.Linear chain of the E2
// projectors (encoder_col.py:52-88).  One wave per output row n streams the weight row
// once with 16-B loads (algorithmic bytes = 4*N*K, x stays L1/L2 resident) and keeps one
// accumulator per batch row; xor-shuffle reduction at the end.
#include "common.h"

#define LIN_BMAX 16

template <int BT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int B, int K,
                                                     int N, float wscale, float bscale, int relu_in, int act, int b0) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = 0.f;
  const float* wr = w + (int64_t)n * K;
  if ((K & 3) == 0) {
    for (int k = lane * 4; k < K; k += 256) {
      float4 ww = *(const float4*)(wr + k);
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b0 + b < B) {
          float4 xv = *(const float4*)(x + (int64_t)(b0 + b) * K + k);
          if (relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
          acc[b] += xv.x * ww.x + xv.y * ww.y + xv.z * ww.z + xv.w * ww.w;
        }
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      float ww = wr[k];
#pragma unroll
      for (int b = 0; b < BT; ++b)
        if (b0 + b < B) {
          float xv = x[(int64_t)(b0 + b) * K + k];
          if (relu_in) xv = fmaxf(xv, 0.f);
          acc[b] += xv * ww;
        }
    }
  }
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = wave_sum(acc[b]);
  if (lane == 0) {
    float bb = bias ? bias[n] * bscale : 0.f;
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b0 + b < B) {
        float v = acc[b] * wscale + bb;
        if (act == PPST_ACT_LRELU) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
        y[(int64_t)(b0 + b) * N + n] = v;
      }
  }
}

// Same contract, long rows (K >= 1024, K % 4 == 0): the 4 waves of a block share ONE output row
// (each streams a quarter of it), so a 512-row layer runs 512 blocks instead of 128 -- the
// one-wave-per-row form leaves half the CUs idle on the StyleMod / projector GEMVs.
template <int BT>
__global__ __launch_bounds__(256) void linear_ksplit_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B, int K,
                                                            int N, float wscale, float bscale, int relu_in, int act, int b0) {
  __shared__ float sm[4][BT];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = 0.f;
  const float* wr = w + (int64_t)n * K;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    float4 ww = *(const float4*)(wr + k);
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b0 + b < B) {
        float4 xv = *(const float4*)(x + (int64_t)(b0 + b) * K + k);
        if (relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
        acc[b] += xv.x * ww.x + xv.y * ww.y + xv.z * ww.z + xv.w * ww.w;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b] = wave_sum(acc[b]);
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < BT; ++b) sm[wv][b] = acc[b];
  }
  __syncthreads();
  if (threadIdx.x < BT && b0 + (int)threadIdx.x < B) {
    const int b = threadIdx.x;
    float v = ((sm[0][b] + sm[1][b]) + (sm[2][b] + sm[3][b])) * wscale + (bias ? bias[n] * bscale : 0.f);
    if (act == PPST_ACT_LRELU) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
    y[(int64_t)(b0 + b) * N + n] = v;
  }
}

extern "C" int ppst_linear(const void* x, const void* w, const void* bias, void* y, int B, int K, int N, float wscale,
                           float bscale, int relu_in, int act, void* stream) {
  if (B < 0 || K <= 0 || N <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !w || !y) return PPST_ENULL;
  for (int b0 = 0; b0 < B; b0 += LIN_BMAX) {
    int nb = B - b0 < LIN_BMAX ? B - b0 : LIN_BMAX;
    const bool ksplit = K >= 1024 && (K & 3) == 0 && N <= 8192;
    dim3 grid(ksplit ? N : cdiv(N, 4));
#define LAUNCH(BT)                                                                                                          \
  do {                                                                                                                      \
    if (ksplit)                                                                                                             \
      PPST_LAUNCH(linear_ksplit_kernel<BT>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)w,       \
                  (const float*)bias, (float*)y, B, K, N, wscale, bscale, relu_in, act, b0);                                \
    else                                                                                                                    \
      PPST_LAUNCH(linear_kernel<BT>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)w,              \
                  (const float*)bias, (float*)y, B, K, N, wscale, bscale, relu_in, act, b0);                                \
  } while (0)
    if (nb <= 1) LAUNCH(1);
    else if (nb <= 2) LAUNCH(2);
    else if (nb <= 4) LAUNCH(4);
    else if (nb <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    int e = PPST_LAUNCH_CHECK();
    if (e) return e;
  }
  return PPST_OK;
}

// mode 0: y = x * rsqrt(sum x^2 + eps)   (util.normalize, util/util.py:18-22)
// mode 1: y = x / max(sqrt(sum x^2), eps) (F.normalize, encoder_col.py:168)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int K, float eps, int mode) {
  __shared__ float sm[4];
  const float* xr = x + (int64_t)blockIdx.x * K;
  float* yr = y + (int64_t)blockIdx.x * K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) { float v = xr[k]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  s = sm[0] + sm[1] + sm[2] + sm[3];
  float f = mode == 0 ? rsqrtf(s + eps) : 1.f / fmaxf(sqrtf(s), eps);
  for (int k = threadIdx.x; k < K; k += 256) yr[k] = xr[k] * f;
}
extern "C" int ppst_l2norm_rows(const void* x, void* y, int B, int K, float eps, int mode, void* stream) {
  if (B < 0 || K <= 0 || mode < 0 || mode > 1) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  PPST_LAUNCH(l2norm_rows_kernel, dim3(B), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, K, eps, mode);
  return PPST_LAUNCH_CHECK();
}
