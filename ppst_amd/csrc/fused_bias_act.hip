// fused bias + activation, elementwise.
// Replaces models/networks/stylegan2_op/fused_bias_act_kernel.cu:19-49 (+launch :52-98):
//   y = f(x + b[(i / step_b) % size_b]) * scale,  f selected by act*10+grad (.cu:35-46).
// HBM-bound: 8 B per element (+4 with a reference tensor).  16-B loads/stores per lane,
// grid-stride, up to 16 blocks per CU.
#include "common.h"

__device__ __forceinline__ float fba_one(float x, float ref, int mode, float alpha) {
  switch (mode) {
    case 30: return x > 0.f ? x : x * alpha;
    case 31: return ref > 0.f ? x : x * alpha;
    case 32: return 0.f;
    case 12: return 0.f;
    default: return x;  // 10, 11 (linear), and the reference's `default:` fallthrough
  }
}

// ST: element type of x, b, ref and y (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79:
// all four tensors share the input's type); the arithmetic runs in fp32 and the result is rounded once.
template <bool VEC, int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(const void* __restrict__ x, const void* __restrict__ b,
                                                             const void* __restrict__ ref, void* __restrict__ y,
                                                             int64_t n, int step_b, int size_b, int mode, float alpha,
                                                             float scale) {
  if (VEC) {
    // step_b % 4 == 0 (or no bias): the 4 elements of a float4 share one bias value
    int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      float4 v = st_ld4<ST>(x, i << 2);
      float bb = b ? st_ld1<ST>(b, ((i << 2) / step_b) % size_b) : 0.f;
      float4 r = ref ? st_ld4<ST>(ref, i << 2) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = fba_one(v.x + bb, r.x, mode, alpha) * scale;
      o.y = fba_one(v.y + bb, r.y, mode, alpha) * scale;
      o.z = fba_one(v.z + bb, r.z, mode, alpha) * scale;
      o.w = fba_one(v.w + bb, r.w, mode, alpha) * scale;
      st_st4<ST>(y, i << 2, o);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      float v = st_ld1<ST>(x, i);
      if (b) v += st_ld1<ST>(b, (i / step_b) % size_b);
      float r = ref ? st_ld1<ST>(ref, i) : 0.f;
      st_st1<ST>(y, i, fba_one(v, r, mode, alpha) * scale);
    }
  }
}

// double: the reference's dispatcher takes it (AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79: scalar_t =
// double, alpha / scale cast to scalar_t); arithmetic in double like the reference.  Not on the hot path: one scalar form.
__global__ __launch_bounds__(256) void fused_bias_act_f64_kernel(const double* __restrict__ x, const double* __restrict__ b,
                                                                 const double* __restrict__ ref, double* __restrict__ y, int64_t n,
                                                                 int step_b, int size_b, int mode, double alpha, double scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    const double r = ref ? ref[i] : 0.0;
    double o;
    switch (mode) {
      case 30: o = v > 0.0 ? v : v * alpha; break;
      case 31: o = r > 0.0 ? v : v * alpha; break;
      case 32: case 12: o = 0.0; break;
      default: o = v;
    }
    y[i] = o * scale;
  }
}

extern "C" int ppst_fused_bias_act(const void* x, const void* b, const void* ref, void* y, int64_t n, int step_b,
                                   int size_b, int act, int grad, float alpha, float scale, int dtype, void* stream) {
  if (dtype != PPST_F32 && dtype != PPST_F16 && dtype != PPST_BF16 && dtype != PPST_F64) return PPST_EUNSUPPORTED;
  if (n < 0 || grad < 0 || grad > 2 || (b && (step_b <= 0 || size_b <= 0))) return PPST_EINVAL;
  if (n == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int mode = act * 10 + grad;
  if (dtype == PPST_F64) {
    int64_t nb = cdiv64(n, 256);
    if (nb > 256 * 16) nb = 256 * 16;
    PPST_LAUNCH(fused_bias_act_f64_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), (const double*)x, (const double*)b,
                (const double*)ref, (double*)y, n, step_b, size_b, mode, (double)alpha, (double)scale);
    return PPST_LAUNCH_CHECK();
  }
  bool vec = (n % 4 == 0) && (!b || step_b % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ref) % (dtype == PPST_F32 ? 16 : 8) == 0);
  int64_t work = vec ? n / 4 : n;
  int64_t blocks = cdiv64(work, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (vec)
    PPST_ST_SWITCH(dtype, PPST_LAUNCH((fused_bias_act_kernel<true, ST_>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                                      x, b, ref, y, n, step_b, size_b, mode, alpha, scale));
  else
    PPST_ST_SWITCH(dtype, PPST_LAUNCH((fused_bias_act_kernel<false, ST_>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                                      x, b, ref, y, n, step_b, size_b, mode, alpha, scale));
  return PPST_LAUNCH_CHECK();
}
