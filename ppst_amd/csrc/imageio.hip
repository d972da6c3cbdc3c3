// Image preprocessing in front of the swap path, on the device:
// Pillow's 8-bit bicubic resample (Image.resize(..., BICUBIC) of data/base_dataset.py:141-168) as two
// integer passes over interleaved uint8 HWC images, and ToTensor + Normalize(0.5, 0.5).
// The 22-bit fixed-point coefficient tables are host logic (ppst_amd/imageio.py mirrors Pillow's
// precompute_coeffs / normalize_coeffs_8bpc); the kernels are pure integer: bit-exact by construction.
#include "common.h"

// One thread = one output sample (pixel, channel).  HORIZ: out[b][y][xx][c] = clip8((2^21 + sum_k in[b][y][xmin+k][c] *
// coef[xx][k]) >> 22); vertical: the same along y.  bounds[i] = (first, count), coef [n_out][ksize].
template <bool HORIZ>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y,
                                                          const int* __restrict__ bounds, const int* __restrict__ coef, int ksize,
                                                          int in_h, int in_w, int out_h, int out_w, int C, unsigned total,
                                                          FastDiv d_c, FastDiv d_w, FastDiv d_h) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned cu, xu, yu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cu);
    r = fd_divmod(r, d_w, xu);
    const int b = (int)fd_divmod(r, d_h, yu);
    const int c = (int)cu, ox = (int)xu, oy = (int)yu;
    const int i = HORIZ ? ox : oy;
    const int first = bounds[2 * i], n = bounds[2 * i + 1];
    const int* k = coef + (int64_t)i * ksize;
    int ss = 1 << 21;
    if (HORIZ) {
      const unsigned char* p = x + (((int64_t)b * in_h + oy) * in_w + first) * C + c;
      for (int j = 0; j < n; ++j) ss += (int)p[(int64_t)j * C] * k[j];
    } else {
      const unsigned char* p = x + (((int64_t)b * in_h + first) * in_w + ox) * C + c;
      for (int j = 0; j < n; ++j) ss += (int)p[(int64_t)j * in_w * C] * k[j];
    }
    ss >>= 22;  // arithmetic shift (Pillow's clip8 table is indexed by the shifted value)
    y[t64] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

extern "C" int ppst_resample_u8(const void* x, void* y, int B, int in_h, int in_w, int C, int out_size, int horizontal,
                                const void* bounds, const void* coef, int ksize, void* stream) {
  if (B < 0 || in_h <= 0 || in_w <= 0 || C <= 0 || out_size <= 0 || ksize <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y || !bounds || !coef) return PPST_ENULL;
  const int out_h = horizontal ? in_h : out_size, out_w = horizontal ? out_size : in_w;
  int64_t total = (int64_t)B * out_h * out_w * C;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  const FastDiv d_c = make_fastdiv(C), d_w = make_fastdiv(out_w), d_h = make_fastdiv(out_h);
  if (horizontal)
    PPST_LAUNCH(resample_u8_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const unsigned char*)x,
                (unsigned char*)y, (const int*)bounds, (const int*)coef, ksize, in_h, in_w, out_h, out_w, C, (unsigned)total, d_c, d_w, d_h);
  else
    PPST_LAUNCH(resample_u8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const unsigned char*)x,
                (unsigned char*)y, (const int*)bounds, (const int*)coef, ksize, in_h, in_w, out_h, out_w, C, (unsigned)total, d_c, d_w, d_h);
  return PPST_LAUNCH_CHECK();
}

// transforms.ToTensor (uint8 HWC -> float CHW, .div(255)) + Normalize(mean, std) per channel: (v/255 - mean)/std in fp32,
// the same operation order as torchvision.
__global__ __launch_bounds__(256) void u8_to_tensor_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, int C, unsigned P,
                                                           unsigned total, float mean, float stdv, FastDiv d_p, FastDiv d_c) {
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned pu, cu;
    unsigned r = fd_divmod((unsigned)t64, d_p, pu);   // output index = ((b*C + c)*P + p)
    const unsigned b = fd_divmod(r, d_c, cu);
    float v = (float)x[((int64_t)b * P + pu) * C + cu] / 255.0f;
    y[t64] = (v - mean) / stdv;
  }
}
extern "C" int ppst_u8_to_tensor(const void* x, void* y, int B, int H, int W, int C, float mean, float stdv, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || stdv == 0.f) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t P = (int64_t)H * W, total = (int64_t)B * C * P;
  if (total > PPST_IDX32_MAX) return PPST_EINVAL;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  PPST_LAUNCH(u8_to_tensor_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const unsigned char*)x, (float*)y, C,
              (unsigned)P, (unsigned)total, mean, stdv, make_fastdiv((unsigned)P), make_fastdiv(C));
  return PPST_LAUNCH_CHECK();
}
