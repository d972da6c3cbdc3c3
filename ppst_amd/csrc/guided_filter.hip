// Colour-guided filter post-process of PPSTModel.decode (models/ppst_model.py:288-306 ->
// photo_gif.py:25-46 -> cv2.ximgproc.guidedFilter(guide=content, src=output, radius, eps)).
// The arithmetic lives in opencv-contrib 4.8.1.78, which is not vendored by the reference:
// this restates the published algorithm (He et al., as OpenCV implements it: fp32 work type,
// normalised (2r+1)^2 box mean with BORDER_REFLECT, eps on the covariance diagonal, 3x3
// symmetric inverse by cofactors, cvRound + saturate to uint8).  PARITY UNPINNED against
// OpenCV itself; pinned against oracle/ppst_oracle.py:guided_filter_color.
//
// Box means are separable direct sums (no running sums: order-independent fp32 error):
//   H pass: one block per image row, row staged in LDS with reflected borders;
//   V pass: lanes along x (coalesced), 2r+1 row reads per output served by L1/L2.
// Pipeline per batch (planes are [B][plane][H][W] fp32 in the caller's workspace):
//   gf_h_stage1 (uint8 -> 21 h-sums: I(3), p(3), I_i*I_j(6), I_i*p_c(9))
//   gf_v        (21 means)
//   gf_solve    (-> 12 planes a_c[3], b_c)
//   gf_h        (12 h-sums)
//   gf_v_final  (means of a, b; q = sum a_k I_k + b; round; uint8 + fp32 NCHW (q/255-0.5)*2)
#include "common.h"

#define GF_MAXW 2048
#define GF_MAXR 64

__device__ __forceinline__ int reflect_idx(int i, int n) {  // cv2.BORDER_REFLECT (edge pixel repeated)
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  return i;
}

// plane value for stage 1 from the uint8 inputs at one pixel
__device__ __forceinline__ float gf_plane_value(int pl, const unsigned char* g, const unsigned char* s) {
  float I0 = g[0], I1 = g[1], I2 = g[2];
  float P0 = s[0], P1 = s[1], P2 = s[2];
  switch (pl) {
    case 0: return I0; case 1: return I1; case 2: return I2;
    case 3: return P0; case 4: return P1; case 5: return P2;
    case 6: return I0 * I0; case 7: return I0 * I1; case 8: return I0 * I2;
    case 9: return I1 * I1; case 10: return I1 * I2; case 11: return I2 * I2;
    default: {
      int k = pl - 12, c = k / 3, i = k - c * 3;  // 12 + c*3 + i = I_i * p_c
      float Iv = i == 0 ? I0 : (i == 1 ? I1 : I2);
      float Pv = c == 0 ? P0 : (c == 1 ? P1 : P2);
      return Iv * Pv;
    }
  }
}

// H pass.  grid = (H, nplanes, B).  STAGE1: read uint8 guide/src; else read fp32 planes.
template <bool STAGE1>
__global__ __launch_bounds__(256) void gf_h_kernel(const unsigned char* __restrict__ guide, const unsigned char* __restrict__ src,
                                                   const float* __restrict__ in, float* __restrict__ out, int H, int W, int r,
                                                   int nplanes) {
  __shared__ float row[GF_MAXW + 2 * GF_MAXR];
  const int y = blockIdx.x, pl = blockIdx.y, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  for (int i = threadIdx.x; i < W + 2 * r; i += 256) {
    int x = reflect_idx(i - r, W);
    float v;
    if (STAGE1) {
      int64_t o = (((int64_t)b * H + y) * W + x) * 3;
      v = gf_plane_value(pl, guide + o, src + o);
    } else {
      v = in[((int64_t)b * nplanes + pl) * P + (int64_t)y * W + x];
    }
    row[i] = v;
  }
  __syncthreads();
  float* o = out + ((int64_t)b * nplanes + pl) * P + (int64_t)y * W;
  for (int x = threadIdx.x; x < W; x += 256) {
    float s = 0.f;
    for (int k = 0; k <= 2 * r; ++k) s += row[x + k];
    o[x] = s;
  }
}

// V pass: out = (sum over 2r+1 rows of in) / (2r+1)^2.  grid-stride over (b, plane, y, x).
__global__ __launch_bounds__(256) void gf_v_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int r,
                                                   int64_t total) {
  const float inv = 1.f / (float)((2 * r + 1) * (2 * r + 1));
  const int64_t P = (int64_t)H * W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int x = (int)(t % W);
    int64_t q = t / W;
    int y = (int)(q % H);
    int64_t bp = q / H;
    const float* col = in + bp * P + x;
    float s = 0.f;
    for (int k = -r; k <= r; ++k) s += col[(int64_t)reflect_idx(y + k, H) * W];
    out[t] = s * inv;
  }
}

// The same V pass for the path's radius (RR = 30), register-blocked: one thread owns GV_R consecutive output rows of one column,
// loads the GV_R + 2 RR rows they span ONCE (76 loads for 16 outputs instead of 61 per output) and sums every output's window
// from registers in the same order k = -RR .. RR as gf_v_kernel -- bit-identical to it.  (The plain kernel issued
// 61 x 44 M loads per batch of 8 images: 1.7 ms, L2-bound; 7 % of the grid workload.)
#define GV_R 16
template <int RR>
__global__ __launch_bounds__(256) void gf_v_blocked_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                           int ytiles, unsigned total, FastDiv d_w, FastDiv d_t) {
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  const int64_t P = (int64_t)H * W;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned xu, tyu;
    unsigned q = fd_divmod((unsigned)t64, d_w, xu);
    const unsigned bp = fd_divmod(q, d_t, tyu);
    const int x = (int)xu, y0 = (int)tyu * GV_R;
    const float* col = in + (int64_t)bp * P + x;
    float v[GV_R + 2 * RR];
#pragma unroll
    for (int i = 0; i < GV_R + 2 * RR; ++i) v[i] = col[(int64_t)reflect_idx(min(y0 - RR + i, H - 1 + RR), H) * W];
    float* o = out + (int64_t)bp * P + x;
#pragma unroll
    for (int j = 0; j < GV_R; ++j) {
      if (y0 + j >= H) break;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k <= 2 * RR; ++k) s += v[j + k];
      o[(int64_t)(y0 + j) * W] = s * inv;
    }
  }
}

// per-pixel 3x3 solve.  means: [B][21][P] -> ab: [B][12][P] (a_c0,a_c1,a_c2,b_c for c=0..2)
__global__ __launch_bounds__(256) void gf_solve_kernel(const float* __restrict__ m, float* __restrict__ ab, int64_t P, float eps,
                                                       int64_t total) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int64_t b = t / P, p = t - b * P;
    const float* mp = m + b * 21 * P + p;
    float mI0 = mp[0], mI1 = mp[P], mI2 = mp[2 * P];
    float a00 = mp[6 * P] - mI0 * mI0 + eps, a01 = mp[7 * P] - mI0 * mI1, a02 = mp[8 * P] - mI0 * mI2;
    float a11 = mp[9 * P] - mI1 * mI1 + eps, a12 = mp[10 * P] - mI1 * mI2, a22 = mp[11 * P] - mI2 * mI2 + eps;
    float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    float c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    float det = a00 * c00 + a01 * c01 + a02 * c02;
    float i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
    float* o = ab + b * 12 * P + p;
    for (int c = 0; c < 3; ++c) {
      float mp_c = mp[(3 + c) * P];
      float cp0 = mp[(12 + c * 3 + 0) * P] - mI0 * mp_c;
      float cp1 = mp[(12 + c * 3 + 1) * P] - mI1 * mp_c;
      float cp2 = mp[(12 + c * 3 + 2) * P] - mI2 * mp_c;
      float A0 = i00 * cp0 + i01 * cp1 + i02 * cp2;
      float A1 = i01 * cp0 + i11 * cp1 + i12 * cp2;
      float A2 = i02 * cp0 + i12 * cp1 + i22 * cp2;
      float bb = mp_c - A0 * mI0 - A1 * mI1 - A2 * mI2;
      o[(c * 4 + 0) * P] = A0; o[(c * 4 + 1) * P] = A1; o[(c * 4 + 2) * P] = A2; o[(c * 4 + 3) * P] = bb;
    }
  }
}

// final V pass over the 12 h-summed (a,b) planes + combination with the guide.
__global__ __launch_bounds__(256) void gf_v_final_kernel(const float* __restrict__ hs, const unsigned char* __restrict__ guide,
                                                         float* __restrict__ out, unsigned char* __restrict__ out_u8, int H, int W,
                                                         int r, int64_t total) {
  const float inv = 1.f / (float)((2 * r + 1) * (2 * r + 1));
  const int64_t P = (int64_t)H * W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int x = (int)(t % W);
    int64_t q = t / W;
    int y = (int)(q % H);
    int64_t b = q / H;
    float m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = 0.f;
    const float* base = hs + b * 12 * P + x;
    for (int k = -r; k <= r; ++k) {
      int64_t ro = (int64_t)reflect_idx(y + k, H) * W;
#pragma unroll
      for (int i = 0; i < 12; ++i) m[i] += base[i * P + ro];
    }
    const unsigned char* g = guide + ((b * H + y) * W + x) * 3;
    float I0 = g[0], I1 = g[1], I2 = g[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float qv = (m[c * 4] * inv) * I0 + (m[c * 4 + 1] * inv) * I1 + (m[c * 4 + 2] * inv) * I2 + m[c * 4 + 3] * inv;
      float rq = fminf(fmaxf(rintf(qv), 0.f), 255.f);
      if (out_u8) out_u8[((b * H + y) * W + x) * 3 + c] = (unsigned char)rq;
      if (out) out[(b * 3 + c) * P + (int64_t)y * W + x] = (rq / 255.0f - 0.5f) * 2.f;  // ToTensor, (x-0.5)*2 (ppst_model.py:301-303)
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same pipeline with SLIDING windows (the path's radius only, RR = 30).  Round 3's passes summed 2 RR + 1 = 61
// values per output and plane -- 61 LDS reads per output in the H passes, 61 L2 reads per output in the final V pass: 1.38 ms per
// batch of four 1024^2 images, 27 GB/s against the algorithmic 9 B / pixel.  Here a thread owns GF_SEG consecutive outputs of one
// plane: its first window is a direct sum, the next ones add the entering and subtract the leaving value (5.7 / 3.9 reads per
// output instead of 61).
//   * the 15 moment planes of stage 1 are INTEGERS: a 61-term row sum is < 2^22, exact in fp32 whatever the order; the column
//     sums (< 2^28) run in uint32 -- exact too, and the mean is formed from the exact sum (the direct fp32 sums of round 3, and
//     of the oracle, carry ~1e-7 relative rounding there);
//   * the (a, b) planes of stage 2 are floats: a window slides over at most GF_SEG - 1 = 15 (rows: GF_VSEG - 1 = 63) steps before
//     the next thread starts from a direct sum again, so the drift is bounded by that many roundings of a sum of 61 terms.
// Bar: <= 1 uint8 LSB against the oracle (tests/gpu_diag.py:t_guided), as before.
#define GF_SEG 16
#ifndef GF_VSEG
#define GF_VSEG 32
#endif
// LDS row index with one pad slot per 16 entries: lanes are 16 entries apart (one segment each) -- unpadded, all of a wave's
// reads fall on two banks
#define GF_ROW(i) ((i) + ((i) >> 4))

// H pass of stage 1: grid (H, column chunks of GF_HCW outputs, B), 256 threads.  Phase 1: the 21 moments of every pixel of the
// chunk (+ RR on both sides, reflected) are formed ONCE and laid out as 21 padded LDS rows; phase 2: thread = (plane group, 16-output
// segment) slides the window over its segment plane by plane -- one LDS read and one add per entering / leaving value.
#define GF_HCW 512
#define GF_HROWLEN (GF_ROW(GF_HCW + 2 * GF_MAXR) + 1)
template <int RR>
__global__ __launch_bounds__(256) void gf_h1_slide_kernel(const unsigned char* __restrict__ guide, const unsigned char* __restrict__ src,
                                                          float* __restrict__ out, int H, int W) {
  __shared__ unsigned mom[21][GF_HROWLEN];
  const int y = blockIdx.x, xc0 = blockIdx.y * GF_HCW, b = blockIdx.z;
  const int cw = min(GF_HCW, W - xc0);
  const int64_t P = (int64_t)H * W;
  for (int i = threadIdx.x; i < cw + 2 * RR; i += 256) {
    const int x = reflect_idx(xc0 + i - RR, W);
    const int64_t o = (((int64_t)b * H + y) * W + x) * 3;
    const unsigned I0 = guide[o], I1 = guide[o + 1], I2 = guide[o + 2], P0 = src[o], P1 = src[o + 1], P2 = src[o + 2];
    const int k = GF_ROW(i);
    mom[0][k] = I0; mom[1][k] = I1; mom[2][k] = I2; mom[3][k] = P0; mom[4][k] = P1; mom[5][k] = P2;
    mom[6][k] = __umul24(I0, I0); mom[7][k] = __umul24(I0, I1); mom[8][k] = __umul24(I0, I2);
    mom[9][k] = __umul24(I1, I1); mom[10][k] = __umul24(I1, I2); mom[11][k] = __umul24(I2, I2);
    mom[12][k] = __umul24(I0, P0); mom[13][k] = __umul24(I1, P0); mom[14][k] = __umul24(I2, P0);
    mom[15][k] = __umul24(I0, P1); mom[16][k] = __umul24(I1, P1); mom[17][k] = __umul24(I2, P1);
    mom[18][k] = __umul24(I0, P2); mom[19][k] = __umul24(I1, P2); mom[20][k] = __umul24(I2, P2);
  }
  __syncthreads();
  const int seg = threadIdx.x & 31, pg = threadIdx.x >> 5;        // 8 plane groups x 32 segments
  const int x0 = seg * GF_SEG;
  if (x0 >= cw) return;
  for (int pl = pg; pl < 21; pl += 8) {
    const unsigned* row = mom[pl];
    unsigned s = 0;
    for (int k = 0; k <= 2 * RR; ++k) s += row[GF_ROW(x0 + k)];
    float o[GF_SEG];
    o[0] = (float)s;
#pragma unroll
    for (int j = 1; j < GF_SEG; ++j) {
      // (past the chunk's end the LDS row holds no pixel: clamp the index, the value is dropped below)
      s += row[GF_ROW(min(x0 + j + 2 * RR, cw + 2 * RR - 1))] - row[GF_ROW(x0 + j - 1)];
      o[j] = (float)s;
    }
    float* op = out + ((int64_t)b * 21 + pl) * P + (int64_t)y * W + xc0 + x0;
    if (x0 + GF_SEG <= cw && (W & 3) == 0) {
#pragma unroll
      for (int j = 0; j < GF_SEG; j += 4) *(float4*)(op + j) = make_float4(o[j], o[j + 1], o[j + 2], o[j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < GF_SEG; ++j)
        if (x0 + j < cw) op[j] = o[j];
    }
  }
}

// V pass of stage 1 + the per-pixel 3x3 solve: a thread owns one column x and GF_VSEG output rows; the 21 column sums slide in
// uint32 (exact), every output row is solved at once: hs [B][21][P] (exact integer row sums) -> ab [B][12][P].
template <int RR>
__global__ __launch_bounds__(256) void gf_v1_solve_slide_kernel(const float* __restrict__ hs, float* __restrict__ ab, int H, int W,
                                                                float eps) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y0 = blockIdx.y * GF_VSEG, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  const float* in = hs + (int64_t)b * 21 * P + x;
  float* o = ab + (int64_t)b * 12 * P + x;
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  unsigned s[21];
#pragma unroll
  for (int pl = 0; pl < 21; ++pl) s[pl] = 0;
  for (int k = -RR; k <= RR; ++k) {
    const int64_t ro = (int64_t)reflect_idx(y0 + k, H) * W;
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) s[pl] += (unsigned)in[pl * P + ro];
  }
  const int y1 = min(y0 + GF_VSEG, H);
  for (int y = y0; y < y1; ++y) {
    float m[21];
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) m[pl] = (float)s[pl] * inv;
    const float mI0 = m[0], mI1 = m[1], mI2 = m[2];
    const float a00 = m[6] - mI0 * mI0 + eps, a01 = m[7] - mI0 * mI1, a02 = m[8] - mI0 * mI2;
    const float a11 = m[9] - mI1 * mI1 + eps, a12 = m[10] - mI1 * mI2, a22 = m[11] - mI2 * mI2 + eps;
    const float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const float det = a00 * c00 + a01 * c01 + a02 * c02;
    const float i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
    const int64_t po = (int64_t)y * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mp_c = m[3 + c];
      const float cp0 = m[12 + c * 3 + 0] - mI0 * mp_c, cp1 = m[12 + c * 3 + 1] - mI1 * mp_c, cp2 = m[12 + c * 3 + 2] - mI2 * mp_c;
      const float A0 = i00 * cp0 + i01 * cp1 + i02 * cp2;
      const float A1 = i01 * cp0 + i11 * cp1 + i12 * cp2;
      const float A2 = i02 * cp0 + i12 * cp1 + i22 * cp2;
      const float bb = mp_c - A0 * mI0 - A1 * mI1 - A2 * mI2;
      o[(c * 4 + 0) * P + po] = A0; o[(c * 4 + 1) * P + po] = A1; o[(c * 4 + 2) * P + po] = A2; o[(c * 4 + 3) * P + po] = bb;
    }
    if (y + 1 < y1) {
      const int64_t rin = (int64_t)reflect_idx(y + 1 + RR, H) * W, rout = (int64_t)reflect_idx(y - RR, H) * W;
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) s[pl] += (unsigned)in[pl * P + rin] - (unsigned)in[pl * P + rout];
    }
  }
}

// ---- stage 1 in the other order (round 4, later): V pass FIRST, on the raw uint8 rows -------------------------------------------
// The H-then-V order above makes the V pass slide over 21 fp32 planes: a thread's window start (61 rows) and every leaving row are
// re-read -- 3.9 row visits per output at 84 bytes each (PMC: 1.34 GB of fetches per batch of four 1024^2 images for 352 MB of
// planes).  With the V pass first, what is re-read is the 6 bytes of a pixel (guide + source): the 21 moments of the entering /
// leaving row are formed on the fly and the 21 column sums (< 2^22: exact as fp32) written once; the H pass then stages each row of
// column sums once in LDS, slides in uint32 (< 2^28: exact) and solves the pixel's 3x3 system from LDS.  Same integers as the H-then-V
// order, so the same (a, b) bit for bit.
template <int RR>
__global__ __launch_bounds__(256) void gf_v1m_slide_kernel(const unsigned char* __restrict__ guide, const unsigned char* __restrict__ src,
                                                           float* __restrict__ out, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y0 = blockIdx.y * GF_VSEG, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  const unsigned char* gp = guide + ((int64_t)b * P + x) * 3;
  const unsigned char* sp = src + ((int64_t)b * P + x) * 3;
  unsigned s[21];
#pragma unroll
  for (int pl = 0; pl < 21; ++pl) s[pl] = 0;
  auto row = [&](int y, bool add) __attribute__((always_inline)) {
    const int64_t o = (int64_t)reflect_idx(y, H) * W * 3;
    const unsigned I0 = gp[o], I1 = gp[o + 1], I2 = gp[o + 2], P0 = sp[o], P1 = sp[o + 1], P2 = sp[o + 2];
#define GFM(a_, b_) ((unsigned)__umul24(a_, b_))
    const unsigned m[21] = {I0, I1, I2, P0, P1, P2, GFM(I0, I0), GFM(I0, I1), GFM(I0, I2), GFM(I1, I1), GFM(I1, I2), GFM(I2, I2),
                            GFM(I0, P0), GFM(I1, P0), GFM(I2, P0), GFM(I0, P1), GFM(I1, P1), GFM(I2, P1), GFM(I0, P2), GFM(I1, P2), GFM(I2, P2)};
#undef GFM
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) s[pl] = add ? s[pl] + m[pl] : s[pl] - m[pl];
  };
  for (int k = -RR; k <= RR; ++k) row(y0 + k, true);
  const int y1 = min(y0 + GF_VSEG, H);
  float* op = out + (int64_t)b * 21 * P + x;
  for (int y = y0; y < y1; ++y) {
    const int64_t po = (int64_t)y * W;
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) op[pl * P + po] = (float)s[pl];
    if (y + 1 < y1) {
      row(y + 1 + RR, true);
      row(y - RR, false);
    }
  }
}

// H pass over the 21 planes of column sums + the per-pixel 3x3 solve: grid (H, column chunks of GF_HCW outputs, B), 256 threads.
// Phase 1: the row of every plane (+ RR on both sides, reflected) into LDS as uint32; phase 2: thread = (plane group, 16-output
// segment) slides the window, keeps its outputs in registers; phase 3 (behind a barrier: every window has been read) the window sums
// replace the row in LDS; phase 4: thread = pixel reads its 21 sums and solves: cs [B][21][P] -> ab [B][12][P].
template <int RR>
__global__ __launch_bounds__(256) void gf_h1s_solve_kernel(const float* __restrict__ cs, float* __restrict__ ab, int H, int W, float eps) {
  __shared__ unsigned mom[21][GF_HROWLEN];
  const int y = blockIdx.x, xc0 = blockIdx.y * GF_HCW, b = blockIdx.z;
  const int cw = min(GF_HCW, W - xc0);
  const int64_t P = (int64_t)H * W;
  {
    const float* ip = cs + (int64_t)b * 21 * P + (int64_t)y * W;
    for (int i = threadIdx.x; i < cw + 2 * RR; i += 256) {       // the 21 planes' loads of a pixel go out together
      const int x = reflect_idx(xc0 + i - RR, W);
      float v[21];
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) v[pl] = ip[pl * P + x];
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) mom[pl][GF_ROW(i)] = (unsigned)v[pl];
    }
  }
  __syncthreads();
  const int seg = threadIdx.x & 31, pg = threadIdx.x >> 5;        // 8 plane groups x 32 segments
  const int x0 = seg * GF_SEG;
  unsigned o[3][GF_SEG];
  if (x0 < cw) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int pl = pg + 8 * q;
      if (pl >= 21) break;
      const unsigned* row = mom[pl];
      unsigned sum = 0;
      for (int k = 0; k <= 2 * RR; ++k) sum += row[GF_ROW(x0 + k)];
      o[q][0] = sum;
#pragma unroll
      for (int j = 1; j < GF_SEG; ++j) {
        sum += row[GF_ROW(min(x0 + j + 2 * RR, cw + 2 * RR - 1))] - row[GF_ROW(x0 + j - 1)];
        o[q][j] = sum;
      }
    }
  }
  __syncthreads();
  if (x0 < cw) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int pl = pg + 8 * q;
      if (pl >= 21) break;
#pragma unroll
      for (int j = 0; j < GF_SEG; ++j) mom[pl][GF_ROW(x0 + j)] = o[q][j];
    }
  }
  __syncthreads();
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  float* op = ab + (int64_t)b * 12 * P + (int64_t)y * W + xc0;
  for (int px = threadIdx.x; px < cw; px += 256) {
    float m[21];
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) m[pl] = (float)mom[pl][GF_ROW(px)] * inv;
    const float mI0 = m[0], mI1 = m[1], mI2 = m[2];
    const float a00 = m[6] - mI0 * mI0 + eps, a01 = m[7] - mI0 * mI1, a02 = m[8] - mI0 * mI2;
    const float a11 = m[9] - mI1 * mI1 + eps, a12 = m[10] - mI1 * mI2, a22 = m[11] - mI2 * mI2 + eps;
    const float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const float det = a00 * c00 + a01 * c01 + a02 * c02;
    const float i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mp_c = m[3 + c];
      const float cp0 = m[12 + c * 3 + 0] - mI0 * mp_c, cp1 = m[12 + c * 3 + 1] - mI1 * mp_c, cp2 = m[12 + c * 3 + 2] - mI2 * mp_c;
      const float A0 = i00 * cp0 + i01 * cp1 + i02 * cp2;
      const float A1 = i01 * cp0 + i11 * cp1 + i12 * cp2;
      const float A2 = i02 * cp0 + i12 * cp1 + i22 * cp2;
      const float bb = mp_c - A0 * mI0 - A1 * mI1 - A2 * mI2;
      op[(c * 4 + 0) * P + px] = A0; op[(c * 4 + 1) * P + px] = A1; op[(c * 4 + 2) * P + px] = A2; op[(c * 4 + 3) * P + px] = bb;
    }
  }
}

// The same kernel carried one pass further: the H pass of STAGE 2 rides on it.  The block solves (a, b) for its GF_HCW outputs AND
// the RR positions on either side (the values gf_h2_slide_kernel would read back, at reflected positions: a virtual position's
// window over the reflection-staged row holds the same columns as the window of the pixel it reflects to), keeps them in LDS in
// place of the window sums (a thread reads the 21 sums of ITS pixel, then writes that pixel's 12 values), and slides the 12 rows
// exactly as gf_h2_slide_kernel does (same summation order: bit-identical).  The 12 (a, b) planes are neither written nor read
// back (96 bytes per pixel) and one launch goes: cs [B][21][P] -> row sums of (a, b) [B][12][P].
template <int RR>
__global__ __launch_bounds__(256) void gf_h1s_solve_h2_kernel(const float* __restrict__ cs, float* __restrict__ out, int H, int W, float eps) {
  __shared__ unsigned mom[21][GF_ROW(GF_HCW + 4 * RR) + 1];       // 56 KB at RR = 30: two blocks per CU
  const int y = blockIdx.x, xc0 = blockIdx.y * GF_HCW, b = blockIdx.z;
  const int cw = min(GF_HCW, W - xc0);
  const int64_t P = (int64_t)H * W;
  const int ns = cw + 4 * RR, na = cw + 2 * RR;          // staged columns, (a, b) positions
  {
    const float* ip = cs + (int64_t)b * 21 * P + (int64_t)y * W;
    for (int i = threadIdx.x; i < ns; i += 256) {
      const int x = reflect_idx(xc0 + i - 2 * RR, W);
      float v[21];
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) v[pl] = ip[pl * P + x];
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) mom[pl][GF_ROW(i)] = (unsigned)v[pl];
    }
  }
  __syncthreads();
  // window sums of the na positions: thread = (plane group, 16-position segment); 37 segments of 16 cover 572 + 20
  constexpr int NSEG = (GF_HCW + 2 * RR + GF_SEG - 1) / GF_SEG;
  static_assert(21 * NSEG <= 4 * 256, "four (plane, segment) items per thread");
  unsigned o[4][GF_SEG];
  int items[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int it = threadIdx.x + 256 * q;           // item = plane * NSEG + segment
    items[q] = it < 21 * NSEG ? it : -1;
    if (items[q] < 0) continue;
    const int pl = it / NSEG, x0 = (it - pl * NSEG) * GF_SEG;
    if (x0 >= na) { items[q] = -1; continue; }
    const unsigned* row = mom[pl];
    unsigned sum = 0;
    for (int k = 0; k <= 2 * RR; ++k) sum += row[GF_ROW(x0 + k)];
    o[q][0] = sum;
#pragma unroll
    for (int j = 1; j < GF_SEG; ++j) {
      sum += row[GF_ROW(min(x0 + j + 2 * RR, ns - 1))] - row[GF_ROW(x0 + j - 1)];
      o[q][j] = sum;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (items[q] < 0) continue;
    const int pl = items[q] / NSEG, x0 = (items[q] - pl * NSEG) * GF_SEG;
#pragma unroll
    for (int j = 0; j < GF_SEG; ++j) mom[pl][GF_ROW(x0 + j)] = o[q][j];
  }
  __syncthreads();
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  for (int px = threadIdx.x; px < na; px += 256) {
    float m[21];
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) m[pl] = (float)mom[pl][GF_ROW(px)] * inv;
    const float mI0 = m[0], mI1 = m[1], mI2 = m[2];
    const float a00 = m[6] - mI0 * mI0 + eps, a01 = m[7] - mI0 * mI1, a02 = m[8] - mI0 * mI2;
    const float a11 = m[9] - mI1 * mI1 + eps, a12 = m[10] - mI1 * mI2, a22 = m[11] - mI2 * mI2 + eps;
    const float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const float det = a00 * c00 + a01 * c01 + a02 * c02;
    const float i00 = c00 / det, i01 = c01 / det, i02 = c02 / det, i11 = c11 / det, i12 = c12 / det, i22 = c22 / det;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mp_c = m[3 + c];
      const float cp0 = m[12 + c * 3 + 0] - mI0 * mp_c, cp1 = m[12 + c * 3 + 1] - mI1 * mp_c, cp2 = m[12 + c * 3 + 2] - mI2 * mp_c;
      const float A0 = i00 * cp0 + i01 * cp1 + i02 * cp2;
      const float A1 = i01 * cp0 + i11 * cp1 + i12 * cp2;
      const float A2 = i02 * cp0 + i12 * cp1 + i22 * cp2;
      const float bb = mp_c - A0 * mI0 - A1 * mI1 - A2 * mI2;
      mom[c * 4 + 0][GF_ROW(px)] = __float_as_uint(A0); mom[c * 4 + 1][GF_ROW(px)] = __float_as_uint(A1);
      mom[c * 4 + 2][GF_ROW(px)] = __float_as_uint(A2); mom[c * 4 + 3][GF_ROW(px)] = __float_as_uint(bb);
    }
  }
  __syncthreads();
  // stage-2 H pass over the 12 rows (positions 0 .. na - 1 = image columns xc0 - RR ..): as gf_h2_slide_kernel
  const int seg = threadIdx.x & 31, pg = threadIdx.x >> 5;        // 8 plane groups x 32 segments
  const int x0 = seg * GF_SEG;
  if (x0 >= cw) return;
  for (int pl = pg; pl < 12; pl += 8) {
    const float* row = (const float*)mom[pl];
    float sum = 0.f;
    for (int k = 0; k <= 2 * RR; ++k) sum += row[GF_ROW(x0 + k)];
    float ov[GF_SEG];
    ov[0] = sum;
#pragma unroll
    for (int j = 1; j < GF_SEG; ++j) {
      sum += row[GF_ROW(min(x0 + j + 2 * RR, cw + 2 * RR - 1))] - row[GF_ROW(x0 + j - 1)];
      ov[j] = sum;
    }
    float* op = out + ((int64_t)b * 12 + pl) * P + (int64_t)y * W + xc0 + x0;
    if (x0 + GF_SEG <= cw && (W & 3) == 0) {
#pragma unroll
      for (int j = 0; j < GF_SEG; j += 4) *(float4*)(op + j) = make_float4(ov[j], ov[j + 1], ov[j + 2], ov[j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < GF_SEG; ++j)
        if (x0 + j < cw) op[j] = ov[j];
    }
  }
}

// H pass of stage 2 (12 float planes): grid (H, column chunks, B), 128 threads = 4 plane groups x 32 segments; the chunk's 12 rows
// are staged at once.
template <int RR>
__global__ __launch_bounds__(128) void gf_h2_slide_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  __shared__ float rows[12][GF_HROWLEN];
  const int y = blockIdx.x, xc0 = blockIdx.y * GF_HCW, b = blockIdx.z;
  const int cw = min(GF_HCW, W - xc0);
  const int64_t P = (int64_t)H * W;
  {
    const float* ip = in + (int64_t)b * 12 * P + (int64_t)y * W;
    for (int i = threadIdx.x; i < cw + 2 * RR; i += 128) {       // the 12 planes' loads of a pixel go out together
      const int x = reflect_idx(xc0 + i - RR, W);
      float v[12];
#pragma unroll
      for (int pl = 0; pl < 12; ++pl) v[pl] = ip[pl * P + x];
#pragma unroll
      for (int pl = 0; pl < 12; ++pl) rows[pl][GF_ROW(i)] = v[pl];
    }
  }
  __syncthreads();
  const int seg = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int x0 = seg * GF_SEG;
  if (x0 >= cw) return;
  for (int pl = pg; pl < 12; pl += 4) {
    const float* row = rows[pl];
    float s = 0.f;
    for (int k = 0; k <= 2 * RR; ++k) s += row[GF_ROW(x0 + k)];
    float o[GF_SEG];
    o[0] = s;
#pragma unroll
    for (int j = 1; j < GF_SEG; ++j) {
      s += row[GF_ROW(min(x0 + j + 2 * RR, cw + 2 * RR - 1))] - row[GF_ROW(x0 + j - 1)];
      o[j] = s;
    }
    float* op = out + ((int64_t)b * 12 + pl) * P + (int64_t)y * W + xc0 + x0;
    if (x0 + GF_SEG <= cw && (W & 3) == 0) {
#pragma unroll
      for (int j = 0; j < GF_SEG; j += 4) *(float4*)(op + j) = make_float4(o[j], o[j + 1], o[j + 2], o[j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < GF_SEG; ++j)
        if (x0 + j < cw) op[j] = o[j];
    }
  }
}

// final V pass: a thread owns one column and GF_VSEG rows; 12 sliding float column sums, combination with the guide, rounding.
template <int RR>
__global__ __launch_bounds__(256) void gf_v2_final_slide_kernel(const float* __restrict__ hs, const unsigned char* __restrict__ guide,
                                                                float* __restrict__ out, unsigned char* __restrict__ out_u8, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y0 = blockIdx.y * GF_VSEG, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  const float* in = hs + (int64_t)b * 12 * P + x;
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  float s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0.f;
  for (int k = -RR; k <= RR; ++k) {
    const int64_t ro = (int64_t)reflect_idx(y0 + k, H) * W;
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] += in[i * P + ro];
  }
  const int y1 = min(y0 + GF_VSEG, H);
  for (int y = y0; y < y1; ++y) {
    const unsigned char* g = guide + (((int64_t)b * H + y) * W + x) * 3;
    const float I0 = g[0], I1 = g[1], I2 = g[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float qv = (s[c * 4] * inv) * I0 + (s[c * 4 + 1] * inv) * I1 + (s[c * 4 + 2] * inv) * I2 + s[c * 4 + 3] * inv;
      const float rq = fminf(fmaxf(rintf(qv), 0.f), 255.f);
      if (out_u8) out_u8[(((int64_t)b * H + y) * W + x) * 3 + c] = (unsigned char)rq;
      if (out) out[((int64_t)b * 3 + c) * P + (int64_t)y * W + x] = (rq / 255.0f - 0.5f) * 2.f;  // ToTensor, (x-0.5)*2 (ppst_model.py:301-303)
    }
    if (y + 1 < y1) {
      const int64_t rin = (int64_t)reflect_idx(y + 1 + RR, H) * W, rout = (int64_t)reflect_idx(y - RR, H) * W;
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] += in[i * P + rin] - in[i * P + rout];
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the radius-30 filter in TWO launches that keep every box sum on the chip (round-4 verdict: 439 B / pixel of HBM traffic
// by counter against 9 B / pixel algorithmic -- the three-launch form hands 21 + 12 fp32 planes through HBM and re-reads them).
//   gf_s1_fused_kernel: a block owns GS_WC = 192 output columns (+ RR on both sides = 252 of its 256 threads) and GS_VS1 rows.  Thread =
//     column: its 21 vertical window sums live in REGISTERS and slide down the rows (the entering / leaving row's moments formed from
//     the 6 uint8 bytes of the pixel, exact uint32); per row the 252 column sums go to LDS, 21 x 12 threads slide the horizontal
//     window over 16-output segments (exact uint32), the window sums go back to LDS and thread = pixel solves its 3x3 system.
//     Only (a, b) leave the kernel -- as 12 IEEE-half planes (24 B / pixel; they feed a 61 x 61 mean): each thread carries the
//     rounding residual of every value down its column (error diffusion), so a vertical window sum of the stored halves differs from
//     the fp32 sum by at most the two end residuals -- not 61 correlated roundings (a flat image region rounds every b alike).
//   gf_s2_fused_kernel: the same structure over the 12 half planes (fp32 sums, sliding with the bounded drift of the round-4
//     passes), ending in q = mean(a) . I + mean(b), rounded to uint8 like cv2 (saturate_cast<uchar>(cvRound)).
// HBM traffic per pixel: 6 B x halo factors in, 24 out; 24 x (1.31 columns x (VS2 + 60) / VS2 rows) in, 3 guide, 12 (+ 3) out.
// Same stage-1 integers as the three-launch form, so the same (a, b) before their rounding to half.
#define GS_WC 192
#define GS_NSEG (GS_WC / GF_SEG)
template <int RR, int VS>
__global__ __launch_bounds__(256) void gf_s1_fused_kernel(const unsigned char* __restrict__ guide, const unsigned char* __restrict__ src,
                                                          unsigned short* __restrict__ ab, int H, int W, float eps) {
  static_assert(GS_WC + 2 * RR <= 256 && GS_WC % GF_SEG == 0 && 21 * GS_NSEG <= 256, "strip geometry");
  __shared__ unsigned col[21][GF_ROW(256) + 1];
  __shared__ unsigned win[21][GF_ROW(GS_WC) + 1];
  const int t = threadIdx.x, x0 = blockIdx.x * GS_WC, y0 = blockIdx.y * VS, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  const int xc = reflect_idx(min(x0 - RR + t, W - 1 + RR), W);          // (columns past the image's reflected margin are never used)
  const unsigned char* gp = guide + ((int64_t)b * P + xc) * 3;
  const unsigned char* sp = src + ((int64_t)b * P + xc) * 3;
  unsigned cs[21];
#pragma unroll
  for (int pl = 0; pl < 21; ++pl) cs[pl] = 0;
  struct Px { unsigned I0, I1, I2, P0, P1, P2; };
  auto fetch = [&](int y) __attribute__((always_inline)) -> Px {
    const int64_t o = (int64_t)reflect_idx(y, H) * W * 3;
    return Px{gp[o], gp[o + 1], gp[o + 2], sp[o], sp[o + 1], sp[o + 2]};
  };
  auto apply = [&](const Px& q, bool add) __attribute__((always_inline)) {
    const unsigned I0 = q.I0, I1 = q.I1, I2 = q.I2, P0 = q.P0, P1 = q.P1, P2 = q.P2;
#define GFM(a_, b_) ((unsigned)__umul24(a_, b_))
    const unsigned m[21] = {I0, I1, I2, P0, P1, P2, GFM(I0, I0), GFM(I0, I1), GFM(I0, I2), GFM(I1, I1), GFM(I1, I2), GFM(I2, I2),
                            GFM(I0, P0), GFM(I1, P0), GFM(I2, P0), GFM(I0, P1), GFM(I1, P1), GFM(I2, P1), GFM(I0, P2), GFM(I1, P2), GFM(I2, P2)};
#undef GFM
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) cs[pl] = add ? cs[pl] + m[pl] : cs[pl] - m[pl];
  };
  for (int k = -RR; k <= RR; ++k) apply(fetch(y0 + k), true);
  const int y1 = min(y0 + VS, H);
  const int ipl = t % 21, iseg = t / 21;                                 // horizontal item of this thread: (plane, 16-output segment)
  const bool pix_ok = t < GS_WC && x0 + t < W;
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  float res[12];                                                         // rounding residuals carried down the column
#pragma unroll
  for (int i = 0; i < 12; ++i) res[i] = 0.f;
  unsigned short* op = ab + (int64_t)b * 12 * P + x0 + t;
  for (int y = y0; y < y1; ++y) {
    // the bytes of the entering / leaving rows are requested here, ahead of the two LDS phases (a barrier is a fence: hipcc does not
    // move the loads up by itself), and used at the bottom of the iteration.  (Requested a whole iteration earlier still -- two sets in
    // flight -- the kernel was SLOWER, 0.36 -> 0.44 ms per batch: the extra registers cost a resident block per CU, and three co-resident
    // blocks hide the latency better than a deeper prefetch in two.)
    const Px pin = fetch(min(y + 1 + RR, H - 1 + RR)), pout = fetch(y - RR);
#pragma unroll
    for (int pl = 0; pl < 21; ++pl) col[pl][GF_ROW(t)] = cs[pl];
    __syncthreads();
    if (iseg < GS_NSEG) {
      const unsigned* r = col[ipl];
      unsigned* wv = win[ipl];
      const int j0 = iseg * GF_SEG;
      unsigned sum = 0;
      for (int k = 0; k <= 2 * RR; ++k) sum += r[GF_ROW(j0 + k)];
      wv[GF_ROW(j0)] = sum;
#pragma unroll
      for (int j = 1; j < GF_SEG; ++j) {
        sum += r[GF_ROW(j0 + j + 2 * RR)] - r[GF_ROW(j0 + j - 1)];
        wv[GF_ROW(j0 + j)] = sum;
      }
    }
    __syncthreads();
    if (pix_ok) {
      float m[21];
#pragma unroll
      for (int pl = 0; pl < 21; ++pl) m[pl] = (float)win[pl][GF_ROW(t)] * inv;
      const float mI0 = m[0], mI1 = m[1], mI2 = m[2];
      const float a00 = m[6] - mI0 * mI0 + eps, a01 = m[7] - mI0 * mI1, a02 = m[8] - mI0 * mI2;
      const float a11 = m[9] - mI1 * mI1 + eps, a12 = m[10] - mI1 * mI2, a22 = m[11] - mI2 * mI2 + eps;
      const float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
      const float c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12, c22 = a00 * a11 - a01 * a01;
      const float det = a00 * c00 + a01 * c01 + a02 * c02;
      const float rdet = 1.f / det;          // (one division: the three-launch form's six differ from it in the last bit)
      const float i00 = c00 * rdet, i01 = c01 * rdet, i02 = c02 * rdet, i11 = c11 * rdet, i12 = c12 * rdet, i22 = c22 * rdet;
      float v[12];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float mp_c = m[3 + c];
        const float cp0 = m[12 + c * 3 + 0] - mI0 * mp_c, cp1 = m[12 + c * 3 + 1] - mI1 * mp_c, cp2 = m[12 + c * 3 + 2] - mI2 * mp_c;
        v[c * 4 + 0] = i00 * cp0 + i01 * cp1 + i02 * cp2;
        v[c * 4 + 1] = i01 * cp0 + i11 * cp1 + i12 * cp2;
        v[c * 4 + 2] = i02 * cp0 + i12 * cp1 + i22 * cp2;
        v[c * 4 + 3] = mp_c - v[c * 4 + 0] * mI0 - v[c * 4 + 1] * mI1 - v[c * 4 + 2] * mI2;
      }
      const int64_t po = (int64_t)y * W;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float want = v[i] + res[i];
        const _Float16 h = (_Float16)want;
        res[i] = want - (float)h;
        op[i * P + po] = __builtin_bit_cast(unsigned short, h);
      }
    }
    if (y + 1 < y1) {
      apply(pin, true);
      apply(pout, false);
    }
  }
}

template <int RR, int VS>
__global__ __launch_bounds__(256) void gf_s2_fused_kernel(const unsigned short* __restrict__ ab, const unsigned char* __restrict__ guide,
                                                          float* __restrict__ out, unsigned char* __restrict__ out_u8, int H, int W) {
  __shared__ float col[12][GF_ROW(256) + 1];
  __shared__ float win[12][GF_ROW(GS_WC) + 1];
  const int t = threadIdx.x, x0 = blockIdx.x * GS_WC, y0 = blockIdx.y * VS, b = blockIdx.z;
  const int64_t P = (int64_t)H * W;
  const int xc = reflect_idx(min(x0 - RR + t, W - 1 + RR), W);
  const unsigned short* in = ab + (int64_t)b * 12 * P + xc;
  auto ld = [&](int i, int64_t ro) -> float { return (float)__builtin_bit_cast(_Float16, in[i * P + ro]); };
  float s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0.f;
  for (int k = -RR; k <= RR; ++k) {
    const int64_t ro = (int64_t)reflect_idx(y0 + k, H) * W;
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] += ld(i, ro);
  }
  const int y1 = min(y0 + VS, H);
  const int ipl = t % 12, iseg = t / 12;
  const bool pix_ok = t < GS_WC && x0 + t < W;
  const float inv = 1.f / (float)((2 * RR + 1) * (2 * RR + 1));
  for (int y = y0; y < y1; ++y) {
    unsigned short din[12], dout[12];      // entering / leaving row (raw halves), requested ahead of the two LDS phases
    {
      const int64_t rin = (int64_t)reflect_idx(min(y + 1 + RR, H - 1 + RR), H) * W, rout = (int64_t)reflect_idx(y - RR, H) * W;
#pragma unroll
      for (int i = 0; i < 12; ++i) { din[i] = in[i * P + rin]; dout[i] = in[i * P + rout]; }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) col[i][GF_ROW(t)] = s[i];
    __syncthreads();
    if (iseg < GS_NSEG) {
      const float* r = col[ipl];
      float* wv = win[ipl];
      const int j0 = iseg * GF_SEG;
      float sum = 0.f;
      for (int k = 0; k <= 2 * RR; ++k) sum += r[GF_ROW(j0 + k)];
      wv[GF_ROW(j0)] = sum;
#pragma unroll
      for (int j = 1; j < GF_SEG; ++j) {
        sum += r[GF_ROW(j0 + j + 2 * RR)] - r[GF_ROW(j0 + j - 1)];
        wv[GF_ROW(j0 + j)] = sum;
      }
    }
    __syncthreads();
    if (pix_ok) {
      const int x = x0 + t;
      const unsigned char* g = guide + (((int64_t)b * H + y) * W + x) * 3;
      const float I0 = g[0], I1 = g[1], I2 = g[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float qv = (win[c * 4][GF_ROW(t)] * inv) * I0 + (win[c * 4 + 1][GF_ROW(t)] * inv) * I1 + (win[c * 4 + 2][GF_ROW(t)] * inv) * I2 +
                         win[c * 4 + 3][GF_ROW(t)] * inv;
        const float rq = fminf(fmaxf(rintf(qv), 0.f), 255.f);
        if (out_u8) out_u8[(((int64_t)b * H + y) * W + x) * 3 + c] = (unsigned char)rq;
        if (out) out[((int64_t)b * 3 + c) * P + (int64_t)y * W + x] = (rq / 255.0f - 0.5f) * 2.f;  // ToTensor, (x-0.5)*2 (ppst_model.py:301-303)
      }
    }
    if (y + 1 < y1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] += (float)__builtin_bit_cast(_Float16, din[i]) - (float)__builtin_bit_cast(_Float16, dout[i]);
    }
  }
}

static int g_gf_vs1 = 0, g_gf_vs2 = 0;      // tuning aid: rows per block of the two fused launches (0 = the rule below; 32 / 64 / 128 = forced)
extern "C" int ppst_guided_filter_tune(int vs1, int vs2) { g_gf_vs1 = vs1; g_gf_vs2 = vs2; return PPST_OK; }

extern "C" int64_t ppst_guided_filter_ws(int B, int H, int W) { return (int64_t)B * 42 * H * W * (int64_t)sizeof(float); }

extern "C" int ppst_guided_filter(const void* guide_u8, const void* src_u8, void* out, void* out_u8, int B, int H, int W, int r,
                                  float eps, void* work, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || W > GF_MAXW || r <= 0 || r > GF_MAXR || r >= H || r >= W) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!guide_u8 || !src_u8 || !work || (!out && !out_u8)) return PPST_ENULL;
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  float* bufA = (float*)work;              // [B][21][P]
  float* bufB = bufA + (int64_t)B * 21 * P;  // [B][21][P]
  const unsigned char* g = (const unsigned char*)guide_u8;
  const unsigned char* s = (const unsigned char*)src_u8;
  auto blocks_for = [](int64_t total) { int64_t b = cdiv64(total, 256); return (unsigned)(b > 256 * 32 ? 256 * 32 : b); };
  int e;
#ifndef GF_THREE_LAUNCH   // (round 4's three-launch form: kept for A/B behind -DGF_THREE_LAUNCH)
  if (r == 30) {       // the path's radius (photo_gif.py:43): two fused launches, box sums kept on the chip (round 5)
    unsigned short* abh = (unsigned short*)work;            // [B][12][P] IEEE half
    // rows per block: 32 in the first launch (its halo re-reads are uint8 rows: 6 B / pixel each), 64 in the second (24 B / pixel each).
    // Measured, batch of four 1024^2 images (tests/gf_prof.sh): (64, 64) 0.36 ms, (32, 64) 0.33 ms, (32, 32) 0.30 ms with 1.25x the traffic,
    // (64, 128) 0.44 ms -- three co-resident blocks per CU (768 blocks) are what hides the per-row latency.
    const int vs1 = g_gf_vs1 ? g_gf_vs1 : 32;
    if (vs1 == 32) PPST_LAUNCH((gf_s1_fused_kernel<30, 32>), dim3(cdiv(W, GS_WC), cdiv(H, 32), B), dim3(256), 0, st, g, s, abh, H, W, eps);
    else PPST_LAUNCH((gf_s1_fused_kernel<30, 64>), dim3(cdiv(W, GS_WC), cdiv(H, 64), B), dim3(256), 0, st, g, s, abh, H, W, eps);
    if ((e = PPST_LAUNCH_CHECK())) return e;
    const int vs2 = g_gf_vs2 ? g_gf_vs2 : 64;
    if (vs2 == 128)
      PPST_LAUNCH((gf_s2_fused_kernel<30, 128>), dim3(cdiv(W, GS_WC), cdiv(H, 128), B), dim3(256), 0, st, (const unsigned short*)abh, g,
                  (float*)out, (unsigned char*)out_u8, H, W);
    else if (vs2 == 32)
      PPST_LAUNCH((gf_s2_fused_kernel<30, 32>), dim3(cdiv(W, GS_WC), cdiv(H, 32), B), dim3(256), 0, st, (const unsigned short*)abh, g,
                  (float*)out, (unsigned char*)out_u8, H, W);
    else
      PPST_LAUNCH((gf_s2_fused_kernel<30, 64>), dim3(cdiv(W, GS_WC), cdiv(H, 64), B), dim3(256), 0, st, (const unsigned short*)abh, g,
                  (float*)out, (unsigned char*)out_u8, H, W);
    return PPST_LAUNCH_CHECK();
  }
#endif
  if (r == 30) {       // the path's radius (photo_gif.py:43): sliding-window passes (round 4)
#ifdef GF_STAGE1_HV      // (the first sliding form: H pass over the moments, then V pass + solve; kept for A/B)
    PPST_LAUNCH(gf_h1_slide_kernel<30>, dim3(H, cdiv(W, GF_HCW), B), dim3(256), 0, st, g, s, bufA, H, W);
    if ((e = PPST_LAUNCH_CHECK())) return e;
    PPST_LAUNCH(gf_v1_solve_slide_kernel<30>, dim3(cdiv(W, 256), cdiv(H, GF_VSEG), B), dim3(256), 0, st, (const float*)bufA, bufB, H, W, eps);
    if ((e = PPST_LAUNCH_CHECK())) return e;
#else
    PPST_LAUNCH(gf_v1m_slide_kernel<30>, dim3(cdiv(W, 256), cdiv(H, GF_VSEG), B), dim3(256), 0, st, g, s, bufA, H, W);
    if ((e = PPST_LAUNCH_CHECK())) return e;
#ifdef GF_STAGE1_SPLIT   // (the solve and the stage-2 H pass as two launches; kept for A/B)
    PPST_LAUNCH(gf_h1s_solve_kernel<30>, dim3(H, cdiv(W, GF_HCW), B), dim3(256), 0, st, (const float*)bufA, bufB, H, W, eps);
    if ((e = PPST_LAUNCH_CHECK())) return e;
#else
    PPST_LAUNCH(gf_h1s_solve_h2_kernel<30>, dim3(H, cdiv(W, GF_HCW), B), dim3(256), 0, st, (const float*)bufA, bufB, H, W, eps);
    if ((e = PPST_LAUNCH_CHECK())) return e;
    PPST_LAUNCH(gf_v2_final_slide_kernel<30>, dim3(cdiv(W, 256), cdiv(H, GF_VSEG), B), dim3(256), 0, st, (const float*)bufB, g, (float*)out,
                (unsigned char*)out_u8, H, W);
    return PPST_LAUNCH_CHECK();
#endif
#endif
    PPST_LAUNCH(gf_h2_slide_kernel<30>, dim3(H, cdiv(W, GF_HCW), B), dim3(128), 0, st, (const float*)bufB, bufA, H, W);
    if ((e = PPST_LAUNCH_CHECK())) return e;
    PPST_LAUNCH(gf_v2_final_slide_kernel<30>, dim3(cdiv(W, 256), cdiv(H, GF_VSEG), B), dim3(256), 0, st, (const float*)bufA, g, (float*)out,
                (unsigned char*)out_u8, H, W);
    return PPST_LAUNCH_CHECK();
  }
  PPST_LAUNCH(gf_h_kernel<true>, dim3(H, 21, B), dim3(256), 0, st, g, s, (const float*)nullptr, bufA, H, W, r, 21);
  if ((e = PPST_LAUNCH_CHECK())) return e;
  int64_t t21 = (int64_t)B * 21 * P;
  const int ytiles = cdiv(H, GV_R);
  const int64_t tv = (int64_t)B * 21 * ytiles * W;
  if (r == 30 && tv <= PPST_IDX32_MAX)     // the path's radius (photo_gif.py:43): register-blocked, bit-identical to the plain form
    PPST_LAUNCH(gf_v_blocked_kernel<30>, dim3(blocks_for(tv)), dim3(256), 0, st, (const float*)bufA, bufB, H, W, ytiles, (unsigned)tv,
                make_fastdiv((unsigned)W), make_fastdiv((unsigned)ytiles));
  else
    PPST_LAUNCH(gf_v_kernel, dim3(blocks_for(t21)), dim3(256), 0, st, (const float*)bufA, bufB, H, W, r, t21);
  if ((e = PPST_LAUNCH_CHECK())) return e;
  PPST_LAUNCH(gf_solve_kernel, dim3(blocks_for(B * P)), dim3(256), 0, st, (const float*)bufB, bufA, P, eps, (int64_t)B * P);
  if ((e = PPST_LAUNCH_CHECK())) return e;
  PPST_LAUNCH(gf_h_kernel<false>, dim3(H, 12, B), dim3(256), 0, st, g, s, (const float*)bufA, bufB, H, W, r, 12);
  if ((e = PPST_LAUNCH_CHECK())) return e;
  // (the same register blocking of this pass -- 12 planes x 8 rows per thread -- measured 3.5x SLOWER: 262 k threads with
  //  ~200 live registers each leave the chip empty; the plain form keeps one thread per pixel)
  PPST_LAUNCH(gf_v_final_kernel, dim3(blocks_for(B * P)), dim3(256), 0, st, (const float*)bufB, g, (float*)out,
                     (unsigned char*)out_u8, H, W, r, (int64_t)B * P);
  return PPST_LAUNCH_CHECK();
}

// 2 (round 5): ppst_pack_job gained a trailing `dual` field and ppst_conv_args `dual_b` / `io_st` / `k64` in round 4 -- the ARRAY stride of
// the job table changed, which "appended" does not cover; callers built against the round-3 header must rebuild.
extern "C" int ppst_version(void) { return 2; }
