// Semantic-correspondence kernels (models/ppst_model.py:330-387):
//   Rselfcorr  -> ppst_rselfcorr      (no 268 MB (B,64,4096,16,16) intermediate)
//   corrm      -> ppst_corr_prep (centre + L2 normalise) + ppst_gemm_nt_split (6 passes) + ppst_softmax_rows
//   warp / E2.warp -> ppst_gemm_nn_split (3 passes) (+ unfold/fold plumbing)
// The logits are cosine similarities divided by T = 0.01, so a bf16-class error would move softmax outputs by ~1e-2
// relative: they are computed from three bf16 planes per operand (all of fp32's mantissa, six MFMA passes); rounds 1-2 used the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak), which stays for precision 2 and K % 16 != 0 (ppst_gemm_*_f32).
#include "common.h"

// ---------------------------------------------------------------- Rselfcorr --
// fea NHWC [B][H][W][C=64]; one wave per 4x4 patch.  lane = channel while loading,
// centring and normalising (wave reductions over the 64 channels); the normalised patch
// goes to LDS [c][16] and lane (i = l>>2, j = 4*(l&3)..+3) accumulates 4 Gram entries
// over c.  out NHWC [B][H/4][W/4][256], channel = i*16 + j, written as one 1-KB row per wave.
__global__ __launch_bounds__(256) void rselfcorr_kernel(const float* __restrict__ fea, float* __restrict__ out, int B, int H, int W,
                                                        int out_ld, float eps, int64_t npatch) {
  __shared__ __attribute__((aligned(16))) float xs[4][64][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gy = H >> 2, gx = W >> 2;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wv; p < npatch; p += (int64_t)gridDim.x * 4) {
    int px = (int)(p % gx);
    int64_t r = p / gx;
    int py = (int)(r % gy);
    int b = (int)(r / gy);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int iy = py * 4 + (i >> 2), ix = px * 4 + (i & 3);
      v[i] = fea[(((int64_t)b * H + iy) * W + ix) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float m = wave_sum(v[i]) * (1.f / 64.f);
      float d = v[i] - m;
      float nrm = sqrtf(wave_sum(d * d)) + eps;
      v[i] = d / nrm;
    }
    float4* row = (float4*)&xs[wv][lane][0];
    row[0] = make_float4(v[0], v[1], v[2], v[3]);
    row[1] = make_float4(v[4], v[5], v[6], v[7]);
    row[2] = make_float4(v[8], v[9], v[10], v[11]);
    row[3] = make_float4(v[12], v[13], v[14], v[15]);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes (no other wave touches xs[wv])
    __builtin_amdgcn_wave_barrier();
    const int i = lane >> 2, j4 = (lane & 3) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
      float xi = xs[wv][c][i];
      float4 xj = *(const float4*)&xs[wv][c][j4];
      acc.x += xi * xj.x; acc.y += xi * xj.y; acc.z += xi * xj.z; acc.w += xi * xj.w;
    }
    *(float4*)(out + (((int64_t)b * gy + py) * gx + px) * out_ld + i * 16 + j4) = acc;
    __builtin_amdgcn_wave_barrier();
  }
}
extern "C" int ppst_rselfcorr(const void* fea, void* out, int B, int H, int W, int C, int out_ld, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || H % 4 || W % 4 || C != 64 || out_ld < 256 || out_ld % 4) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!fea || !out) return PPST_ENULL;
  int64_t npatch = (int64_t)B * (H / 4) * (W / 4);
  int64_t blocks = cdiv64(npatch, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  PPST_LAUNCH(rselfcorr_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)fea, (float*)out, B, H,
                     W, out_ld, 2.220446049250313e-16f, npatch);
  return PPST_LAUNCH_CHECK();
}

// ---------------------------------------------------------------- corr prep --
// rows [B*P][C]: mean-centre the first `ncenter` channels, L2-normalise all C (+eps).
// One wave per row, C <= 1024, C % 64 == 0.
__global__ __launch_bounds__(256) void corr_prep_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int C,
                                                        int ncenter, float eps) {
  const int lane = threadIdx.x & 63;
  const int per = C >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const float* xr = x + r * C;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < per) {
        int c = i * 64 + lane;
        v[i] = xr[c];
        if (c < ncenter) s += v[i];
      }
    float mean = ncenter > 0 ? wave_sum(s) / (float)ncenter : 0.f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < per) {
        int c = i * 64 + lane;
        if (c < ncenter) v[i] -= mean;
        q += v[i] * v[i];
      }
    float nrm = sqrtf(wave_sum(q)) + eps;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < per) y[r * C + i * 64 + lane] = v[i] / nrm;
  }
}
// the same for any C (match_kernel != 1: rows of 512 k^2 unfolded values): one wave per row, the row is re-read per pass
__global__ __launch_bounds__(256) void corr_prep_loop_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int C,
                                                             int ncenter, float eps) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const float* xr = x + r * C;
    float s = 0.f;
    for (int c = lane; c < ncenter; c += 64) s += xr[c];
    const float mean = ncenter > 0 ? wave_sum(s) / (float)ncenter : 0.f;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float z = xr[c] - (c < ncenter ? mean : 0.f);
      q += z * z;
    }
    const float nrm = sqrtf(wave_sum(q)) + eps;
    for (int c = lane; c < C; c += 64) y[r * C + c] = (xr[c] - (c < ncenter ? mean : 0.f)) / nrm;
  }
}
extern "C" int ppst_corr_prep(const void* fea, void* out, int B, int P, int C, int ncenter, void* stream) {
  if (B < 0 || P <= 0 || C <= 0 || ncenter < 0 || ncenter > C) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!fea || !out) return PPST_ENULL;
  int64_t rows = (int64_t)B * P;
  int64_t blocks = cdiv64(rows, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (C % 64 || C > 1024) {
    PPST_LAUNCH(corr_prep_loop_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)fea, (float*)out, rows, C,
                ncenter, 2.220446049250313e-16f);
    return PPST_LAUNCH_CHECK();
  }
  PPST_LAUNCH(corr_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)fea, (float*)out, rows, C,
                     ncenter, 2.220446049250313e-16f);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------- F.unfold rows (match_kernel != 1) --
// corrm with opt.match_kernel = k (ppst_model.py:345-347): F.unfold(fea, k, padding = k / 2) of an NHWC map, written as the
// GEMM's rows: out[b][p][c k^2 + ky k + kx] = x[b][y + ky - r][x + kx - r][c] (zero outside), p = y W + x, r = k / 2 (k odd).
// One block per output row; the k^2 x C neighbourhood it gathers stays in L1.  Backward = the gather in the other direction.
__global__ __launch_bounds__(256) void unfold_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C, int k,
                                                          int64_t rows) {
  const int kk = k * k, r = k >> 1, K = C * kk;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int p = (int)(row % ((int64_t)H * W));
    const int64_t b = row / ((int64_t)H * W);
    const int y = p / W, xx = p - y * W;
    for (int j = threadIdx.x; j < K; j += 256) {
      const int c = j / kk, t = j - c * kk, ky = t / k, kx = t - ky * k;
      const int iy = y + ky - r, ix = xx + kx - r;
      out[row * K + j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((b * H + iy) * W + ix) * C + c] : 0.f;
    }
  }
}
__global__ __launch_bounds__(256) void unfold_rows_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int H, int W, int C,
                                                              int k, int64_t total) {
  const int kk = k * k, r = k >> 1;
  const int64_t K = (int64_t)C * kk;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int xx = (int)(q % W); q /= W;
    const int y = (int)(q % H);
    const int64_t b = q / H;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int py = y - ky + r;
      if (py < 0 || py >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int px = xx - kx + r;
        if (px < 0 || px >= W) continue;
        acc += g[((b * H + py) * W + px) * K + (int64_t)c * kk + ky * k + kx];
      }
    }
    dx[i] = acc;
  }
}
extern "C" int ppst_unfold_rows(const void* x, void* out, int B, int H, int W, int C, int k, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k % 2 == 0 || (int64_t)C * k * k > 0x7fffffffll) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !out) return PPST_ENULL;
  const int64_t rows = (int64_t)B * H * W;
  PPST_LAUNCH(unfold_rows_kernel, dim3((unsigned)(rows > 65536 ? 65536 : rows)), dim3(256), 0, as_stream(stream), (const float*)x,
              (float*)out, H, W, C, k, rows);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_unfold_rows_bwd(const void* g, void* dx, int B, int H, int W, int C, int k, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k % 2 == 0 || (int64_t)C * k * k > 0x7fffffffll) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!g || !dx) return PPST_ENULL;
  const int64_t total = (int64_t)B * H * W * C;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 65536) blocks = 65536;
  PPST_LAUNCH(unfold_rows_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)g, (float*)dx, H, W, C, k,
              total);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------------------ fp32 MFMA GEMM --
// C[b] = alpha * A[b] (M x K, row-major lda) * op(B[b]); B_NT: B is N x K row-major (ldb)
// else K x N row-major (ldb).  Block 128 x (32*NTL) x 16, 4 waves (32 rows each), register
// prefetch + double-buffered LDS (one barrier per 16-deep K tile).
template <int NTL, bool B_NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                       int M, int N, int K, int lda, int ldb, int ldc, int64_t sA, int64_t sB,
                                                       int64_t sC, float alpha) {
  constexpr int BM = 128, BN = 32 * NTL, BK = 16;
  constexpr int LDA_S = BK + 1;
  constexpr int LDB_S = B_NT ? BK + 1 : BN + 1;
  constexpr int BS_SIZE = B_NT ? BN * LDB_S : BK * LDB_S;
  constexpr int B_F4 = BN * BK / 4;
  constexpr int B_IT = (B_F4 + 255) / 256;
  __shared__ float As[2][BM * LDA_S];
  __shared__ float Bs[2][BS_SIZE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  A += (int64_t)blockIdx.z * sA;
  Bm += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  float4 ra[2], rb[B_IT];
  auto g_load = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int idx = tid + it * 256, row = idx >> 2, c4 = idx & 3;
      ra[it] = (m0 + row < M) ? *(const float4*)(A + (int64_t)(m0 + row) * lda + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      int idx = tid + it * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        if (B_NT) {
          int row = idx >> 2, c4 = idx & 3;
          if (n0 + row < N) v = *(const float4*)(Bm + (int64_t)(n0 + row) * ldb + k0 + c4 * 4);
        } else {
          int k = idx / (BN / 4), n4 = idx - k * (BN / 4);
          if (n0 + n4 * 4 < N) v = *(const float4*)(Bm + (int64_t)(k0 + k) * ldb + n0 + n4 * 4);
        }
      }
      rb[it] = v;
    }
  };
  auto s_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int idx = tid + it * 256, row = idx >> 2, c4 = idx & 3;
      float* d = &As[buf][row * LDA_S + c4 * 4];
      d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      int idx = tid + it * 256;
      if (idx < B_F4) {
        float* d;
        if (B_NT) { int row = idx >> 2, c4 = idx & 3; d = &Bs[buf][row * LDB_S + c4 * 4]; }
        else { int k = idx / (BN / 4), n4 = idx - k * (BN / 4); d = &Bs[buf][k * LDB_S + n4 * 4]; }
        d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w;
      }
    }
  };
  f32x16 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  g_load(0);
  s_store(0);
  __syncthreads();
  const int nk = K / BK;
  const int li = lane & 31, lk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) g_load((kt + 1) * BK);
    const float* as = &As[buf][(wv * 32 + li) * LDA_S];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av = as[kk + lk];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        float bv = B_NT ? Bs[buf][(t * 32 + li) * LDB_S + kk + lk] : Bs[buf][(kk + lk) * LDB_S + t * 32 + li];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) s_store(buf ^ 1);
    __syncthreads();
  }
  // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    int n = n0 + t * 32 + li;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      int m = m0 + wv * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lk;
      if (m < M && n < N) C[(int64_t)m * ldc + n] = acc[t][rg] * alpha;
    }
  }
}

template <bool B_NT>
static int gemm_dispatch(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc,
                         int64_t sA, int64_t sB, int64_t sC, float alpha, hipStream_t st) {
  int ntl = N >= 160 ? 5 : (N >= 128 ? 4 : (N + 31) / 32);
  if (N % 160 != 0 && N % 128 == 0) ntl = 4;
  dim3 grid(cdiv(M, 128), cdiv(N, 32 * ntl), batch);
#define GL(NTL) PPST_LAUNCH((gemm_f32_kernel<NTL, B_NT>), grid, dim3(256), 0, st, A, Bm, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha)
  switch (ntl) {
    case 1: GL(1); break;
    case 2: GL(2); break;
    case 3: GL(3); break;
    case 4: GL(4); break;
    default: GL(5); break;
  }
#undef GL
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_gemm_nt_f32(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, float alpha, void* stream) {
  if (batch < 0 || M <= 0 || N <= 0 || K <= 0 || K % 16) return PPST_EINVAL;
  if (batch == 0) return PPST_OK;
  if (!A || !Bm || !C) return PPST_ENULL;
  return gemm_dispatch<true>((const float*)A, (const float*)Bm, (float*)C, batch, M, N, K, K, K, N, (int64_t)M * K, (int64_t)N * K,
                             (int64_t)M * N, alpha, as_stream(stream));
}
extern "C" int ppst_gemm_nn_f32(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int ldb, int ldc, void* stream) {
  if (batch < 0 || M <= 0 || N <= 0 || K <= 0 || K % 16 || N % 4 || ldb % 4 || ldb < N || ldc < N) return PPST_EINVAL;
  if (batch == 0) return PPST_OK;
  if (!A || !Bm || !C) return PPST_ENULL;
  return gemm_dispatch<false>((const float*)A, (const float*)Bm, (float*)C, batch, M, N, K, K, ldb, ldc, (int64_t)M * K,
                              (int64_t)K * ldb, (int64_t)M * ldc, 1.f, as_stream(stream));
}

// ------------------------------------------------ split-bf16 MFMA GEMM (fp32 in / out) --
// The same products on the bf16 matrix pipe (2.5 PFLOP/s dense against the fp32 MFMA's 157 TFLOP/s):
//   NP = 3 planes, 6 passes: x = h + m + l (8 + 8 + 8 mantissa bits = all of fp32); h.h + h.m + m.h + m.m + h.l + l.h, the
//     dropped terms (m.l, l.m, l.l) are <= 2^-24 relative each: fp32-class results (the cosine logits, which the softmax
//     multiplies by 1 / T = 100), ceiling 2500 / 6 = 417 TFLOP/s;
//   NP = 2 planes, 3 passes: x = h + l, h.h + h.l + l.h (the convs' bf16x3: ~2^-16 relative per product), ceiling 833 TFLOP/s:
//     the products of softmax rows with feature / gradient matrices.
// WM x WN waves, each TM x TN tiles of v_mfma_f32_32x32x16_bf16; register prefetch TWO K tiles ahead, double-buffered LDS, one
// barrier per K tile.  fp32 -> planes while staging.  A (and B when it is N x K) sit in LDS as [row][BK] bf16 with a 16-byte row
// pad (ds_read_b128 operands, conflict-free: row stride = 4 x odd dwords); a K x N matrix B sits as [32-column slab][k][32]
// bf16 and is read through ds_read_b64_tr_b16 (the k index of the operand is the ROW).
// Two shapes: 2 x 2 waves of 64 x 64 (block 128 x 128, two blocks per CU) and 2 x 4 waves of 128 x 64 (block 256 x 256, one
// block per CU).  Timing ablations of the 128 x 128 form on q.k^T (8 x 4096 x 4096 x 512, 3 passes): everything 0.50 ms, without
// the global loads 0.30, staging alone (no MFMA) 0.36 -- its fp32 operand tiles ask the L2 for 4.3 GB per launch (12 TB/s);
// the 256 x 256 block asks for half.
template <bool B_NT, int NP, int BK, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void gemm_split_kernel(
    const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
    int64_t sA, int64_t sB, int64_t sC, float alpha) {
  constexpr int NTHR = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ROW = BK * 2 + 16;                      // bytes per padded row of one plane
  constexpr int PL_A = BM * ROW;
  constexpr int PL_B = B_NT ? BN * ROW : (BN / 32) * BK * 64;
  constexpr int STAGE = NP * (PL_A + PL_B);
  constexpr int FA = BM * BK / 4 / NTHR, FB = BN * BK / 4 / NTHR;      // float4 per thread and operand tile
  static_assert(FA * NTHR * 4 == BM * BK && FB * NTHR * 4 == BN * BK, "tile / thread count");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int li = lane & 31, kb = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  A += (int64_t)blockIdx.z * sA;
  Bm += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  // operand tiles through buffer loads: descriptor = this batch element's matrix, per-item byte offset fixed for the block
  // (-1 = row / column outside the matrix: the hardware returns zeros), the K-tile offset as SGPR soffset -- no per-load VALU
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((int64_t)M * lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs =
      __builtin_amdgcn_make_buffer_rsrc((void*)Bm, 0, (int)((int64_t)(B_NT ? N : K) * ldb * 4), 0x00020000);
  int aoff[FA], boff[FB];
#pragma unroll
  for (int it = 0; it < FA; ++it) {
    const int idx = tid + it * NTHR, row = idx / (BK / 4), c4 = idx % (BK / 4);
    aoff[it] = (m0 + row < M) ? ((m0 + row) * lda + c4 * 4) * 4 : -1;
  }
#pragma unroll
  for (int it = 0; it < FB; ++it) {
    const int idx = tid + it * NTHR;
    if (B_NT) {
      const int row = idx / (BK / 4), c4 = idx % (BK / 4);
      boff[it] = (n0 + row < N) ? ((n0 + row) * ldb + c4 * 4) * 4 : -1;
    } else {
      const int k = idx / (BN / 4), n4 = idx % (BN / 4);
      boff[it] = (n0 + n4 * 4 < N) ? (k * ldb + n0 + n4 * 4) * 4 : -1;
    }
  }
  // two register sets: the loads of K tile t + 2 are issued while tile t is multiplied and tile t + 1 (requested one
  // iteration earlier) waits in the other set -- a full iteration of MFMAs between a request and its first use
  float4 ra0[FA], rb0[FB], ra1[FA], rb1[FB];
  auto g_load = [&](float4 (&ra)[FA], float4 (&rb)[FB], int k0) {
#if defined(GEMM_ABL) && (GEMM_ABL & 4)
    if (kb >= 0) return;
#endif
    const int sa_off = k0 * 4, sb_off = B_NT ? k0 * 4 : k0 * ldb * 4;
#pragma unroll
    for (int it = 0; it < FA; ++it)
      ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff[it], sa_off, 0));
#pragma unroll
    for (int it = 0; it < FB; ++it)
      rb[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, boff[it], sb_off, 0));
  };
  // fp32 x 4 -> NP planes of bf16 x 4 (p[0] = h, then the successive remainders), two elements per v_cvt_pk_bf16_f32
  auto planes = [](float4 v, uint2* p) {
    typedef __bf16 __attribute__((ext_vector_type(2))) bf2;
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 r0 = {v.x, v.y}, r1 = {v.z, v.w};
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf2));
      const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
      p[pl] = make_uint2(q0, q1);
      if (pl + 1 < NP) {
        r0 -= (f2){__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
        r1 -= (f2){__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
      }
    }
  };
  auto s_store = [&](const float4 (&ra)[FA], const float4 (&rb)[FB], int buf) {
#if defined(GEMM_ABL) && (GEMM_ABL & 2)
    if (kb >= 0) return;
#endif
    unsigned char* const sa = smem + buf * STAGE;
    unsigned char* const sb = sa + NP * PL_A;
#pragma unroll
    for (int it = 0; it < FA; ++it) {
      const int idx = tid + it * NTHR, row = idx / (BK / 4), c4 = idx % (BK / 4);
      uint2 p[NP];
      planes(ra[it], p);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) *(uint2*)(sa + pl * PL_A + row * ROW + c4 * 8) = p[pl];
    }
#pragma unroll
    for (int it = 0; it < FB; ++it) {
      const int idx = tid + it * NTHR;
      uint2 p[NP];
      planes(rb[it], p);
      if (B_NT) {
        const int row = idx / (BK / 4), c4 = idx % (BK / 4);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) *(uint2*)(sb + pl * PL_B + row * ROW + c4 * 8) = p[pl];
      } else {
        const int k = idx / (BN / 4), n4 = idx % (BN / 4);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) *(uint2*)(sb + pl * PL_B + (n4 >> 3) * (BK * 64) + k * 64 + (n4 & 7) * 8) = p[pl];
      }
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // transposed-read lane geometry (train.hip, conv_wgrad_tr_kernel): lane -> row 8 kb + qd (+ 4), column group 16 (g & 1) + 4 pp
  const int g = lane >> 4, qd = (lane & 15) >> 2, pp = lane & 3;
  const int tr_off = (8 * (g >> 1) + qd) * 64 + (16 * (g & 1) + 4 * pp) * 2;
  auto compute = [&](int buf) {
#if defined(GEMM_ABL) && (GEMM_ABL & 1)
    if (kb >= 0) return;
#endif
    const unsigned char* const sa = smem + buf * STAGE;
    const unsigned char* const sb = sa + NP * PL_A;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM][NP], fb[TN][NP];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          fa[i][pl] = *(const bf16x8*)(sa + pl * PL_A + ((wm * TM + i) * 32 + li) * ROW + (ks * 16 + kb * 8) * 2);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          if (B_NT) {
            fb[j][pl] = *(const bf16x8*)(sb + pl * PL_B + ((wn * TN + j) * 32 + li) * ROW + (ks * 16 + kb * 8) * 2);
          } else {
            typedef short __attribute__((ext_vector_type(4))) v4s;
            const unsigned char* p0 = sb + pl * PL_B + (wn * TN + j) * (BK * 64) + ks * 16 * 64 + tr_off;
            const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p0));
            const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p0 + 4 * 64));
            fb[j][pl] = (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
          }
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          if (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][NP - 1], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NP - 1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };
  // Every request and every staging store below is unconditional (tiles past the end re-read the last tile; their stores go
  // to a buffer nobody reads again), and no branch sits between a request and its use: with a request under a branch hipcc's
  // wait-count pass has to assume it was NOT issued and turns the counted wait in front of the older set's conversion into
  // vmcnt(7..0) -- it waits for the youngest loads -- and LLVM sinks loads below a branch their users sit behind.
  const int nk = K / BK;
  auto tile_k0 = [&](int t) { return (t < nk ? t : nk - 1) * BK; };
  g_load(ra0, rb0, 0);
  s_store(ra0, rb0, 0);
  __syncthreads();
  g_load(ra0, rb0, tile_k0(1));
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    g_load(ra1, rb1, tile_k0(kt + 2));
    __builtin_amdgcn_sched_barrier(0);                      // (the scheduler would move the requests down to their use)
    compute(0);
    s_store(ra0, rb0, 1);
    __syncthreads();
    g_load(ra0, rb0, tile_k0(kt + 3));
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    s_store(ra1, rb1, 0);
    __syncthreads();
  }
  if (kt < nk) compute(0);                                  // odd tile count: the last tile sits in buffer 0
  // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int m = m0 + (wm * TM + i) * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kb;
        if (m < M && n < N) C[(int64_t)m * ldc + n] = acc[i][j][rg] * alpha;
      }
    }
}

// passes: 6 (fp32-class) | 3 (bf16x3).  K % 32 == 0 (3 passes) / K % 16 == 0 (6 passes).  The 256 x 256 block runs when its
// grid still gives every CU a block.
template <bool B_NT>
static int gemm_split_dispatch(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc,
                               int64_t sA, int64_t sB, int64_t sC, float alpha, int passes, hipStream_t st) {
  const bool big = (int64_t)cdiv(M, 256) * cdiv(N, 256) * batch >= 224 && N > 128 && M > 128;
#define GS(NP, BK, WM, WN, TM, TN)                                                                                                 \
  PPST_LAUNCH((gemm_split_kernel<B_NT, NP, BK, WM, WN, TM, TN>), dim3(cdiv(M, WM * TM * 32), cdiv(N, WN * TN * 32), batch),        \
              dim3(WM * WN * 64), 0, st, A, Bm, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha)
  if (passes == 6) {
    if (big) GS(3, 16, 2, 4, 4, 2); else GS(3, 16, 2, 2, 2, 2);
  } else {
    if (big) GS(2, 16, 2, 4, 4, 2); else GS(2, 32, 2, 2, 2, 2);
  }
#undef GS
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_gemm_nt_split(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, float alpha, int passes,
                                  void* stream) {
  if (batch < 0 || M <= 0 || N <= 0 || K <= 0 || (passes != 3 && passes != 6) || K % (passes == 6 ? 16 : 32)) return PPST_EINVAL;
  if (batch == 0) return PPST_OK;
  if (!A || !Bm || !C) return PPST_ENULL;
  if (((uintptr_t)A | (uintptr_t)Bm) % 16 || (int64_t)M * K * 4 > 0x7fffffffll || (int64_t)N * K * 4 > 0x7fffffffll) return PPST_EINVAL;
  return gemm_split_dispatch<true>((const float*)A, (const float*)Bm, (float*)C, batch, M, N, K, K, K, N, (int64_t)M * K,
                                   (int64_t)N * K, (int64_t)M * N, alpha, passes, as_stream(stream));
}
extern "C" int ppst_gemm_nn_split(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int ldb, int ldc, int passes,
                                  void* stream) {
  if (batch < 0 || M <= 0 || N <= 0 || K <= 0 || (passes != 3 && passes != 6) || K % (passes == 6 ? 16 : 32) || N % 4 || ldb % 4 ||
      ldb < N || ldc < N)
    return PPST_EINVAL;
  if (batch == 0) return PPST_OK;
  if (!A || !Bm || !C) return PPST_ENULL;
  if (((uintptr_t)A | (uintptr_t)Bm) % 16 || (int64_t)M * K * 4 > 0x7fffffffll || (int64_t)K * ldb * 4 > 0x7fffffffll) return PPST_EINVAL;
  return gemm_split_dispatch<false>((const float*)A, (const float*)Bm, (float*)C, batch, M, N, K, K, ldb, ldc, (int64_t)M * K,
                                    (int64_t)K * ldb, (int64_t)M * ldc, 1.f, passes, as_stream(stream));
}

// ------------------------------------------------------------------ softmax --
// in-place softmax(x / div) over rows of `cols` (<= 16384) floats; one block per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int cols, float div) {
  __shared__ float red[4];
  float* row = x + (int64_t)blockIdx.x * cols;
  constexpr int MAXV = 16;
  float4 v[MAXV];
  const int n4 = cols >> 2;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = threadIdx.x + i * 256;
    if (idx < n4) {
      float4 t = ((const float4*)row)[idx];
      t.x /= div; t.y /= div; t.z /= div; t.w /= div;
      v[i] = t;
      mx = fmaxf(mx, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = threadIdx.x + i * 256;
    if (idx < n4) {
      v[i].x = __expf(v[i].x - mx); v[i].y = __expf(v[i].y - mx); v[i].z = __expf(v[i].z - mx); v[i].w = __expf(v[i].w - mx);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = threadIdx.x + i * 256;
    if (idx < n4) ((float4*)row)[idx] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
  }
}
extern "C" int ppst_softmax_rows(void* x, int64_t rows, int cols, float div, void* stream) {
  if (rows < 0 || cols <= 0 || cols % 4 || cols > 16384 || div == 0.f || rows > 0x7fffffff) return PPST_EINVAL;
  if (rows == 0) return PPST_OK;
  if (!x) return PPST_ENULL;
  PPST_LAUNCH(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), (float*)x, cols, div);
  return PPST_LAUNCH_CHECK();
}

// ------------------------------------------------- unfold / fold (F.unfold) --
// x NCHW [B][C][H][W] <-> y [B][P][C*s*s], P = (H/s)*(W/s), column = c*s*s + ky*s + kx
template <bool FOLD>
__global__ __launch_bounds__(256) void patches_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int s,
                                                      int64_t total) {
  const int gx = W / s, css = C * s * s;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int col = (int)(t % css);
    int64_t r = t / css;
    int p = (int)(r % ((H / s) * gx));
    int b = (int)(r / ((H / s) * gx));
    int c = col / (s * s), k = col - c * s * s, ky = k / s, kx = k - ky * s;
    int py = p / gx, px = p - py * gx;
    int64_t img = (((int64_t)b * C + c) * H + py * s + ky) * W + px * s + kx;
    if (FOLD) dst[img] = src[t];
    else dst[t] = src[img];
  }
}
extern "C" int ppst_unfold_patches(const void* x, void* y, int B, int C, int H, int W, int s, void* stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || s <= 0 || H % s || W % s) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * C * H * W;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  PPST_LAUNCH(patches_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, C, H, W, s, total);
  return PPST_LAUNCH_CHECK();
}
extern "C" int ppst_fold_patches(const void* x, void* y, int B, int C, int H, int W, int s, void* stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || s <= 0 || H % s || W % s) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  if (!x || !y) return PPST_ENULL;
  int64_t total = (int64_t)B * C * H * W;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 4096) blocks = 4096;
  PPST_LAUNCH(patches_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, C, H, W, s, total);
  return PPST_LAUNCH_CHECK();
}
