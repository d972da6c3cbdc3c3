// Shared by rscl_rows_kernel (train.hip) and rscl_rows_bwd_kernel (train_g.hip): the logits of one query row of rsclLoss
// (networks/rscl.py:42-64) in shared memory, their maximum and the softmax denominator.
#pragma once
#include "common.h"

#define RS_T 1024
struct RsclShared {
  float part[4][512];      // partial dot products per quarter of the C reduction
  float prob[512];         // logits / T of the negatives [queue | k0], then exp(logit - max)
  float red[RS_T / 64];
  float bc;
};

__device__ __forceinline__ float rs_block_sum(float v, RsclShared& sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < RS_T / 64; ++w) r += sh.red[w];
  return r;
}
__device__ __forceinline__ float rs_block_max(float v, RsclShared& sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh.red[0];
#pragma unroll
  for (int w = 1; w < RS_T / 64; ++w) r = fmaxf(r, sh.red[w]);
  return r;
}

// After the call: sh.prob[j] = exp(logit_j - m) for j < K + n0 (every thread may read it), s_pos = positive logit,
// m = max over all logits (incl. the n masked entries at -10 / T), ssum = sum of exp(. - m) over all of them.
__device__ __forceinline__ void rscl_logits(const float* __restrict__ qi, const float* __restrict__ ki, const float* __restrict__ k0,
                                            const float* __restrict__ queue, int n, int n0, int C, int K, float invT, RsclShared& sh,
                                            float& s_pos, float& m, float& ssum) {
  const int t = threadIdx.x;
  float p = 0.f;
  for (int c = t; c < C; c += RS_T) p += qi[c] * ki[c];
  s_pos = rs_block_sum(p, sh) * invT;
  const int nneg = K + n0;
  const int qd = t >> 8, cq = (C + 3) >> 2;
  const int c0 = qd * cq, c1 = (c0 + cq < C) ? c0 + cq : C;
  for (int jb = 0; jb < nneg; jb += 256) {
    const int j = jb + (t & 255);
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (j < K) {
      const float* col = queue + j;               // queue is [C][K]: a warp reads 64 consecutive columns of one row
      int c = c0;
      for (; c + 3 < c1; c += 4) {
        d0 += qi[c] * col[(int64_t)c * K];
        d1 += qi[c + 1] * col[(int64_t)(c + 1) * K];
        d2 += qi[c + 2] * col[(int64_t)(c + 2) * K];
        d3 += qi[c + 3] * col[(int64_t)(c + 3) * K];
      }
      for (; c < c1; ++c) d0 += qi[c] * col[(int64_t)c * K];
    } else if (j < nneg) {
      const float* kj = k0 + (int64_t)(j - K) * C;
      int c = c0;
      for (; c + 3 < c1; c += 4) {
        d0 += qi[c] * kj[c]; d1 += qi[c + 1] * kj[c + 1]; d2 += qi[c + 2] * kj[c + 2]; d3 += qi[c + 3] * kj[c + 3];
      }
      for (; c < c1; ++c) d0 += qi[c] * kj[c];
    }
    if (j < nneg) sh.part[qd][j] = (d0 + d1) + (d2 + d3);
  }
  __syncthreads();
  float lmax = fmaxf(s_pos, -10.0f * invT);
  if (t < nneg) {
    const float l = ((sh.part[0][t] + sh.part[1][t]) + (sh.part[2][t] + sh.part[3][t])) * invT;
    sh.prob[t] = l;
    lmax = fmaxf(lmax, l);
  }
  m = rs_block_max(lmax, sh);
  float e = 0.f;
  if (t < nneg) {
    e = expf(sh.prob[t] - m);
    sh.prob[t] = e;
  }
  if (t == 0) e += expf(s_pos - m) + (float)n * expf(-10.0f * invT - m);
  ssum = rs_block_sum(e, sh);     // (its barriers also publish sh.prob)
}
