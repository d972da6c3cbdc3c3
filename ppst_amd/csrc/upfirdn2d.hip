// upfirdn2d: zero-insert upsample -> pad/crop -> 2-D FIR (true convolution) -> decimate.
// Replaces models/networks/stylegan2_op/upfirdn2d_kernel.cu:52-137 (+dispatch :140-272).
// Tensor layout [major, H, W, minor] fp32 as in the reference (upfirdn2d.py:104,123-127).
//
// Three kernels, all HBM-bound (algorithmic bytes = 4*(in + out) per element):
//   * planes (minor == 1, up = down = 1): the NCHW hot case of the path (Blur of the
//     encoders / discriminator).  One 32x64 output tile per block, input tile (+kh-1,
//     +kw-1 halo) staged once in LDS, each lane owns one output column and slides down
//     8 rows so every LDS word is read kw times only; stores are 256-B wave rows.
//   * chan (minor % 4 == 0, up = down = 1): NHWC; lane = 4 channels (16-B loads/stores),
//     each thread slides over 4 output pixels along x re-using the kh x (4+kw-1) window.
//   * generic: any up/down/minor, one output per thread (modes 3-6 of the reference
//     dispatcher, never hit on the PPST path).
#include "common.h"
#include <stdlib.h>

#define UF_MAXK 8

struct UfParams {
  int major, in_h, in_w, minor, kh, kw;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int out_h, out_w;
  const float* k;  // device taps [kh][kw]; flipped on read (true convolution, .cu:81)
  int pad_mode;    // chan kernel only: PPST_PAD_ZERO or PPST_PAD_REFLECT (fused nn.ReflectionPad2d)
  const float* in_ss;  // chan kernel only: optional [major][minor][2] (a, s): input read as in_act(a*x + s)
  int in_act;          // PPST_ACT_NONE / PPST_ACT_LRELU
};

__device__ __forceinline__ int uf_reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// ---- planes: minor == 1, up == down == 1 ---------------------------------
template <int KH, int KW, int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void upfirdn2d_planes(const void* __restrict__ x, void* __restrict__ y, UfParams p) {
  constexpr int TH = 32, TW = 64, IH = TH + KH - 1, IW = TW + KW - 1;
  __shared__ float sx[IH][IW + 1];
  float kf[KH * KW];
#pragma unroll
  for (int i = 0; i < KH * KW; ++i) kf[i] = p.k[KH * KW - 1 - i];
  const int tiles_x = (p.out_w + TW - 1) / TW;
  const int tile_x = (blockIdx.x % tiles_x) * TW;
  const int tile_y = (blockIdx.x / tiles_x) * TH;
  const int64_t plane = blockIdx.y;
  const int64_t xp = plane * (int64_t)p.in_h * p.in_w, yp = plane * (int64_t)p.out_h * p.out_w;     // element offsets of the plane
  const int in_x0 = tile_x - p.pad_x0, in_y0 = tile_y - p.pad_y0;
  for (int i = threadIdx.x; i < IH * IW; i += 256) {
    int ry = i / IW, rx = i - ry * IW;
    int iy = in_y0 + ry, ix = in_x0 + rx;
    float v = 0.f;
    if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = st_ld1<ST>(x, xp + (int64_t)iy * p.in_w + ix);
    sx[ry][rx] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ox = tile_x + lane;
  float win[KH][KW];
#pragma unroll
  for (int r = 0; r < KH - 1; ++r)
#pragma unroll
    for (int c = 0; c < KW; ++c) win[r + 1][c] = sx[w * 8 + r][lane + c];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
#pragma unroll
    for (int rr = 0; rr < KH - 1; ++rr)
#pragma unroll
      for (int c = 0; c < KW; ++c) win[rr][c] = win[rr + 1][c];
#pragma unroll
    for (int c = 0; c < KW; ++c) win[KH - 1][c] = sx[w * 8 + r + KH - 1][lane + c];
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) acc += win[ky][kx] * kf[ky * KW + kx];
    int oy = tile_y + w * 8 + r;
    if (oy < p.out_h && ox < p.out_w) st_st1<ST>(y, yp + (int64_t)oy * p.out_w + ox, acc);
  }
}

#ifndef UF_PY
#define UF_PY 4                      // output rows per thread of the DOWN == 1 forms (2: -5..8 % at batch 24, equal at batch 8; 8: -15 % at batch 8)
#endif
// ---- chan: minor % 4 == 0, up == down == 1 --------------------------------
// DOWN = 2 keeps every second sample (the blur in front of a stride-2 1x1 conv only needs
// those).  S2D writes the output space-to-depth: out[m][oy>>1][ox>>1][((oy&1)*2+(ox&1))*C + c]
// with spatial extent ceil(out/2) -- the layout the fused conv consumes for stride-2 3x3 convs.
// ST: storage type of x and y (common.h): the taps are accumulated in fp32 either way, rounded once at the store.
// CV: channels per thread, 4 (16-byte items of an fp32 tensor) or 8 (16-byte items of a half tensor: with 4 a half launch is bound
// by instruction issue, not HBM -- 1024^2 x 32 ch fp16 s2d blur 0.209 ms against 0.229 in fp32).  CV = 8 halves the row group
// (PY = 2) to keep the accumulators in half the registers.
template <int KH, int KW, int DOWN, bool S2D, int ST = PPST_ST_F32, int CV = 4>
__global__ __launch_bounds__(256) void upfirdn2d_chan(const void* __restrict__ x, void* __restrict__ y, UfParams p, unsigned nwork,
                                                      FastDiv d_c, FastDiv d_xs, FastDiv d_oh) {
  // One thread = PX x PY output pixels x CV channels: the (PY-1+KH) x ((PX-1)*DOWN+KW) input patch is read once into
  // registers (DOWN = 1, PY = 4: 6 x 6 loads for 16 outputs; the rows shared with the thread above / below come through L2).
  constexpr int PX = 4, PY = (DOWN == 1) ? (CV == 8 ? UF_PY / 2 : UF_PY) : 1;
  constexpr int NQ = CV / 4;
  const int cvn = p.minor / CV;
  float kf[KH * KW];
#pragma unroll
  for (int i = 0; i < KH * KW; ++i) kf[i] = p.k[KH * KW - 1 - i];
  // S2D: the padded extent (2*ceil(oh/2) x 2*ceil(ow/2)) is written in full, zeros beyond (oh, ow)
  const int ew = S2D ? ((p.out_w + 1) & ~1) : p.out_w;
  // XCD-aware order (round 2): blocks are dealt round-robin over the 8 XCDs, so with a plain linear order the two
  // blocks that share halo rows (consecutive row pairs) sit on different XCDs and each XCD's L2 fetches those rows from
  // HBM itself -- the 1.4x over-fetch of round 1's PMC pass.  Virtual block v works on chunk
  // (v % 8) * ceil(T / 8) + v / 8: every XCD walks a contiguous band of rows (about one image at batch 8) in order.
  const unsigned nblk = (nwork + 255u) >> 8, per = (nblk + 7u) >> 3;
  auto ldv = [&](float4 (&v)[NQ], int64_t item) {        // item: index in units of CV channels
    if (CV == 8) st_ld8<ST>(x, item * 8, v[0], v[NQ - 1]);
    else v[0] = st_ld4<ST>(x, item * 4);
  };
  auto stv = [&](int64_t item, const float4 (&v)[NQ]) {
    if (CV == 8) st_st8<ST>(y, item * 8, v[0], v[NQ - 1]);
    else st_st4<ST>(y, item * 4, v[0]);
  };
  for (unsigned v = blockIdx.x; v < per * 8u; v += gridDim.x) {
    const unsigned wblk = (v & 7u) * per + (v >> 3);
    const uint64_t t64 = (uint64_t)wblk * 256 + threadIdx.x;
    if (wblk >= nblk || t64 >= nwork) continue;
    unsigned cqu, sxu, oyu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cqu);
    r = fd_divmod(r, d_xs, sxu);
    const int m = (int)fd_divmod(r, d_oh, oyu);   // d_oh divides by the number of PY-row groups
    const int cq = (int)cqu, oy0 = (int)oyu * PY;
    int ox0 = (int)sxu * PX;
    float4 acc[PY][PX][NQ];
#pragma unroll
    for (int j = 0; j < PY; ++j)
#pragma unroll
      for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[j][i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sa[NQ], sb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { sa[q] = make_float4(1.f, 1.f, 1.f, 1.f); sb[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (p.in_ss) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4* qp = (const float4*)(p.in_ss + ((int64_t)m * p.minor + cq * CV + q * 4) * 2);
        float4 q0 = qp[0], q1 = qp[1];
        sa[q] = make_float4(q0.x, q0.z, q1.x, q1.z); sb[q] = make_float4(q0.y, q0.w, q1.y, q1.w);
      }
    }
    const bool any_row = oy0 < p.out_h;
#pragma unroll
    for (int ry = 0; ry < (PY - 1) * DOWN + KH; ++ry) {
      int iy = oy0 * DOWN + ry - p.pad_y0;
      if (p.pad_mode == PPST_PAD_REFLECT) iy = uf_reflect(iy, p.in_h);
      bool yok = any_row && iy >= 0 && iy < p.in_h;
      const int64_t row = ((int64_t)m * p.in_h + (yok ? iy : 0)) * p.in_w * cvn + cq;      // (units of CV channels)
#pragma unroll
      for (int j = 0; j < (PX - 1) * DOWN + KW; ++j) {
        int ix = ox0 * DOWN + j - p.pad_x0;
        if (p.pad_mode == PPST_PAD_REFLECT) ix = uf_reflect(ix, p.in_w);
        float4 vv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) vv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yok && ix >= 0 && ix < p.in_w) {
          ldv(vv, row + (int64_t)ix * cvn);
          if (p.in_ss) {  // normalise on load; zero padding stays zero (it pads the normalised tensor)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              float4& v = vv[q];
              v.x = sa[q].x * v.x + sb[q].x; v.y = sa[q].y * v.y + sb[q].y; v.z = sa[q].z * v.z + sb[q].z; v.w = sa[q].w * v.w + sb[q].w;
              if (p.in_act == PPST_ACT_LRELU) {
                v.x = (v.x > 0.f ? v.x : v.x * 0.2f) * 1.41421356237309515f; v.y = (v.y > 0.f ? v.y : v.y * 0.2f) * 1.41421356237309515f;
                v.z = (v.z > 0.f ? v.z : v.z * 0.2f) * 1.41421356237309515f; v.w = (v.w > 0.f ? v.w : v.w * 0.2f) * 1.41421356237309515f;
              }
            }
          }
        }
#pragma unroll
        for (int py = 0; py < PY; ++py) {
          const int ky = ry - py * DOWN;
          if (ky < 0 || ky >= KH) continue;
#pragma unroll
          for (int i = 0; i < PX; ++i) {
            int kx = j - i * DOWN;
            if (kx >= 0 && kx < KW) {
              float f = kf[ky * KW + kx];
#pragma unroll
              for (int q = 0; q < NQ; ++q) {
                acc[py][i][q].x += vv[q].x * f; acc[py][i][q].y += vv[q].y * f; acc[py][i][q].z += vv[q].z * f; acc[py][i][q].w += vv[q].w * f;
              }
            }
          }
        }
      }
    }
    float4 zero[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) zero[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int py = 0; py < PY; ++py) {
      const int oy = oy0 + py;
      const bool row_ok = oy < p.out_h;
      if (S2D) {
        const int oh2 = (p.out_h + 1) >> 1, ow2 = (p.out_w + 1) >> 1;
        if (oy >= 2 * oh2) continue;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          int ox = ox0 + i;
          if (ox < ew)
            stv(((((int64_t)m * oh2 + (oy >> 1)) * ow2 + (ox >> 1)) * 4 + (oy & 1) * 2 + (ox & 1)) * cvn + cq,
                (row_ok && ox < p.out_w) ? acc[py][i] : zero);
        }
      } else {
        if (!row_ok) continue;
        const int64_t orow = ((int64_t)m * p.out_h + oy) * p.out_w * cvn + cq;
#pragma unroll
        for (int i = 0; i < PX; ++i)
          if (ox0 + i < p.out_w) stv(orow + (int64_t)(ox0 + i) * cvn, acc[py][i]);
      }
    }
  }
}

// ---- sliding form of upfirdn2d_chan for DOWN == 1 (round 5: the 4 x 4 blur of the train step's discriminator) ------------------
// upfirdn2d_chan re-requests a (PY - 1 + KH) x (PX - 1 + KW) patch per PX x PY outputs: 49 requests per 16 outputs with 4 x 4 taps --
// and it is bound by those requests, not by bytes (a bf16 tensor half the size ran SLOWER than its fp32 twin).  Here a thread owns a
// column strip of PX outputs and walks UF_SLIDE rows down it: every input row is requested ONCE per strip (PX - 1 + KW items) and
// feeds the KH output rows it belongs to, kept in a ring of KH accumulator rows (static slots: the row loop is unrolled by KH).  An
// output element receives its taps in the same order as in upfirdn2d_chan (patch rows ascending, columns ascending, one fma each):
// bit-identical results.  Same block order (XCD bands), padding modes, normalise-on-load and space-to-depth output.
#ifndef UF_SLIDE
#define UF_SLIDE 16
#endif
template <int KH, int KW, bool S2D, int ST = PPST_ST_F32, int CV = 4>
__global__ __launch_bounds__(256) void upfirdn2d_slide(const void* __restrict__ x, void* __restrict__ y, UfParams p, unsigned nwork,
                                                       FastDiv d_c, FastDiv d_xs, FastDiv d_oh) {
  constexpr int PX = 4, NQ = CV / 4, NJ = PX - 1 + KW, NR = UF_SLIDE + KH - 1;
  const int cvn = p.minor / CV;
  float kf[KH * KW];
#pragma unroll
  for (int i = 0; i < KH * KW; ++i) kf[i] = p.k[KH * KW - 1 - i];
  const int ew = S2D ? ((p.out_w + 1) & ~1) : p.out_w;
  const int oh2 = (p.out_h + 1) >> 1, ow2 = (p.out_w + 1) >> 1;
  const int eh = S2D ? 2 * oh2 : p.out_h;
  const unsigned nblk = (nwork + 255u) >> 8, per = (nblk + 7u) >> 3;
  auto ldv = [&](float4 (&v)[NQ], int64_t item) {
    if (CV == 8) st_ld8<ST>(x, item * 8, v[0], v[NQ - 1]);
    else v[0] = st_ld4<ST>(x, item * 4);
  };
  auto stv = [&](int64_t item, const float4 (&v)[NQ]) {
    if (CV == 8) st_st8<ST>(y, item * 8, v[0], v[NQ - 1]);
    else st_st4<ST>(y, item * 4, v[0]);
  };
  for (unsigned v = blockIdx.x; v < per * 8u; v += gridDim.x) {
    const unsigned wblk = (v & 7u) * per + (v >> 3);
    const uint64_t t64 = (uint64_t)wblk * 256 + threadIdx.x;
    if (wblk >= nblk || t64 >= nwork) continue;
    unsigned cqu, sxu, bandu;
    unsigned r = fd_divmod((unsigned)t64, d_c, cqu);
    r = fd_divmod(r, d_xs, sxu);
    const int m = (int)fd_divmod(r, d_oh, bandu);   // d_oh divides by the number of UF_SLIDE-row bands
    const int cq = (int)cqu, oy0 = (int)bandu * UF_SLIDE, ox0 = (int)sxu * PX;
    float4 acc[KH][PX][NQ];
#pragma unroll
    for (int j = 0; j < KH; ++j)
#pragma unroll
      for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[j][i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sa[NQ], sb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { sa[q] = make_float4(1.f, 1.f, 1.f, 1.f); sb[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (p.in_ss) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4* qp = (const float4*)(p.in_ss + ((int64_t)m * p.minor + cq * CV + q * 4) * 2);
        float4 q0 = qp[0], q1 = qp[1];
        sa[q] = make_float4(q0.x, q0.z, q1.x, q1.z); sb[q] = make_float4(q0.y, q0.w, q1.y, q1.w);
      }
    }
    // the column offsets of the strip's NJ items are the same in every row
    int colo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int ix = ox0 + j - p.pad_x0;
      if (p.pad_mode == PPST_PAD_REFLECT) ix = uf_reflect(ix, p.in_w);
      colo[j] = (ix >= 0 && ix < p.in_w) ? ix * cvn : -1;
    }
    float4 zero[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) zero[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < NR; r0 += KH) {
#pragma unroll
      for (int rr = 0; rr < KH; ++rr) {
        const int ry = r0 + rr;
        if (ry >= NR) continue;
        int iy = oy0 + ry - p.pad_y0;
        if (p.pad_mode == PPST_PAD_REFLECT) iy = uf_reflect(iy, p.in_h);
        const bool yok = iy >= 0 && iy < p.in_h;
        const int64_t row = ((int64_t)m * p.in_h + (yok ? iy : 0)) * p.in_w * cvn + cq;
        float4 vv[NJ][NQ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) vv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (yok && colo[j] >= 0) {
            ldv(vv[j], row + colo[j]);
            if (p.in_ss) {  // normalise on load; zero padding stays zero (it pads the normalised tensor)
#pragma unroll
              for (int q = 0; q < NQ; ++q) {
                float4& t = vv[j][q];
                t.x = sa[q].x * t.x + sb[q].x; t.y = sa[q].y * t.y + sb[q].y; t.z = sa[q].z * t.z + sb[q].z; t.w = sa[q].w * t.w + sb[q].w;
                if (p.in_act == PPST_ACT_LRELU) {
                  t.x = (t.x > 0.f ? t.x : t.x * 0.2f) * 1.41421356237309515f; t.y = (t.y > 0.f ? t.y : t.y * 0.2f) * 1.41421356237309515f;
                  t.z = (t.z > 0.f ? t.z : t.z * 0.2f) * 1.41421356237309515f; t.w = (t.w > 0.f ? t.w : t.w * 0.2f) * 1.41421356237309515f;
                }
              }
            }
          }
        }
        // this input row is tap row ky of output row py = ry - ky (ring slot (rr - ky) mod KH: r0 is a multiple of KH)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int ky = 0; ky < KH; ++ky) {
            const int py = ry - ky;
            if (py < 0 || py >= UF_SLIDE) continue;
            constexpr int KHc = KH;
            const int slot = (rr - ky + KHc) % KHc;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
              const int kx = j - i;
              if (kx >= 0 && kx < KW) {
                const float f = kf[ky * KW + kx];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                  acc[slot][i][q].x += vv[j][q].x * f; acc[slot][i][q].y += vv[j][q].y * f;
                  acc[slot][i][q].z += vv[j][q].z * f; acc[slot][i][q].w += vv[j][q].w * f;
                }
              }
            }
          }
        // output row ry - (KH - 1) has all its tap rows: store it, free its slot
        if (ry >= KH - 1) {
          const int slot = (rr + 1) % KH;
          const int oy = oy0 + ry - (KH - 1);
          const bool row_ok = oy < p.out_h;
          if (S2D) {
            if (oy < eh) {
#pragma unroll
              for (int i = 0; i < PX; ++i) {
                const int ox = ox0 + i;
                if (ox < ew)
                  stv(((((int64_t)m * oh2 + (oy >> 1)) * ow2 + (ox >> 1)) * 4 + (oy & 1) * 2 + (ox & 1)) * cvn + cq,
                      (row_ok && ox < p.out_w) ? acc[slot][i] : zero);
              }
            }
          } else if (row_ok) {
            const int64_t orow = ((int64_t)m * p.out_h + oy) * p.out_w * cvn + cq;
#pragma unroll
            for (int i = 0; i < PX; ++i)
              if (ox0 + i < p.out_w) stv(orow + (int64_t)(ox0 + i) * cvn, acc[slot][i]);
          }
#pragma unroll
          for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[slot][i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
}

// ---- generic ---------------------------------------------------------------
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void upfirdn2d_generic(const void* __restrict__ x, void* __restrict__ y, UfParams p, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    int mi = (int)(t % p.minor);
    int64_t r = t / p.minor;
    int ox = (int)(r % p.out_w); r /= p.out_w;
    int oy = (int)(r % p.out_h);
    int m = (int)(r / p.out_h);
    float acc = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
      int Y = oy * p.down_y + ky - p.pad_y0;
      if (Y < 0 || Y % p.up_y) continue;
      int iy = Y / p.up_y;
      if (iy >= p.in_h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        int X = ox * p.down_x + kx - p.pad_x0;
        if (X < 0 || X % p.up_x) continue;
        int ix = X / p.up_x;
        if (ix >= p.in_w) continue;
        acc += st_ld1<ST>(x, (((int64_t)m * p.in_h + iy) * p.in_w + ix) * p.minor + mi) * p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    st_st1<ST>(y, t, acc);
  }
}

// double (the reference: AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:225 -- scalar_t = double, taps double too,
// accumulation in scalar_t).  Not on the hot path: the generic form only.
__global__ __launch_bounds__(256) void upfirdn2d_generic_f64(const double* __restrict__ x, const double* __restrict__ k,
                                                             double* __restrict__ y, UfParams p, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    int mi = (int)(t % p.minor);
    int64_t r = t / p.minor;
    int ox = (int)(r % p.out_w); r /= p.out_w;
    int oy = (int)(r % p.out_h);
    int m = (int)(r / p.out_h);
    double acc = 0.0;
    for (int ky = 0; ky < p.kh; ++ky) {
      int Y = oy * p.down_y + ky - p.pad_y0;
      if (Y < 0 || Y % p.up_y) continue;
      int iy = Y / p.up_y;
      if (iy >= p.in_h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        int X = ox * p.down_x + kx - p.pad_x0;
        if (X < 0 || X % p.up_x) continue;
        int ix = X / p.up_x;
        if (ix >= p.in_w) continue;
        acc += x[(((int64_t)m * p.in_h + iy) * p.in_w + ix) * p.minor + mi] * k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    y[t] = acc;
  }
}

// ---- up2: minor % 4 == 0, up == 2, down == 1 (the adjoint of the decimating blur: zero-insert x2, FIR, crop) ------------
// One thread = 4 channels of one output pixel; only the taps whose upsampled coordinate is even touch an input sample
// (at most ceil(K/2)^2 of them).  32-bit index math with multiplier division (the generic kernel's int64 % and / cost
// ~100 instructions per element: 0.11 ms per launch in the train step).
template <int ST = PPST_ST_F32>
__global__ __launch_bounds__(256) void upfirdn2d_up2_chan(const void* __restrict__ x, void* __restrict__ y, UfParams p, unsigned total,
                                                          FastDiv d_c, FastDiv d_w, FastDiv d_h) {
  const int c4n = p.minor >> 2;
  for (uint64_t t64 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t64 < total; t64 += (uint64_t)gridDim.x * 256) {
    unsigned c4, oxu, oyu;
    unsigned r = fd_divmod((unsigned)t64, d_c, c4);
    r = fd_divmod(r, d_w, oxu);
    const int m = (int)fd_divmod(r, d_h, oyu);
    const int ox = (int)oxu, oy = (int)oyu;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < p.kh; ++ky) {
      const int Y = oy + ky - p.pad_y0;
      if (Y < 0 || (Y & 1) || (Y >> 1) >= p.in_h) continue;
      const int64_t row = ((int64_t)m * p.in_h + (Y >> 1)) * p.in_w * c4n + c4;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int X = ox + kx - p.pad_x0;
        if (X < 0 || (X & 1) || (X >> 1) >= p.in_w) continue;
        const float4 v = st_ld4<ST>(x, (row + (int64_t)(X >> 1) * c4n) * 4);
        const float f = p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
        acc.x += v.x * f; acc.y += v.y * f; acc.z += v.z * f; acc.w += v.w * f;
      }
    }
    st_st4<ST>(y, ((((int64_t)m * p.out_h + oy) * p.out_w + ox) * c4n + c4) * 4, acc);
  }
}

template <int KH, int KW, int ST = PPST_ST_F32, int CV = 4>
static int launch_chan(const void* x, void* y, const UfParams& p, int down, bool s2d, hipStream_t st) {
  const int eh = s2d ? ((p.out_h + 1) & ~1) : p.out_h, ew = s2d ? ((p.out_w + 1) & ~1) : p.out_w;
  static const bool no_slide = getenv("PPST_UF_NO_SLIDE") != nullptr;      // (A/B: the patch form for the 4 x 4 taps too)
  // 4 x 4 taps on half tensors of >= 192 rows: the sliding form (bit-identical, fewer requests per output).  Measured
  // (tests/blur_time.py, 4 images, us: patch form -> sliding form): bf16 512^2 x 64 158 -> 125, 256^2 x 128 88 -> 65, but 128^2 x 256
  // 52 -> 60 and 64^2 x 512 32 -> 47 (a band of 16 rows per thread leaves too few threads), and fp32 116 -> 128 / 52 -> 65: the fp32
  // patch form already moves its bytes at the HBM rate (4.6 TB/s) with all 49 requests of a thread in flight
  if (KH * KW >= 16 && down == 1 && !no_slide && ST != PPST_ST_F32 && eh >= 192) {
    const int bands = cdiv(eh, UF_SLIDE);
    const int64_t nwork = (int64_t)p.major * bands * cdiv(ew, 4) * (p.minor / CV);
    if (nwork > PPST_IDX32_MAX) return PPST_EINVAL;
    int64_t blocks = cdiv64(cdiv64(nwork, 256), 8) * 8;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const FastDiv d_c = make_fastdiv(p.minor / CV), d_xs = make_fastdiv(cdiv(ew, 4)), d_oh = make_fastdiv(bands);
    if (s2d) PPST_LAUNCH((upfirdn2d_slide<KH, KW, true, ST, CV>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, p, (unsigned)nwork, d_c, d_xs, d_oh);
    else PPST_LAUNCH((upfirdn2d_slide<KH, KW, false, ST, CV>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, p, (unsigned)nwork, d_c, d_xs, d_oh);
    return PPST_LAUNCH_CHECK();
  }
  const int upy = CV == 8 ? UF_PY / 2 : UF_PY;
  const int rows = down == 1 ? cdiv(eh, upy) : eh;   // row groups: PY output rows per thread when down == 1
  int64_t nwork = (int64_t)p.major * rows * cdiv(ew, 4) * (p.minor / CV);
  if (nwork > PPST_IDX32_MAX) return PPST_EINVAL;
  int64_t blocks = cdiv64(cdiv64(nwork, 256), 8) * 8;      // a multiple of 8: v % 8 is the XCD slot in every stride iteration
  if (blocks > 256 * 16) blocks = 256 * 16;
  dim3 g((unsigned)blocks), b(256);
  const FastDiv d_c = make_fastdiv(p.minor / CV), d_xs = make_fastdiv(cdiv(ew, 4)), d_oh = make_fastdiv(rows);
  const unsigned nw = (unsigned)nwork;
  if (s2d) PPST_LAUNCH((upfirdn2d_chan<KH, KW, 1, true, ST, CV>), g, b, 0, st, x, y, p, nw, d_c, d_xs, d_oh);
  else if (down == 2) PPST_LAUNCH((upfirdn2d_chan<KH, KW, 2, false, ST, CV>), g, b, 0, st, x, y, p, nw, d_c, d_xs, d_oh);
  else PPST_LAUNCH((upfirdn2d_chan<KH, KW, 1, false, ST, CV>), g, b, 0, st, x, y, p, nw, d_c, d_xs, d_oh);
  return PPST_LAUNCH_CHECK();
}

template <int KH, int KW, int ST = PPST_ST_F32>
static int launch_fast(const void* x, void* y, const UfParams& p, hipStream_t st) {
  if (p.minor == 1) {
    dim3 grid(cdiv(p.out_w, 64) * cdiv(p.out_h, 32), p.major);
    PPST_LAUNCH((upfirdn2d_planes<KH, KW, ST>), grid, dim3(256), 0, st, x, y, p);
  } else {
    return launch_chan<KH, KW, ST>(x, y, p, 1, false, st);
  }
  return PPST_LAUNCH_CHECK();
}

extern "C" int ppst_upfirdn2d(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor,
                              int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                              int pad_y0, int pad_y1, int dtype, void* stream) {
  // dtype: element type of x and y (the reference: AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:225); the taps k are fp32
  // at this boundary and the products are accumulated in fp32 (the reference accumulates in the tensor's type), rounded once.
  // PPST_F64: x, k AND y are double, accumulation in double (scalar_t = double in the reference's dispatch)
  if (dtype != PPST_F32 && dtype != PPST_F16 && dtype != PPST_BF16 && dtype != PPST_F64) return PPST_EUNSUPPORTED;
  if (dtype != PPST_F32 && (((uintptr_t)x | (uintptr_t)y) % 8)) return PPST_EINVAL;
  if (major != 0 && (!x || !k || !y)) return PPST_ENULL;  // an empty batch has no storage
  if (major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0 || kh > UF_MAXK || kw > UF_MAXK ||
      up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
    return PPST_EINVAL;
  UfParams p;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  if (p.out_h <= 0 || p.out_w <= 0) return PPST_EINVAL;
  if (major == 0) return PPST_OK;
  p.in_ss = nullptr; p.in_act = PPST_ACT_NONE;
  p.k = (const float*)k;
  p.pad_mode = PPST_PAD_ZERO;
  hipStream_t st = as_stream(stream);
  const float* xf = (const float*)x;
  float* yf = (float*)y;
  bool fast = up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && (minor == 1 || minor % 4 == 0) && kh == kw;
  bool down2 = up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2 && minor % 4 == 0 && kh == kw;
  int64_t n = (int64_t)major * p.out_h * p.out_w * minor;
  if (dtype == PPST_F64) {
    int64_t nb = cdiv64(n, 256);
    if (nb > 256 * 32) nb = 256 * 32;
    PPST_LAUNCH(upfirdn2d_generic_f64, dim3((unsigned)nb), dim3(256), 0, st, (const double*)x, (const double*)k, (double*)y, p, n);
    return PPST_LAUNCH_CHECK();
  }
  if (dtype != PPST_F32) {          // half / bfloat16: the same kernels on the tensor's storage type (the zero-insert x2 form: generic)
#define UF_HALF(ST_)                                                                            \
  do {                                                                                          \
    if (fast && kh == 3) return launch_fast<3, 3, ST_>(x, y, p, st);                            \
    if (fast && kh == 4) return launch_fast<4, 4, ST_>(x, y, p, st);                            \
    if (down2 && kh == 3) return launch_chan<3, 3, ST_>(x, y, p, 2, false, st);                 \
    if (down2 && kh == 4) return launch_chan<4, 4, ST_>(x, y, p, 2, false, st);                 \
    if (up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && minor % 4 == 0 && n / 4 <= PPST_IDX32_MAX) {   \
      int64_t b4_ = cdiv64(n / 4, 256);                                                         \
      if (b4_ > 256 * 32) b4_ = 256 * 32;                                                       \
      PPST_LAUNCH(upfirdn2d_up2_chan<ST_>, dim3((unsigned)b4_), dim3(256), 0, st, x, y, p, (unsigned)(n / 4),   \
                  make_fastdiv((unsigned)(minor / 4)), make_fastdiv((unsigned)p.out_w), make_fastdiv((unsigned)p.out_h)); \
      return PPST_LAUNCH_CHECK();                                                               \
    }                                                                                           \
    int64_t blocks_ = cdiv64(n, 256);                                                           \
    if (blocks_ > 256 * 32) blocks_ = 256 * 32;                                                 \
    PPST_LAUNCH(upfirdn2d_generic<ST_>, dim3((unsigned)blocks_), dim3(256), 0, st, x, y, p, n); \
    return PPST_LAUNCH_CHECK();                                                                 \
  } while (0)
    if (dtype == PPST_F16) UF_HALF(PPST_ST_F16);
    UF_HALF(PPST_ST_BF16);
#undef UF_HALF
  }
  if (fast && kh == 3) return launch_fast<3, 3>(xf, yf, p, st);
  if (fast && kh == 4) return launch_fast<4, 4>(xf, yf, p, st);
  if (down2 && kh == 3) return launch_chan<3, 3>(xf, yf, p, 2, false, st);
  if (down2 && kh == 4) return launch_chan<4, 4>(xf, yf, p, 2, false, st);
  if (up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && minor % 4 == 0 && n / 4 <= PPST_IDX32_MAX &&
      (((uintptr_t)x | (uintptr_t)y) % 16) == 0) {
    int64_t b4 = cdiv64(n / 4, 256);
    if (b4 > 256 * 32) b4 = 256 * 32;
    PPST_LAUNCH(upfirdn2d_up2_chan<PPST_ST_F32>, dim3((unsigned)b4), dim3(256), 0, st, (const void*)xf, (void*)yf, p, (unsigned)(n / 4),
                make_fastdiv((unsigned)(minor / 4)), make_fastdiv((unsigned)p.out_w), make_fastdiv((unsigned)p.out_h));
    return PPST_LAUNCH_CHECK();
  }
  int64_t blocks = cdiv64(n, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  PPST_LAUNCH(upfirdn2d_generic<PPST_ST_F32>, dim3((unsigned)blocks), dim3(256), 0, st, xf, yf, p, n);
  return PPST_LAUNCH_CHECK();
}

// Fused-path blur on NHWC activations (Blur of ConvLayer(downsample=True),
// stylegan2_layers.py:142-164,513-520): zero or reflection padding folded in (the reference
// runs nn.ReflectionPad2d first, :151-159), optional decimation by 2 (all a following
// stride-2 1x1 conv reads) or space-to-depth output for a following stride-2 3x3 conv:
//   s2d: y [B][ceil(oh/2)][ceil(ow/2)][4*C], phase (oy&1)*2+(ox&1) major over channels.
extern "C" int ppst_blur_nhwc_st(const void* x, const void* k, void* y, int B, int in_h, int in_w, int C, int ksize, int pad0,
                                 int pad1, int pad_mode, int down, int s2d, const void* in_scale_shift, int in_act, int st,
                                 void* stream) {
  if (!x || !k || !y) return PPST_ENULL;
  if ((unsigned)st > 2u) return PPST_EINVAL;
  if (B < 0 || in_h <= 0 || in_w <= 0 || C <= 0 || C % 4 || (ksize != 3 && ksize != 4) || (down != 1 && down != 2) ||
      (s2d && down != 1) || (pad_mode != PPST_PAD_ZERO && pad_mode != PPST_PAD_REFLECT) ||
      (in_act != PPST_ACT_NONE && in_act != PPST_ACT_LRELU) || (in_act != PPST_ACT_NONE && !in_scale_shift))
    return PPST_EINVAL;
  UfParams p;
  p.major = B; p.in_h = in_h; p.in_w = in_w; p.minor = C; p.kh = ksize; p.kw = ksize;
  p.up_x = p.up_y = 1; p.down_x = p.down_y = down; p.pad_x0 = pad0; p.pad_y0 = pad0;
  p.out_h = (in_h + pad0 + pad1 - ksize + down) / down;
  p.out_w = (in_w + pad0 + pad1 - ksize + down) / down;
  p.k = (const float*)k;
  p.pad_mode = pad_mode;
  p.in_ss = (const float*)in_scale_shift; p.in_act = in_act;
  if (p.out_h <= 0 || p.out_w <= 0) return PPST_EINVAL;
  if (B == 0) return PPST_OK;
  // half tensors: 8 channels per thread when every item is 16-byte addressable
  // (3 x 3 taps only: the 4 x 4 kernel with 8-channel items needs > 256 registers -- ONE wave per SIMD -- and the bf16 blurs of the
  //  train step's discriminator ran at half the speed of their fp32 twins on tensors half the size, 66 against 36 us per launch;
  //  PPST_UF_C8_4X4=1 restores that form for A/B)
  static const bool c8_4x4 = getenv("PPST_UF_C8_4X4") != nullptr;
  const bool c8 = st != PPST_ST_F32 && C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0 && (ksize == 3 || c8_4x4);
  if (st != PPST_ST_F32 && (((uintptr_t)x | (uintptr_t)y) % 8)) return PPST_EINVAL;
#define UF_GO(K_)                                                                                             \
  do {                                                                                                        \
    if (st == PPST_ST_F16) return c8 ? launch_chan<K_, K_, PPST_ST_F16, 8>(x, y, p, down, s2d != 0, as_stream(stream))   \
                                     : launch_chan<K_, K_, PPST_ST_F16, 4>(x, y, p, down, s2d != 0, as_stream(stream));  \
    if (st == PPST_ST_BF16) return c8 ? launch_chan<K_, K_, PPST_ST_BF16, 8>(x, y, p, down, s2d != 0, as_stream(stream)) \
                                      : launch_chan<K_, K_, PPST_ST_BF16, 4>(x, y, p, down, s2d != 0, as_stream(stream)); \
    return launch_chan<K_, K_, PPST_ST_F32, 4>(x, y, p, down, s2d != 0, as_stream(stream));                   \
  } while (0)
  if (ksize == 3) UF_GO(3);
  UF_GO(4);
#undef UF_GO
}
extern "C" int ppst_blur_nhwc(const void* x, const void* k, void* y, int B, int in_h, int in_w, int C, int ksize, int pad0,
                              int pad1, int pad_mode, int down, int s2d, const void* in_scale_shift, int in_act, void* stream) {
  return ppst_blur_nhwc_st(x, k, y, B, in_h, in_w, C, ksize, pad0, pad1, pad_mode, down, s2d, in_scale_shift, in_act, PPST_ST_F32, stream);
}
