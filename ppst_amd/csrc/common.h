// Shared device/host helpers for libppst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ppst_hip.h"

// hipGetLastError() is sticky across libraries: clear stale errors (e.g. from the caller's own
// device probing) before a launch so the check after it reports this launch only.
#define PPST_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define PPST_LAUNCH_CHECK() ((int)hipGetLastError())

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Output store of the conv epilogues.  -DPPST_EPI_NT (experiment, measured: no effect): non-temporal hint.
#if defined(PPST_EPI_NOSTORE)   // timing ablation (results WRONG): what the output stores and their acknowledgements cost
#define PPST_EPI_STORE(ptr, o) ((void)(ptr))
#elif defined(PPST_EPI_NT)
#define PPST_EPI_STORE(ptr, o) __builtin_nontemporal_store((f32x4){(o)[0], (o)[1], (o)[2], (o)[3]}, (f32x4*)(ptr))
#else
#define PPST_EPI_STORE(ptr, o) (*(float4*)(ptr) = make_float4((o)[0], (o)[1], (o)[2], (o)[3]))
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Tags for the conv epilogues' (activation, residual mode): EpiC = compile-time (the production kernels: the pass loops are
// instantiated per combination), EpiR = run-time (experiment / reduced-precision kernels: one generic instance, shorter build).
template <int V> struct EpiC { static constexpr int value = V; };
struct EpiR { int value; };

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// ---- across-block K split of a conv launch (ppst_conv_args.ksplit; conv_mfma.hip holds the state).  The S blocks of an output
// tile are grid rows y = 0 .. S-1 of the same x: rows 0 .. S-2 (dispatched first: x runs fastest) store their accumulators --
// component e of register r of thread t at float ((r * 4 + e) * NT + t), coalesced -- and raise flag[tile][y] to the launch's epoch; row S-1 waits for the
// S-1 flags (they belong to blocks dispatched before it, which wait for nothing: no deadlock whatever the residency), adds the
// partial sums in row order and goes on to its unchanged epilogue.  The wait is bounded (about a second): a block that gives up
// sets the error word instead of hanging the device.
// Coherence: the XCDs' L2s are not coherent with each other.  The hand-over goes through coherent accesses (sc bits: stores write
// through to memory, loads do not hit a stale line), ordered by counters -- all of a wave's stores acknowledged (vmcnt 0), block
// barrier, flag -- and NOT through release / acquire fences: a fence at agent scope writes back / invalidates the WHOLE L2 of the
// XCD (buffer_wbl2 / buffer_inv sc1), under blocks that are streaming their weights from it (first form of this code: the split
// launches were SLOWER than the unsplit ones, 67 -> 85 us on 256 -> 256 @64^2 x 2).
struct KSplitDev {
  float* scratch;        // [tile][S-1][acc regs][NT] floats
  unsigned* flags;       // [tile][S-1]; word KS_FLAG_WORDS - 1: error marker
  unsigned epoch;
  int S;
  int start[9];          // block row y runs steps [start[y], start[y + 1]) of every group; start[S] = nsteps
};
#define KS_FLAG_WORDS 4096
#define KS_MAX_SLOTS 512
#define KS_SCRATCH_BYTES ((size_t)256 * 512 * 128 * 4)      /* 256 producer blocks x 512 threads x 128 accumulator registers (512 of the tile kernel's) */
int ppst_ksplit_prepare_(int S, const int32_t* starts, int64_t tiles, int nsteps, int acc_regs, int threads, hipStream_t st, KSplitDev* out);   // conv_mfma.hip

// Agent-scope relaxed atomic accesses, one dword each (sc1; a 16-byte volatile access gets sc0 sc1 -- system scope -- and measured
// ~3 us slower per launch).  hipcc tracks them with counted vmcnt like plain loads but is free to hoist them: left alone it moved the
// (S - 1) x 64 .. 128 requests of a thread to the front and spilled 200-700 registers to scratch in EVERY instance of the kernels.
// The scheduling fences below pin the order instead: the requests of register r + 1, then the adds of register r.
template <int NTHR>
__device__ __forceinline__ void ks_vst(float* p, f32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) __hip_atomic_store(p + e * NTHR, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NTHR>
__device__ __forceinline__ f32x4 ks_vld(const float* p) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __hip_atomic_load(p + e * NTHR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
// acc[I][J] (f32x4 each) += the S-1 partner slots; component e of register r of thread t lives at float ((r * 4 + e) * NTHR + t) of
// a slot: a wave's request is 256 contiguous bytes (`slot0` already points at this thread's float of slot 0).  Double-buffered by hand: the S-1 requests of register r+1 go out before the adds of
// register r, so 2 (S - 1) float4 are live and one memory round trip is exposed, not one per register.
#define KS_GATHER_BODY(ACC, I_, J_, NTHR_, P_, SLOT0)                                                  \
  {                                                                                                   \
    constexpr int R_ = (I_) * (J_);                                                                   \
    f32x4 t_[2][P_];                                                                                  \
    _Pragma("unroll") for (int z = 0; z < (P_); ++z) t_[0][z] = ks_vld<NTHR_>((SLOT0) + (int64_t)z * (R_ * 4 * (NTHR_)));        \
    _Pragma("unroll") for (int r = 0; r < R_; ++r) {                                                  \
      if (r + 1 < R_) {                                                                               \
        _Pragma("unroll") for (int z = 0; z < (P_); ++z)                                              \
          t_[(r + 1) & 1][z] = ks_vld<NTHR_>((SLOT0) + (int64_t)z * (R_ * 4 * (NTHR_)) + (r + 1) * 4 * (NTHR_));                 \
      }                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                              \
      _Pragma("unroll") for (int z = 0; z < (P_); ++z) ACC[r / (J_)][r % (J_)] += t_[r & 1][z];       \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  }
// (S = 8 only where a thread holds 64 accumulator registers: with 128 the 56 registers of its double buffer spill)
#define KS_GATHER(ACC, I_, J_, NTHR_, S_, SLOT0)                                                       \
  do {                                                                                                \
    if ((S_) == 2) KS_GATHER_BODY(ACC, I_, J_, NTHR_, 1, SLOT0)                                       \
    else if ((S_) == 4 || (I_) * (J_) > 16) KS_GATHER_BODY(ACC, I_, J_, NTHR_, 3, SLOT0)              \
    else KS_GATHER_BODY(ACC, I_, J_, NTHR_, (((I_) * (J_) > 16) ? 3 : 7), SLOT0)                      \
  } while (0)
#define KS_SCATTER(ACC, I_, J_, NTHR_, DST)                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < (I_); ++i_)                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < (J_); ++j_) ks_vst<NTHR_>((DST) + (i_ * (J_) + j_) * 4 * (NTHR_), ACC[i_][j_]);
// steps [s0, s1) of block row y (a compare chain over constant indices: indexing the kernel-argument array with y made hipcc copy the
// whole argument structure to scratch memory)
__device__ __forceinline__ void ks_range(const KSplitDev& k, int y, int& s0, int& s1) {
  s0 = 0; s1 = k.start[1];
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (y == i) { s0 = k.start[i]; s1 = k.start[i + 1]; }
}
__device__ __forceinline__ void ks_publish(const KSplitDev& k, int tile, int y, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores are acknowledged
  __syncthreads();
  if (tid == 0) __hip_atomic_store(k.flags + tile * (k.S - 1) + y, k.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ks_wait(const KSplitDev& k, int tile, int tid) {
  if (tid < k.S - 1) {
    const unsigned* f = k.flags + tile * (k.S - 1) + tid;
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k.epoch) {
      if (++spins > (1 << 21)) { __hip_atomic_store(k.flags + KS_FLAG_WORDS - 1, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Division of a 32-bit index by a launch constant without the 64-bit divide sequence hipcc emits
// for `int64 / int` (about 100 instructions per quotient: it made the elementwise kernels
// ALU-bound): Granlund-Montgomery round-up multiplier, exact for every n < 2^32 and d >= 1.
struct FastDiv { unsigned d, m, s1, s2; };
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  if (d == 0) d = 1;  // never dereferenced for empty shapes; keeps the host math defined
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l == 0 ? 0 : l - 1;
  return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned n, const FastDiv& f) {
  unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
// n -> (n / d, n % d)
__device__ __forceinline__ unsigned fd_divmod(unsigned n, const FastDiv& f, unsigned& rem) {
  unsigned q = fd_div(n, f);
  rem = n - q * f.d;
  return q;
}
#define PPST_IDX32_MAX 0xFFFFFFFFll  // element-index limit of the kernels that use FastDiv

// fp32 -> bf16 round-to-nearest-even as raw 16 bits (plain cast: hipcc emits
// v_cvt_pk_bf16_f32 on gfx950, which keeps NaN a NaN)
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// split x = hi + lo (+O(2^-17 |x|)) with hi, lo bf16
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// the same split for four values at once, two elements per instruction: v_cvt_pk_bf16_f32 converts a PAIR, v_pk_add_f32
// subtracts a pair -- 10 vector instructions per float4 (2 cvt + 4 shift / mask + 2 sub + 2 cvt).  The element-wise form above
// compiled to 12 cvt_pk + 12 SDWA merges + 8 more per float4 in the conv kernels' staging (hipcc does not pair scalar casts).
// Bit-identical: the same round-to-nearest-even conversions and the same fp32 subtraction.
typedef __bf16 __attribute__((ext_vector_type(2))) ppst_bf2;
typedef float __attribute__((ext_vector_type(2))) ppst_f2;
__device__ __forceinline__ unsigned f2bf_pk(ppst_f2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ppst_bf2)); }
__device__ __forceinline__ void split_bf16x4(float4 v, uint2& hi, uint2& lo) {
  const ppst_f2 a = {v.x, v.y}, b = {v.z, v.w};
  const unsigned h0 = f2bf_pk(a), h1 = f2bf_pk(b);
  const ppst_f2 ra = a - (ppst_f2){__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
  const ppst_f2 rb = b - (ppst_f2){__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
  hi = make_uint2(h0, h1);
  lo = make_uint2(f2bf_pk(ra), f2bf_pk(rb));
}
__device__ __forceinline__ uint2 f2bf_x4(float4 v) {        // single-pass bf16: the hi halves only
  return make_uint2(f2bf_pk((ppst_f2){v.x, v.y}), f2bf_pk((ppst_f2){v.z, v.w}));
}

// ---- activation storage type (round 4: half-precision activation storage of precision modes 1 / 3) ------------------------
// An NHWC activation tensor at the ABI is fp32 (PPST_ST_F32), IEEE half (PPST_ST_F16, mode 3) or bfloat16 (PPST_ST_BF16,
// mode 1).  Kernels that take a storage type compute in fp32 exactly as their fp32 form and round ONCE, to nearest even, when
// they store: a half-storage launch == the fp32 launch on the widened inputs, rounded (tests/gpu_diag.py t_half_storage).
// Indices of the helpers are ELEMENT offsets (multiples of 4 for the vector forms).
// (PPST_ST_F32 / PPST_ST_F16 / PPST_ST_BF16 = 0 / 1 / 2: include/ppst_hip.h)
template <int ST> __device__ __forceinline__ float4 st_unpack4(uint2 u) {
  if (ST == PPST_ST_F16) {
    typedef _Float16 __attribute__((ext_vector_type(2))) h2;
    const h2 a = __builtin_bit_cast(h2, u.x), b = __builtin_bit_cast(h2, u.y);
    return make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
  }
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
template <int ST> __device__ __forceinline__ uint2 st_pack4(float4 v) {
  if (ST == PPST_ST_F16) {
    typedef _Float16 __attribute__((ext_vector_type(2))) h2;
    typedef float __attribute__((ext_vector_type(2))) f2;
    const h2 a = __builtin_convertvector((f2){v.x, v.y}, h2), b = __builtin_convertvector((f2){v.z, v.w}, h2);
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
  }
  return f2bf_x4(v);
}
template <int ST> __device__ __forceinline__ float4 st_ld4(const void* base, int64_t i) {
  if (ST == PPST_ST_F32) return *(const float4*)((const float*)base + i);
  return st_unpack4<ST>(*(const uint2*)((const unsigned short*)base + i));
}
template <int ST> __device__ __forceinline__ void st_st4(void* base, int64_t i, float4 v) {
  if (ST == PPST_ST_F32) *(float4*)((float*)base + i) = v;
  else *(uint2*)((unsigned short*)base + i) = st_pack4<ST>(v);
}
// eight consecutive elements: one 16-byte access of a half tensor (two of an fp32 one); i a multiple of 8, base 16-byte aligned
template <int ST> __device__ __forceinline__ void st_ld8(const void* base, int64_t i, float4& a, float4& b) {
  if (ST == PPST_ST_F32) {
    a = *(const float4*)((const float*)base + i);
    b = *(const float4*)((const float*)base + i + 4);
  } else {
    const uint4 u = *(const uint4*)((const unsigned short*)base + i);
    a = st_unpack4<ST>(make_uint2(u.x, u.y));
    b = st_unpack4<ST>(make_uint2(u.z, u.w));
  }
}
template <int ST> __device__ __forceinline__ void st_st8(void* base, int64_t i, float4 a, float4 b) {
  if (ST == PPST_ST_F32) {
    *(float4*)((float*)base + i) = a;
    *(float4*)((float*)base + i + 4) = b;
  } else {
    const uint2 p = st_pack4<ST>(a), q = st_pack4<ST>(b);
    *(uint4*)((unsigned short*)base + i) = make_uint4(p.x, p.y, q.x, q.y);
  }
}
template <int ST> __device__ __forceinline__ float st_ld1(const void* base, int64_t i) {
  if (ST == PPST_ST_F32) return ((const float*)base)[i];
  const unsigned short h = ((const unsigned short*)base)[i];
  if (ST == PPST_ST_F16) return (float)__builtin_bit_cast(_Float16, h);
  return bf2f(h);
}
template <int ST> __device__ __forceinline__ void st_st1(void* base, int64_t i, float v) {
  if (ST == PPST_ST_F32) ((float*)base)[i] = v;
  else if (ST == PPST_ST_F16) ((unsigned short*)base)[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
  else ((unsigned short*)base)[i] = f2bf(v);
}
static inline int st_bytes(int st) { return st == PPST_ST_F32 ? 4 : 2; }
// run `BODY` with the compile-time constant ST_ = st (host-side dispatch of the templated launches)
#define PPST_ST_SWITCH(st, BODY)                                              \
  do {                                                                        \
    if ((st) == PPST_ST_F16) { constexpr int ST_ = PPST_ST_F16; BODY; }       \
    else if ((st) == PPST_ST_BF16) { constexpr int ST_ = PPST_ST_BF16; BODY; } \
    else { constexpr int ST_ = PPST_ST_F32; BODY; }                           \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
