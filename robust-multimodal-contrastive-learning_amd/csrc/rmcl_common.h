// Common device/host helpers for the RMCL gfx950 kernels.  gfx950 (CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <atomic>

#define RMCL_F32 0
#define RMCL_BF16 1

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);  // RNE, NaN-preserving (v_cvt_pk_bf16_f32 at -O3)
  return *reinterpret_cast<bf16_t*>(&h);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// Fast GELU / GELU' for the bf16 path's GEMM epilogues: erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, below bf16 and fp32-accumulation noise) - one v_exp + one v_rcp instead of
// the ~40-instruction erff.  exp(-z^2) with z = x/sqrt(2) is exp(-x^2/2): shared by GELU and GELU'.
// The exact-f32 parity path keeps erff.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& pdf_e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;                    // erf(|z|)
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));              // Phi(x)
  pdf_e = e;                                                // exp(-x^2/2)
}
// Cheaper still, for the 192x192 GEMM's epilogue (72 values per lane, where the A-S form costs ~6 us per output tile of
// VALU time with the MFMAs idle): erf(y) ~ y * P(y^2) on the clamped range |y| <= 3.2 (least-squares Chebyshev fit, degree 8;
// max |erf error| 4.4e-5 -> |GELU error| <= 1.3e-4 absolute, 3e-5 relative to |x| - two orders below the bf16 rounding of
// the stored activation).  No transcendental for GELU, one v_exp for GELU'.
// Round 4: y = x / sqrt(2) and the 1/2 of Phi = (1 + erf) / 2 are folded into the coefficients - Phi(x) - 1/2 = xc * Q(xc^2) with
// xc = x clamped to +-3.2 sqrt(2), Q_k = P_k / (2 sqrt(2) 2^k) - so GELU = x * Phi costs 13 instructions per value pair instead of 15
// and GELU' = Phi + x phi(x), with exp(-x^2/2) taken as exp2((x * -log2(e)/2) * x), 20 instead of 24 (max |GELU error| 8.3e-5).
// The two-wide forms (f32x2) are the same operations per component: the epilogues hold four consecutive columns per accumulator and name
// the pairs (0, 1) / (2, 3) themselves - left to the SLP vectoriser the pairs came out as (0, 2) / (1, 3), which cost six register moves
// in and four fix-up instructions out per four values around v_pk_fma_f32 / v_cvt_pk_bf16_f32 (fc1's epilogue: 2 400 VALU instructions
// per lane and tile with the MFMAs idle, 1 640 with named pairs and once-per-tile addresses).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RMCL_PHI_CLAMP 4.5254833996f
#define RMCL_PHI_Q0 3.988472601e-01f
#define RMCL_PHI_Q1 -6.619333253e-02f
#define RMCL_PHI_Q2 9.695875257e-03f
#define RMCL_PHI_Q3 -1.065986489e-03f
#define RMCL_PHI_Q4 8.545539012e-05f
#define RMCL_PHI_Q5 -4.784974427e-06f
#define RMCL_PHI_Q6 1.750769354e-07f
#define RMCL_PHI_Q7 -3.726298549e-09f
#define RMCL_PHI_Q8 3.477167974e-11f
__device__ __forceinline__ float phi_poly(float x) {         // Phi(x), the standard normal CDF
  const float xc = __builtin_amdgcn_fmed3f(x, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  const float s = xc * xc;
  float q = RMCL_PHI_Q8;
  q = fmaf(q, s, RMCL_PHI_Q7);
  q = fmaf(q, s, RMCL_PHI_Q6);
  q = fmaf(q, s, RMCL_PHI_Q5);
  q = fmaf(q, s, RMCL_PHI_Q4);
  q = fmaf(q, s, RMCL_PHI_Q3);
  q = fmaf(q, s, RMCL_PHI_Q2);
  q = fmaf(q, s, RMCL_PHI_Q1);
  q = fmaf(q, s, RMCL_PHI_Q0);
  return fmaf(xc, q, 0.5f);
}
__device__ __forceinline__ float gelu_poly(float x) { return x * phi_poly(x); }
__device__ __forceinline__ float gelu_poly_grad(float x) {
  return fmaf(x * 0.39894228040143268f, __builtin_amdgcn_exp2f((x * -0.72134752044448170f) * x), phi_poly(x));
}
__device__ __forceinline__ f32x2 phi_poly2(f32x2 x) {
  f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  const f32x2 s = xc * xc;
  f32x2 q = RMCL_PHI_Q8;
  q = q * s + RMCL_PHI_Q7;
  q = q * s + RMCL_PHI_Q6;
  q = q * s + RMCL_PHI_Q5;
  q = q * s + RMCL_PHI_Q4;
  q = q * s + RMCL_PHI_Q3;
  q = q * s + RMCL_PHI_Q2;
  q = q * s + RMCL_PHI_Q1;
  q = q * s + RMCL_PHI_Q0;
  return xc * q + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) { return x * phi_poly2(x); }
__device__ __forceinline__ f32x2 gelu_poly_grad2(f32x2 x) {
  const f32x2 t = (x * -0.72134752044448170f) * x;
  const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  return (x * 0.39894228040143268f) * e + phi_poly2(x);
}
// Four-wide forms: the same operations again, written on all four values of an accumulator at once so that every polynomial step is
// issued for the pair (0, 1) and then for the pair (2, 3) - two INDEPENDENT v_pk_fma_f32 chains interleaved.  A dependent v_pk_fma_f32
// cannot issue back to back (the compiler pads one chain with s_nop), and in the epilogue a wave has its SIMD to itself (the other
// wave group is one barrier away, storing): one chain after the other ran at ~8 cycles per instruction (tools/st_trace.py fc1: 2.7 us
// for the 820 VALU instructions of a chunk).
__device__ __forceinline__ f32x4 phi_poly4(f32x4 x) {
  f32x4 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  xc.z = __builtin_amdgcn_fmed3f(x.z, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  xc.w = __builtin_amdgcn_fmed3f(x.w, -RMCL_PHI_CLAMP, RMCL_PHI_CLAMP);
  const f32x4 s = xc * xc;
  f32x4 q = RMCL_PHI_Q8;
  q = q * s + RMCL_PHI_Q7;
  q = q * s + RMCL_PHI_Q6;
  q = q * s + RMCL_PHI_Q5;
  q = q * s + RMCL_PHI_Q4;
  q = q * s + RMCL_PHI_Q3;
  q = q * s + RMCL_PHI_Q2;
  q = q * s + RMCL_PHI_Q1;
  q = q * s + RMCL_PHI_Q0;
  return xc * q + 0.5f;
}
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 x) { return x * phi_poly4(x); }
__device__ __forceinline__ f32x4 gelu_poly_grad4(f32x4 x) {
  const f32x4 t = (x * -0.72134752044448170f) * x;
  const f32x4 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
  return (x * 0.39894228040143268f) * e + phi_poly4(x);
}
// two fp32 -> one dword of two bf16 (RNE; v_cvt_pk_bf16_f32)
typedef __bf16 rmcl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rmcl_bf16x2)); }
__device__ __forceinline__ uint2 f2bf4(f32x4 v) { return make_uint2(f2bf2(f32x2{v.x, v.y}), f2bf2(f32x2{v.z, v.w})); }
__device__ __forceinline__ f32x4 bf2f4(uint2 u) {
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 f4v(float4 a) { return f32x4{a.x, a.y, a.z, a.w}; }
__device__ __forceinline__ f32x2 bf2f2(uint32_t u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ float gelu_fast(float x) {
  float c, e;
  gelu_fast_parts(x, c, e);
  return x * c;
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
  float c, e;
  gelu_fast_parts(x, c, e);
  return c + x * 0.39894228040143268f * e;
}

// Counter-based dropout RNG: keep/scale factor of element `idx` of a dropout site is a pure function of
// (site seed, idx), so the backward regenerates the forward's mask instead of storing it.
__host__ __device__ __forceinline__ uint32_t rmcl_rng_hash(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rmcl_site_seed(uint32_t seed, int layer, int site) {
  return rmcl_rng_hash(seed ^ 0xA511E9B3u, (uint32_t)(layer * 8 + site + 1));
}
// Round 4: the mask of FOUR consecutive elements (idx >> 2) comes from two 32-bit words - a two-multiply mix of (seed, idx >> 2) and one more
// multiply + shift of that word - and element idx is kept iff its 16-bit quarter is >= thresh >> 16 (p resolved to 2^-16: 0.1 -> 0.09999).
// v_mul_lo_u32 runs at a quarter of the VALU rate: with one three-multiply hash per element the dropout epilogues cost fc1 +10 us per
// launch (144 elements per lane), with one per pair +14 us on a 76 us launch after the epilogue rework; three multiplies per four elements
// now.  Keep rates 0.8998-0.9002 and |pairwise correlation| <= 1.2e-3 over 4 M groups (inside a group, between neighbouring groups, one
// row apart) for the seeds tried.  The mask is a pure function of (site seed, element index) wherever it is evaluated.
__host__ __device__ __forceinline__ void rmcl_drop_words(uint32_t seed, uint32_t group, uint32_t& h0, uint32_t& h1) {
  uint32_t x = group * 0x9E3779B1u ^ seed;
  x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
  uint32_t y = x * 0xC2B2AE35u;
  y ^= y >> 16;
  h0 = x;
  h1 = y;
}
__host__ __device__ __forceinline__ bool rmcl_drop_keep(uint32_t seed, uint32_t idx, uint32_t thresh) {
  uint32_t h0, h1;
  rmcl_drop_words(seed, idx >> 2, h0, h1);
  const uint32_t w = (idx & 2u) ? h1 : h0;
  return ((idx & 1u) ? (w >> 16) : (w & 0xffffu)) >= (thresh >> 16);
}
__device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
  return rmcl_drop_keep(seed, idx, thresh) ? inv_keep : 0.f;
}
// v[0..3] *= mask of elements idx .. idx + 3, idx a multiple of 4 (every caller: column offsets and leading dimensions are multiples of 4)
__device__ __forceinline__ void drop_scale4(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep, float& v0, float& v1, float& v2, float& v3) {
  uint32_t h0, h1;
  rmcl_drop_words(seed, idx >> 2, h0, h1);
  const uint32_t t = thresh >> 16;
  v0 *= (h0 & 0xffffu) >= t ? inv_keep : 0.f;
  v1 *= (h0 >> 16) >= t ? inv_keep : 0.f;
  v2 *= (h1 & 0xffffu) >= t ? inv_keep : 0.f;
  v3 *= (h1 >> 16) >= t ? inv_keep : 0.f;
}
__device__ __forceinline__ void drop_scale4(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep, f32x2& v01, f32x2& v23) {
  float t0 = v01.x, t1 = v01.y, t2 = v23.x, t3 = v23.y;
  drop_scale4(seed, idx, thresh, inv_keep, t0, t1, t2, t3);
  v01 = f32x2{t0, t1};
  v23 = f32x2{t2, t3};
}
__device__ __forceinline__ void drop_scale4(uint32_t seed, uint32_t idx, uint32_t thresh, float inv_keep, f32x4& v) {
  float t0 = v.x, t1 = v.y, t2 = v.z, t3 = v.w;
  drop_scale4(seed, idx, thresh, inv_keep, t0, t1, t2, t3);
  v = f32x4{t0, t1, t2, t3};
}
enum { DROP_SITE_PROJ = 0, DROP_SITE_HIDDEN = 1, DROP_SITE_FC2 = 2, DROP_SITE_TEXT = 3, DROP_SITE_IMAGE = 4 };

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

// ds_read_b64_tr_b16 through inline asm.  The builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16) carries no alias information, so
// hipcc (ROCm 7.2) orders it behind EVERY LDS-DMA still in flight with an s_waitcnt vmcnt(0) - in a k-loop that keeps the next
// tiles' global_load_lds in flight across barriers that wait drains the whole prefetch at the head of every k-tile (found in the
// weight-gradient launch's .s: one vmcnt(0) per k-tile in front of its 36 transposed reads; row reads through a bf16x8-typed
// pointer are spared by type-based alias analysis).  The asm form is invisible to that pass; the CALLER owns both waits: the counted
// vmcnt + barrier that makes the DMA'd bytes visible, and an explicit s_waitcnt lgkmcnt(0) followed by sched_barrier(0) before the
// first use of the result (cdna_hip_programming.md 5.7 form (iii)).  `a` = LDS byte address (lds_addr), OFF = immediate offset.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16_asm(uint32_t a) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}

// ---- host side -------------------------------------------------------------------------------
#ifdef __cplusplus
extern "C" void rmcl_set_error(const char* msg);
#endif

// Dynamic-LDS limit of a kernel, set once per (launcher, device) - hipFuncSetAttribute costs the host ~170 us, so not per launch - and
// again when a larger size is asked for.  Keyed by the current device (a second device in the process would otherwise launch with the
// default limit), safe to race (the call is idempotent; ctypes releases the GIL around these entry points), and the HIP status is
// surfaced through rmcl_last_error instead of being dropped.   Use:  static RmclLdsOnce once; RMCL_TRY(rmcl_set_max_lds(once, fn, bytes));
struct RmclLdsOnce { std::atomic<int> bytes[16]; };
static inline int rmcl_set_max_lds(RmclLdsOnce& o, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  dev &= 15;
  if (o.bytes[dev].load(std::memory_order_acquire) >= bytes) return 0;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    rmcl_set_error(hipGetErrorString(e));
    return (int)e;
  }
  o.bytes[dev].store(bytes, std::memory_order_release);
  return 0;
}

// Launch with a clean error slate: hipGetLastError() is per-thread sticky state that other HIP users in
// the process (torch) may have left set; clear it first so RMCL_CHECK_LAUNCH reports OUR launch only.
#define RMCL_LAUNCH(...)                 \
  do {                                   \
    (void)hipGetLastError();             \
    hipLaunchKernelGGL(__VA_ARGS__);     \
  } while (0)

#define RMCL_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t _e = hipGetLastError();                        \
    if (_e != hipSuccess) {                                   \
      rmcl_set_error(hipGetErrorString(_e));                  \
      return (int)_e;                                         \
    }                                                         \
  } while (0)

#define RMCL_REQUIRE(cond, msg)                               \
  do {                                                        \
    if (!(cond)) {                                            \
      rmcl_set_error("rmcl: requirement failed: " msg);       \
      return -1;                                              \
    }                                                         \
  } while (0)

#define RMCL_TRY(expr)                                        \
  do {                                                        \
    int _r = (expr);                                          \
    if (_r != 0) return _r;                                   \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
