// Fused masked multi-head self-attention for gfx950 (bf16 MFMA, fp32 softmax), forward and backward.
// Replaces Attention.forward (vilt/modules/vision_transformer.py:309-332): QK^T * 0.125, key-padding
// mask as -inf, softmax, @V  -- without materialising the [B,12,185,185] score / probability tensors.
//
// Shape regime of ViLT-B/32: N = 185 tokens (<= 256), head dim 64, so ALL keys of one (batch, head)
// fit in LDS (192 x 64 bf16 = 24 KiB per image) and a whole score row (12 key tiles) fits in a wave's
// registers: no online-softmax rescaling is needed.  One workgroup (4 waves) per (batch, head).
//
//  forward  : per 16-query tile S^T = K Q^T (keys on registers, query on the lane), softmax in
//             registers, then O = P V with the S^T accumulators re-used directly as the A operand
//             (permuted k order) and V read with the hardware-transposing ds_read_b64_tr_b16.
//  backward : two kernels, both recompute P from the saved log-sum-exp:
//     dq  kernel (query-parallel): delta = rowsum(P*dP), dS, dQ = dS K          (K via tr-read)
//     dkv kernel (key-parallel)  : dV = P^T dO, dK = dS^T Q on 16x16x16 MFMAs    (Q, dO via tr-read)
// LDS images are filled with 16-byte global_load_lds; bank conflicts are removed by permuting the
// per-lane SOURCE chunk (row images: chunk ^= row&7; transposed-read images: 32-B group ^= (row>>1)&3).
#include "rmcl_common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define SCALE 0.125f
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define SCALE_L2 (SCALE * LOG2E)   // scores are kept in the log2 domain: v_exp_f32 is exp2, so exp(s - m) costs one subtract + one v_exp
int g_attn_fwd_waves = 8;    // waves per workgroup of the forward kernel (rmcl_tune_set key 4: 6, 8 or 12): 12 query tiles -> 1-2 per wave
#define ATT_DQW 8           // dQ kernel: its 96 score/dP accumulators + hoisted fragments need > 256 VGPRs, so one wave per SIMD

// ---- staging: rows [0,N) of a [*, 64] bf16 slice (row pitch ld elements) -> LDS image of NKP rows x 128 B
template <bool TR, int NW = 4>
__device__ __forceinline__ void stage_rows(char* img, const bf16_t* __restrict__ src, long ld, int N, int NKP, int wave, int lane) {
  for (int inst = wave; inst < NKP / 8; inst += NW) {
    const int row = inst * 8 + (lane >> 3), cp = lane & 7;
    int c;
    if (TR) c = (((cp >> 1) ^ ((row >> 1) & 3)) << 1) | (cp & 1);
    else c = cp ^ (row & 7);
    const bf16_t* p = src + (long)min(row, N - 1) * ld + c * 8;
    __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(img + inst * 1024), 16, 0, 0);
  }
}

// One dword of the 128-byte line at p -> L2 (and a 256-byte per-wave LDS sink nobody reads): an LDS-DMA load has no register
// destination, so nothing has to stay reserved while it is in flight; written as inline assembly so that the compiler's waitcnt
// insertion does not see an LDS write it would order the transposed LDS reads of phase 1 behind (vmcnt(0), found in round 3).
// Untracked loads only make later compiler-placed vmcnt(N) waits longer than needed (VMEM loads return in order), never shorter.
__device__ __forceinline__ void touch_line(const void* p, uint32_t sink_lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(sink_lds_addr));
}

__device__ __forceinline__ bf16x8 frag_row_lds(const char* img, int row0, int s, int lane) {
  const int row = row0 + (lane & 15);
  const int chunk = (4 * s + (lane >> 4)) ^ (row & 7);
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + chunk * 16);
}

// row fragment (A[row = lane&15][k = 32 s + 8 (lane>>4) ..]) out of a TRANSPOSED-read image (stage_rows<true>): the image's 32-byte
// group swizzle applied to the 16-byte chunk index; 2-way bank conflict among the 16 rows (rows r, r+8 share a slot)
__device__ __forceinline__ bf16x8 frag_row_ldsT(const char* img, int row0, int s, int lane) {
  const int row = row0 + (lane & 15);
  const int c = 4 * s + (lane >> 4);
  const int cp = (((c >> 1) ^ ((row >> 1) & 3)) << 1) | (c & 1);
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + cp * 16);
}

__device__ __forceinline__ bf16x8 frag_row_global(const bf16_t* __restrict__ src, long ld, int row0, int N, int s, int lane) {
  const int row = min(row0 + (lane & 15), N - 1);
  return *reinterpret_cast<const bf16x8*>(src + (long)row * ld + 32 * s + 8 * (lane >> 4));
}

__device__ __forceinline__ s16x4 tr4(const char* img, int row, int dt, int p) {
  const int t = dt ^ ((row >> 1) & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + row * 128 + t * 32 + p * 8));
}

// B operand for a 32-deep k step whose k order is the accumulator order: k(g,j) = base + 16*(j>>2) + 4g + (j&3)
__device__ __forceinline__ bf16x8 frag_tr32_lds(const char* img, int base, int dt, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = tr4(img, base + 4 * g + q, dt, p);
  u.h[1] = tr4(img, base + 16 + 4 * g + q, dt, p);
  return u.v;
}
// B operand for the 16x16x16 MFMA: rows base + 4g + j
__device__ __forceinline__ s16x4 frag_tr16_lds(const char* img, int base, int dt, int lane) {
  return tr4(img, base + 4 * (lane >> 4) + ((lane & 15) >> 2), dt, lane & 3);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  union { bf16x8 v; bf16_t e[8]; } u;
#pragma unroll
  for (int i = 0; i < 4; ++i) { u.e[i] = f2bf(a[i]); u.e[4 + i] = f2bf(b[i]); }
  return u.v;
}
__device__ __forceinline__ s16x4 pack4(const f32x4& a) {
  union { s16x4 v; bf16_t e[4]; } u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u.e[i] = f2bf(a[i]);
  return u.v;
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups that share lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void fill_maskbias(float* mb, const int* __restrict__ mask, int N, int NKP, int t, int nthreads = 256) {
  for (int j = t; j < NKP; j += nthreads) mb[j] = (j < N && mask[j] != 0) ? 0.f : -INFINITY;
}

// In-kernel phase trace (developer builds: RMCL_EXTRA_FLAGS=-DST_TRACE, tools/st_trace.py attnbwd / attnfwd), as in gemm_st.hip
#ifdef ST_TRACE
__device__ long long g_at_trace[2][32];
#define AT_STAMP(i)                                                                                              \
  if (blockIdx.x == 300 && (threadIdx.x == 0 || threadIdx.x == 320)) g_at_trace[threadIdx.x != 0][i] = wall_clock64()
extern "C" int rmcl_debug_at_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_at_trace), sizeof(long long) * 64);
}
#define AT_STAMPB(i)                                                                                             \
  if (prob == 300 && (threadIdx.x == 0 || threadIdx.x == 320)) g_at_trace[threadIdx.x != 0][i] = wall_clock64()
#else
#define AT_STAMP(i)
#define AT_STAMPB(i)
#endif

// ================================================================================== forward
template <int NKT, int ATT_QW>
__global__ __launch_bounds__(ATT_QW * 64) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ mask,
                                                       bf16_t* __restrict__ out, float* __restrict__ lse, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int NKP = NKT * 16;
  char* Kimg = sm;
  char* Vimg = sm + NKP * 128;
  float* mb = reinterpret_cast<float*>(sm + 2 * NKP * 128);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * 64;
  AT_STAMP(16);
  const long ld = 3 * D;
  const bf16_t* base = qkv + (long)b * N * ld + h * 64;
  // every global load of the load phase is issued before the first one is waited for: the mask row's load is consumed at once
  // (vmcnt(0): the staging DMA has landed by then), and the Q fragments issued behind it were a second, serial HBM round trip
  // (tools/st_trace.py attnfwd: 2.8 us of a 19.8 us workgroup)
  bf16x8 qn[2];                                              // Q fragments of the wave's NEXT tile (global-load latency off the loop's critical path)
  qn[0] = frag_row_global(base, ld, wave * 16, N, 0, lane);
  qn[1] = frag_row_global(base, ld, wave * 16, N, 1, lane);
  stage_rows<false, ATT_QW>(Kimg, base + D, ld, N, NKP, wave, lane);
  stage_rows<true, ATT_QW>(Vimg, base + 2 * D, ld, N, NKP, wave, lane);
  fill_maskbias(mb, mask + (long)b * N, N, NKP, t, ATT_QW * 64);
  AT_STAMP(17);
  const int nqt = (N + 15) / 16;
  AT_STAMP(18);
  __syncthreads();
  AT_STAMP(19);
  for (int qt = wave; qt < nqt; qt += ATT_QW) {
    bf16x8 qf[2] = {qn[0], qn[1]};
    if (qt + ATT_QW < nqt) {
      qn[0] = frag_row_global(base, ld, (qt + ATT_QW) * 16, N, 0, lane);
      qn[1] = frag_row_global(base, ld, (qt + ATT_QW) * 16, N, 1, lane);
    }
    f32x4 S[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) S[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row_lds(Kimg, kt * 16, s, lane), qf[s], S[kt], 0, 0, 0);
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);         // at most two key tiles of fragment reads live: fewer VGPRs, more resident waves
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 bias = *reinterpret_cast<const float4*>(mb + kt * 16 + 4 * g);   // 0 or -inf: the same in either log base
      S[kt][0] = fmaf(S[kt][0], SCALE_L2, bias.x); S[kt][1] = fmaf(S[kt][1], SCALE_L2, bias.y);
      S[kt][2] = fmaf(S[kt][2], SCALE_L2, bias.z); S[kt][3] = fmaf(S[kt][3], SCALE_L2, bias.w);
      m = fmaxf(m, fmaxf(fmaxf(S[kt][0], S[kt][1]), fmaxf(S[kt][2], S[kt][3])));
    }
    m = group_max(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[kt][r] = __builtin_amdgcn_exp2f(S[kt][r] - m); l += S[kt][r]; }
    l = group_sum(l);
    const int q_lane = qt * 16 + (lane & 15);
    if (g == 0 && q_lane < N) lse[((long)blockIdx.x) * NKP + q_lane] = m * LN2 + __logf(l);   // natural-log lse, as the backward expects
    // O^T = V^T P^T: swapped operands leave the lane with 4 adjacent head dims of query lane%16 -> 8-byte stores, and the
    // row's 1/l is already in the lane
    f32x4 O[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      const bf16x8 pa = pack8(S[2 * u], S[2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr32_lds(Vimg, 32 * u, dt, lane), pa, O[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    AT_STAMP(20 + (qt >= ATT_QW ? 2 : 0));
    const float linv = 1.0f / l;
    if (q_lane < N) {
      bf16_t* o = out + ((long)b * N + q_lane) * D + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk;
        pk.x = (uint32_t)f2bf(O[dt][0] * linv) | ((uint32_t)f2bf(O[dt][1] * linv) << 16);
        pk.y = (uint32_t)f2bf(O[dt][2] * linv) | ((uint32_t)f2bf(O[dt][3] * linv) << 16);
        *reinterpret_cast<uint2*>(o + 16 * dt) = pk;
      }
    }
  }
#ifdef ST_TRACE
  AT_STAMP(24);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AT_STAMP(25);
#endif
}

// ================================================================================== backward: dQ, delta
template <int NKT>
__global__ __launch_bounds__(ATT_DQW * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ mask,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int NKP = NKT * 16;
  char* Krow = sm;
  char* Vrow = sm + NKP * 128;
  char* Ktr = sm + 2 * NKP * 128;
  float* mb = reinterpret_cast<float*>(sm + 3 * NKP * 128);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * 64;
  const long ld = 3 * D;
  const bf16_t* base = qkv + (long)b * N * ld + h * 64;
  const bf16_t* dob = dout + (long)b * N * D + h * 64;
  stage_rows<false, ATT_DQW>(Krow, base + D, ld, N, NKP, wave, lane);
  stage_rows<false, ATT_DQW>(Vrow, base + 2 * D, ld, N, NKP, wave, lane);
  stage_rows<true, ATT_DQW>(Ktr, base + D, ld, N, NKP, wave, lane);
  fill_maskbias(mb, mask + (long)b * N, N, NKP, t, ATT_DQW * 64);

  const int nqt = (N + 15) / 16;
  bf16x8 qn[2], dn[2];                                       // Q / dO fragments and lse of the wave's NEXT tile
  float Ln;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    qn[s] = frag_row_global(base, ld, wave * 16, N, s, lane);
    dn[s] = frag_row_global(dob, D, wave * 16, N, s, lane);
  }
  Ln = (wave * 16 + (lane & 15)) < N ? lse[((long)blockIdx.x) * NKP + wave * 16 + (lane & 15)] : INFINITY;
  __syncthreads();
  for (int qt = wave; qt < nqt; qt += ATT_DQW) {
    bf16x8 qf[2] = {qn[0], qn[1]}, df[2] = {dn[0], dn[1]};
    const float L = Ln * LOG2E;                                 // (+inf for pad rows stays +inf)
    const int q_lane = qt * 16 + (lane & 15);
    if (qt + ATT_DQW < nqt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        qn[s] = frag_row_global(base, ld, (qt + ATT_DQW) * 16, N, s, lane);
        dn[s] = frag_row_global(dob, D, (qt + ATT_DQW) * 16, N, s, lane);
      }
      Ln = (q_lane + 16 * ATT_DQW) < N ? lse[((long)blockIdx.x) * NKP + q_lane + 16 * ATT_DQW] : INFINITY;
    }
    f32x4 S[NKT], dP[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dP[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        S[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row_lds(Krow, kt * 16, s, lane), qf[s], S[kt], 0, 0, 0);
        dP[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row_lds(Vrow, kt * 16, s, lane), df[s], dP[kt], 0, 0, 0);
      }
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);         // keeps the fragment reads of at most two key tiles live (8 waves: 256 VGPRs)
    }
    float dl = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 bias = *reinterpret_cast<const float4*>(mb + kt * 16 + 4 * g);
      const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[kt][r] = __builtin_amdgcn_exp2f(fmaf(S[kt][r], SCALE_L2, bb[r] - L));   // P
        dl += S[kt][r] * dP[kt][r];
      }
    }
    dl = group_sum(dl);
    if (g == 0 && q_lane < N) delta[((long)blockIdx.x) * NKP + q_lane] = dl;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[kt][r] = S[kt][r] * (dP[kt][r] - dl) * SCALE;   // dS
    f32x4 dQ[4];                                             // dQ^T = K^T dS^T (swapped operands: 4 adjacent head dims per lane)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      const bf16x8 sa = pack8(S[2 * u], S[2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dQ[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr32_lds(Ktr, 32 * u, dt, lane), sa, dQ[dt], 0, 0, 0);
    }
    if (q_lane < N) {
      bf16_t* o = dqkv + ((long)b * N + q_lane) * ld + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk;
        pk.x = (uint32_t)f2bf(dQ[dt][0]) | ((uint32_t)f2bf(dQ[dt][1]) << 16);
        pk.y = (uint32_t)f2bf(dQ[dt][2]) | ((uint32_t)f2bf(dQ[dt][3]) << 16);
        *reinterpret_cast<uint2*>(o + 16 * dt) = pk;
      }
    }
  }
}

// ================================================================================== backward: dK, dV
// NW waves per workgroup, each owning NKT / NW key tiles
// (NW = 4 everywhere; see the launcher).
template <int NKT, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ mask,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int NKP = NKT * 16, TPW = NKT / NW;
  char* Qtr = sm;
  char* Dtr = sm + NKP * 128;
  float* Ls = reinterpret_cast<float*>(sm + 2 * NKP * 128);   // lse per query (+inf for pad rows)
  float* Ds = Ls + NKP;                                       // delta per query
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * 64;
  const long ld = 3 * D;
  const bf16_t* base = qkv + (long)b * N * ld + h * 64;
  const bf16_t* dob = dout + (long)b * N * D + h * 64;
  stage_rows<true, NW>(Qtr, base, ld, N, NKP, wave, lane);
  stage_rows<true, NW>(Dtr, dob, D, N, NKP, wave, lane);
  for (int j = t; j < NKP; j += NW * 64) {
    Ls[j] = j < N ? lse[((long)blockIdx.x) * NKP + j] * LOG2E : INFINITY;   // log2 domain
    Ds[j] = j < N ? delta[((long)blockIdx.x) * NKP + j] : 0.f;
  }
  // this wave's key tiles: kt = wave + NW*i ; K / V fragments and key mask stay in registers
  bf16x8 kf[TPW][2], vf[TPW][2];
  float mbk[TPW];
  const int* mrow = mask + (long)b * N;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + NW * i, key = kt * 16 + (lane & 15);
    mbk[i] = (key < N && mrow[key] != 0) ? 0.f : -INFINITY;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kf[i][s] = frag_row_global(base + D, ld, kt * 16, N, s, lane);
      vf[i][s] = frag_row_global(base + 2 * D, ld, kt * 16, N, s, lane);
    }
  }
  f32x4 dK[TPW][4], dV[TPW][4];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dK[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();

  const int nqt = (N + 15) / 16;
  bf16x8 qn[2], dn[2];                                       // row-layout Q / dO fragments of the NEXT query tile
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    qn[s] = frag_row_global(base, ld, 0, N, s, lane);
    dn[s] = frag_row_global(dob, D, 0, N, s, lane);
  }
  for (int qt = 0; qt < nqt; ++qt) {
    bf16x8 qf[2] = {qn[0], qn[1]}, df[2] = {dn[0], dn[1]};
    if (qt + 1 < nqt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        qn[s] = frag_row_global(base, ld, (qt + 1) * 16, N, s, lane);
        dn[s] = frag_row_global(dob, D, (qt + 1) * 16, N, s, lane);
      }
    }
    const float4 L4 = *reinterpret_cast<const float4*>(Ls + qt * 16 + 4 * g);
    const float4 D4 = *reinterpret_cast<const float4*>(Ds + qt * 16 + 4 * g);
    const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
    s16x4 dot[4], qtr[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dot[dt] = frag_tr16_lds(Dtr, qt * 16, dt, lane);
      qtr[dt] = frag_tr16_lds(Qtr, qt * 16, dt, lane);
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf[i][s], S, 0, 0, 0);        // S[q=4g+r][key=lane&15]
        dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[s], vf[i][s], dP, 0, 0, 0);
      }
      f32x4 P, dS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[r] = __builtin_amdgcn_exp2f(fmaf(S[r], SCALE_L2, mbk[i] - Lr[r]));
        dS[r] = P[r] * (dP[r] - Dr[r]) * SCALE;
      }
      const s16x4 pa = pack4(P), sa = pack4(dS);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {                       // swapped operands: dV^T[d = 4g+r][key = lane%16]
        dV[i][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dot[dt], pa, dV[i][dt], 0, 0, 0);   // += dO^T P
        dK[i][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qtr[dt], sa, dK[i][dt], 0, 0, 0);   // += Q^T dS
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int key = (wave + NW * i) * 16 + (lane & 15);
    if (key < N) {
      bf16_t* o = dqkv + ((long)b * N + key) * ld + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk, pv;
        pk.x = (uint32_t)f2bf(dK[i][dt][0]) | ((uint32_t)f2bf(dK[i][dt][1]) << 16);
        pk.y = (uint32_t)f2bf(dK[i][dt][2]) | ((uint32_t)f2bf(dK[i][dt][3]) << 16);
        pv.x = (uint32_t)f2bf(dV[i][dt][0]) | ((uint32_t)f2bf(dV[i][dt][1]) << 16);
        pv.y = (uint32_t)f2bf(dV[i][dt][2]) | ((uint32_t)f2bf(dV[i][dt][3]) << 16);
        *reinterpret_cast<uint2*>(o + D + 16 * dt) = pk;
        *reinterpret_cast<uint2*>(o + 2 * D + 16 * dt) = pv;
      }
    }
  }
}


// ================================================================================== backward: ONE kernel (dQ, dK, dV)
// One workgroup of NKT waves per (batch, head); wave w owns KEY tile w in phase 1 and QUERY tile w in phase 2.
//   phase 0: Q, dO, K -> LDS (transposed-read images); K / V row fragments of the wave's key tile -> registers;
//            delta[q] = sum_d dO[q][d] * O[q][d]  (= rowsum(P * dP); O is the forward's bf16 output) for query tile w
//   phase 1: for every query tile: S, dP on the MFMA (once - the two-kernel form computed them twice), P and dS in
//            registers, dV += P^T dO, dK += dS^T Q (16x16x16 MFMAs, accumulators stay in the wave), and the dS^T tile
//            -> LDS image [key][q] (bf16, written as the transposed-read layout of the 64-column sub-image q / 64)
//   phase 2: dQ^T = K^T dS^T for the wave's query tile: both operands by ds_read_b64_tr_b16, same permuted k order.
// Nothing but dqkv is written to HBM; qkv / dO / O are read once per workgroup (+ L1-resident fragment re-reads).
template <int NKT, int TPW>
__global__ __launch_bounds__(NKT / TPW * 64) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ mask,
                                                                       const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                                       const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int H, int P) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int NKP = NKT * 16, NW = NKT / TPW, NSUB = (NKP + 63) / 64;
  static_assert(NKT % TPW == 0, "key tiles per wave must divide the tile count");
  char* Qtr = sm;
  char* Dtr = sm + NKP * 128;
  char* Ktr = sm + 2 * NKP * 128;
  char* dSimg = sm + 3 * NKP * 128;                             // NSUB sub-images of [NKP keys][64 q] bf16 (128-B rows)
  float* Ls = reinterpret_cast<float*>(sm + (3 + NSUB) * NKP * 128);   // lse per query, log2 domain (+inf for pad rows)
  float* Ds = Ls + NKP;                                         // delta per query
  char* touch_sink = reinterpret_cast<char*>(Ds + NKP);         // 256 B per wave: where the next problem's touch loads land (never read)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
  const int D = H * 64;
  const long ld = 3 * D;
  // One workgroup per CU walks problems p, p + grid, ... (grid == P: one problem each).  The load phase of a problem is a burst of
  // ~118 KB per CU that every CU issues at the same moment and then waits for; while phase 1 computes, each thread touches a few
  // 128-byte lines of the NEXT problem's operands, so that the next burst is served by L2 / MALL instead of HBM.
  // Wave w owns key tiles (and, in phase 2, query tiles) w * TPW .. w * TPW + TPW - 1: every Q / dO fragment it reads from LDS in
  // phase 1 serves TPW key tiles (phase 1 is LDS-bandwidth bound: each wave reads all of Q and dO in two layouts).
  for (int prob = blockIdx.x; prob < P; prob += gridDim.x) {
  const int b = prob / H, h = prob % H;
  AT_STAMPB(0);
  const bf16_t* base = qkv + (long)b * N * ld + h * 64;
  const bf16_t* dob = dout + (long)b * N * D + h * 64;
  const bf16_t* ob = out + (long)b * N * D + h * 64;
  // the wave's key tiles: K / V row fragments and the key mask stay in registers.  Issued FIRST: behind the lse load below (consumed at
  // once, i.e. after a vmcnt(0) that also waits for the staging) they were a second, serial HBM round trip of the load phase
  bf16x8 kf[TPW][2], vf[TPW][2], of[TPW][2];
  const int* mrow = mask + (long)b * N;
  int mkey[TPW];                                                // consumed after the barrier: a branch around the load made it a serial round trip
#pragma unroll
  for (int i = 0; i < TPW; ++i) mkey[i] = mrow[min((wave * TPW + i) * 16 + (lane & 15), N - 1)];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) vf[i][s] = frag_row_global(base + 2 * D, ld, (wave * TPW + i) * 16, N, s, lane);   // (K: out of its LDS image, below)
  // delta of the wave's query tiles: O row fragments from global now, dO out of its LDS image after the barrier (dO is staged for the
  // transposed reads anyway: no second global read of it)
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) of[i][s] = frag_row_global(ob, D, (wave * TPW + i) * 16, N, s, lane);
  stage_rows<true, NW>(Qtr, base, ld, N, NKP, wave, lane);
  stage_rows<true, NW>(Dtr, dob, D, N, NKP, wave, lane);
  stage_rows<true, NW>(Ktr, base + D, ld, N, NKP, wave, lane);
  AT_STAMPB(1);
  for (int j = t; j < NKP; j += NW * 64) Ls[j] = j < N ? lse[((long)prob) * NKP + j] * LOG2E : INFINITY;
  AT_STAMPB(2);
  f32x4 dK[TPW][4], dV[TPW][4];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dK[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  AT_STAMPB(3);
  __syncthreads();
  AT_STAMPB(4);
  float mbk[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int key = (wave * TPW + i) * 16 + (lane & 15);
    mbk[i] = (key < N && mkey[i] != 0) ? 0.f : -INFINITY;
#pragma unroll
    for (int s = 0; s < 2; ++s) kf[i][s] = frag_row_ldsT(Ktr, (wave * TPW + i) * 16, s, lane);   // the K image is staged anyway (phase 2): no second global read of K
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; uint32_t w[4]; } a, c;
      a.v = frag_row_ldsT(Dtr, (wave * TPW + i) * 16, s, lane);
      c.v = of[i][s];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dl = fmaf(__uint_as_float(a.w[e] << 16), __uint_as_float(c.w[e] << 16), dl);
        dl = fmaf(__uint_as_float(a.w[e] & 0xffff0000u), __uint_as_float(c.w[e] & 0xffff0000u), dl);
      }
    }
    dl = group_sum(dl);
    if (g == 0) Ds[key] = key < N ? dl : 0.f;
  }
  __syncthreads();                                             // delta of every query tile is in LDS
  if (prob + (int)gridDim.x < P) {
    const int pn = prob + gridDim.x, bn = pn / H, hn = pn % H;
    const bf16_t* basen = qkv + (long)bn * N * ld + hn * 64;
    const bf16_t* dobn = dout + (long)bn * N * D + hn * 64;
    const bf16_t* obn = out + (long)bn * N * D + hn * 64;
    const uint32_t sink = __builtin_amdgcn_readfirstlane(lds_addr(touch_sink + wave * 256));
#pragma unroll
    for (int i = 0; i < (5 * NKP + NW * 64 - 1) / (NW * 64); ++i) {   // 5 N lines: Q, K, V, dO, O rows of 128 bytes (the tail re-touches the last line)
      const int l = min(t + i * NW * 64, 5 * N - 1), r = l / 5, w = l - 5 * r;
      const bf16_t* a = w < 3 ? basen + (long)r * ld + w * D : (w == 3 ? dobn : obn) + (long)r * D;
      touch_line(a, sink);
    }
  }

  // ---- phase 1 -------------------------------------------------------------------------------------------------------
  for (int qt = 0; qt < NKT; ++qt) {
    // Q / dO row fragments out of the SAME LDS images the transposed reads use (a per-iteration global load would put an
    // L2 round trip on every one of the 12 short iterations)
    bf16x8 qf[2], df[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qf[s] = frag_row_ldsT(Qtr, qt * 16, s, lane);
      df[s] = frag_row_ldsT(Dtr, qt * 16, s, lane);
    }
    const float4 L4 = *reinterpret_cast<const float4*>(Ls + qt * 16 + 4 * g);
    const float4 D4 = *reinterpret_cast<const float4*>(Ds + qt * 16 + 4 * g);
    const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
    s16x4 dot[4], qtr[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dot[dt] = frag_tr16_lds(Dtr, qt * 16, dt, lane);
      qtr[dt] = frag_tr16_lds(Qtr, qt * 16, dt, lane);
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf[i][s], S, 0, 0, 0);        // S[q = 4g+r][key = lane&15]
        dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[s], vf[i][s], dP, 0, 0, 0);
      }
      f32x4 Pr, dS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Pr[r] = __builtin_amdgcn_exp2f(fmaf(S[r], SCALE_L2, mbk[i] - Lr[r]));
        dS[r] = Pr[r] * (dP[r] - Dr[r]) * SCALE;
      }
      const s16x4 pa = pack4(Pr), sa = pack4(dS);
      {                                                        // dS^T tile -> image [key][q]: 4 adjacent q of key lane&15, 8 bytes
        const int row = (wave * TPW + i) * 16 + (lane & 15);
        char* img = dSimg + (qt >> 2) * (NKP * 128);
        *reinterpret_cast<s16x4*>(img + row * 128 + (((qt & 3) ^ ((row >> 1) & 3)) * 32) + g * 8) = sa;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {                         // swapped operands: dV^T[d = 4g+r][key = lane&15]
        dV[i][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dot[dt], pa, dV[i][dt], 0, 0, 0);   // += dO^T P
        dK[i][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qtr[dt], sa, dK[i][dt], 0, 0, 0);   // += Q^T dS
      }
    }
  }
  AT_STAMPB(5);
  __syncthreads();                                             // every wave is done with the Q / dO images; dS^T is complete
  AT_STAMPB(6);
  // dK / dV: the accumulators hold 4 adjacent d of one key per lane (8 bytes; a direct store touches 32 bytes of each of 16 rows per
  // instruction: 3.3 us of a 17 us problem went into issuing them).  Through the wave's OWN rows of the dead Q / dO images instead
  // (row-major, 16-byte chunk c at c ^ (row & 7)), read back as whole 128-byte rows: 4 full-line stores per tile; phase 2 runs under them.
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int row = (wave * TPW + i) * 16 + (lane & 15);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk, pv;
      pk.x = (uint32_t)f2bf(dK[i][dt][0]) | ((uint32_t)f2bf(dK[i][dt][1]) << 16);
      pk.y = (uint32_t)f2bf(dK[i][dt][2]) | ((uint32_t)f2bf(dK[i][dt][3]) << 16);
      pv.x = (uint32_t)f2bf(dV[i][dt][0]) | ((uint32_t)f2bf(dV[i][dt][1]) << 16);
      pv.y = (uint32_t)f2bf(dV[i][dt][2]) | ((uint32_t)f2bf(dV[i][dt][3]) << 16);
      const int off = row * 128 + (((2 * dt + (g >> 1)) ^ (row & 7)) * 16) + (g & 1) * 8;
      *reinterpret_cast<uint2*>(Qtr + off) = pk;
      *reinterpret_cast<uint2*>(Dtr + off) = pv;
    }
  }
  asm volatile("" ::: "memory");                               // (same wave, LDS operations complete in order: no barrier)
#pragma unroll
  for (int it = 0; it < 2 * TPW; ++it) {
    const int r = wave * TPW * 16 + it * 8 + (lane >> 3), c = lane & 7;
    const int off = r * 128 + ((c ^ (r & 7)) * 16);
    const uint2 k0 = *reinterpret_cast<const uint2*>(Qtr + off), k1 = *reinterpret_cast<const uint2*>(Qtr + off + 8);
    const uint2 v0 = *reinterpret_cast<const uint2*>(Dtr + off), v1 = *reinterpret_cast<const uint2*>(Dtr + off + 8);
    if (r < N) {
      bf16_t* o = dqkv + ((long)b * N + r) * ld + h * 64 + c * 8;
      *reinterpret_cast<uint4*>(o + D) = uint4{k0.x, k0.y, k1.x, k1.y};
      *reinterpret_cast<uint4*>(o + 2 * D) = uint4{v0.x, v0.y, v1.x, v1.y};
    }
  }
  AT_STAMPB(7);

  // ---- phase 2: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] for the wave's query tiles: a K^T fragment serves all TPW of them ------
  {
    f32x4 dQ[TPW][4];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dQ[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      bf16x8 sb[TPW];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int qt = wave * TPW + i;
        sb[i] = frag_tr32_lds(dSimg + (qt >> 2) * (NKP * 128), 32 * u, qt & 3, lane);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 ka = frag_tr32_lds(Ktr, 32 * u, dt, lane);
#pragma unroll
        for (int i = 0; i < TPW; ++i) dQ[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sb[i], dQ[i][dt], 0, 0, 0);
      }
    }
    AT_STAMPB(8);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int q_lane = (wave * TPW + i) * 16 + (lane & 15);
      if (q_lane < N) {
        bf16_t* o = dqkv + ((long)b * N + q_lane) * ld + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(dQ[i][dt][0]) | ((uint32_t)f2bf(dQ[i][dt][1]) << 16);
          pk.y = (uint32_t)f2bf(dQ[i][dt][2]) | ((uint32_t)f2bf(dQ[i][dt][3]) << 16);
          *reinterpret_cast<uint2*>(o + 16 * dt) = pk;
        }
      }
    }
  }
#ifdef ST_TRACE
  AT_STAMPB(9);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AT_STAMPB(10);
#endif
  __syncthreads();                                             // phase 2 has read Ktr / dS^T: the next problem may stage
  }
}

// ================================================================================== launchers
static inline int nkt_for(int N) { return N <= 64 ? 4 : N <= 128 ? 8 : N <= 192 ? 12 : 16; }

long rmcl_attn_stat_elems(int B, int H, int N) { return (long)B * H * nkt_for(N) * 16; }

template <int NKT, int QW> static int launch_fwd_w(const bf16_t* qkv, const int* mask, bf16_t* out, float* lse, int B, int N, int H, hipStream_t s) {
  const size_t lds = (size_t)2 * NKT * 16 * 128 + NKT * 16 * 4;
  static RmclLdsOnce once;                                     // (once per instantiation and device: the call costs the host ~170 us)
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>((attn_fwd_kernel<NKT, QW>)), (int)lds));
  RMCL_LAUNCH((attn_fwd_kernel<NKT, QW>), dim3(B * H), dim3(QW * 64), lds, s, qkv, mask, out, lse, N, H);
  RMCL_CHECK_LAUNCH();
  return 0;
}
template <int NKT> static int launch_fwd(const bf16_t* qkv, const int* mask, bf16_t* out, float* lse, int B, int N, int H, hipStream_t s) {
  if (g_attn_fwd_waves == 6) return launch_fwd_w<NKT, 6>(qkv, mask, out, lse, B, N, H, s);
  if (g_attn_fwd_waves == 12) return launch_fwd_w<NKT, 12>(qkv, mask, out, lse, B, N, H, s);
  if (g_attn_fwd_waves == 4) return launch_fwd_w<NKT, 4>(qkv, mask, out, lse, B, N, H, s);
  return launch_fwd_w<NKT, 8>(qkv, mask, out, lse, B, N, H, s);
}
int g_attn_bwd_tpw = 1;              // rmcl_tune_set key 9: key tiles per wave of the fused backward (1, 2 or 3)
int g_attn_bwd_persist = 0;          // rmcl_tune_set key 8: workgroups of the fused backward (each walks problems p, p + grid, ...); 0: one per problem
bool g_attn_fused_bwd = true;        // rmcl_tune_set key 2: 0 selects the two-kernel backward (A/B and parity tests)

template <int NKT, int TPW> static int launch_bwd_fused_t(const bf16_t* qkv, const int* mask, const bf16_t* dout, const bf16_t* out, const float* lse,
                                                          bf16_t* dqkv, int B, int N, int H, hipStream_t s) {
  constexpr int NKP = NKT * 16, NSUB = (NKP + 63) / 64, NW = NKT / TPW;
  const size_t lds = (size_t)(3 + NSUB) * NKP * 128 + 2 * NKP * 4 + NW * 256;
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(attn_bwd_fused_kernel<NKT, TPW>), (int)lds));
  const int P = B * H, grid = g_attn_bwd_persist > 0 ? min(P, g_attn_bwd_persist) : P;
  RMCL_LAUNCH((attn_bwd_fused_kernel<NKT, TPW>), dim3(grid), dim3(NW * 64), lds, s, qkv, mask, dout, out, lse, dqkv, N, H, P);
  RMCL_CHECK_LAUNCH();
  return 0;
}
template <int NKT> static int launch_bwd_fused(const bf16_t* qkv, const int* mask, const bf16_t* dout, const bf16_t* out, const float* lse,
                                               bf16_t* dqkv, int B, int N, int H, hipStream_t s) {
  if constexpr (NKT % 2 == 0) if (g_attn_bwd_tpw == 2) return launch_bwd_fused_t<NKT, 2>(qkv, mask, dout, out, lse, dqkv, B, N, H, s);
  if constexpr (NKT % 3 == 0) if (g_attn_bwd_tpw == 3) return launch_bwd_fused_t<NKT, 3>(qkv, mask, dout, out, lse, dqkv, B, N, H, s);
  return launch_bwd_fused_t<NKT, 1>(qkv, mask, dout, out, lse, dqkv, B, N, H, s);
}

template <int NKT> static int launch_bwd(const bf16_t* qkv, const int* mask, const bf16_t* dout, const float* lse, float* delta,
                                         bf16_t* dqkv, int B, int N, int H, hipStream_t s) {
  const size_t lds1 = (size_t)3 * NKT * 16 * 128 + NKT * 16 * 4, lds2 = (size_t)2 * NKT * 16 * 128 + 2 * NKT * 16 * 4;
  constexpr int NW = 4;                                      // (6 waves x 2 key tiles at NKT = 12 measured slower: 51.8 vs 44.0 us)
  static RmclLdsOnce once1, once2;
  RMCL_TRY(rmcl_set_max_lds(once1, reinterpret_cast<const void*>(attn_bwd_dq_kernel<NKT>), (int)lds1));
  RMCL_TRY(rmcl_set_max_lds(once2, reinterpret_cast<const void*>((attn_bwd_dkv_kernel<NKT, NW>)), (int)lds2));
  RMCL_LAUNCH(attn_bwd_dq_kernel<NKT>, dim3(B * H), dim3(ATT_DQW * 64), lds1, s, qkv, mask, dout, lse, delta, dqkv, N, H);
  RMCL_CHECK_LAUNCH();
  RMCL_LAUNCH((attn_bwd_dkv_kernel<NKT, NW>), dim3(B * H), dim3(NW * 64), lds2, s, qkv, mask, dout, lse, delta, dqkv, N, H);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_attn_fused_fwd(const void* qkv, const int* mask, void* out, float* lse, int B, int N, int H, hipStream_t s) {
  RMCL_REQUIRE(N >= 1 && N <= 256, "fused attention: N must be <= 256");
  const bf16_t* q = (const bf16_t*)qkv;
  bf16_t* o = (bf16_t*)out;
  switch (nkt_for(N)) {
    case 4: return launch_fwd<4>(q, mask, o, lse, B, N, H, s);
    case 8: return launch_fwd<8>(q, mask, o, lse, B, N, H, s);
    case 12: return launch_fwd<12>(q, mask, o, lse, B, N, H, s);
    default: return launch_fwd<16>(q, mask, o, lse, B, N, H, s);
  }
}

int rmcl_attn_fused_bwd(const void* qkv, const int* mask, const void* dout, const void* out, const float* lse, float* delta, void* dqkv,
                        int B, int N, int H, hipStream_t s) {
  RMCL_REQUIRE(N >= 1 && N <= 256, "fused attention: N must be <= 256");
  const bf16_t* q = (const bf16_t*)qkv;
  const bf16_t* d = (const bf16_t*)dout;
  bf16_t* o = (bf16_t*)dqkv;
  if (out && g_attn_fused_bwd && N <= 192) {                   // one kernel; needs the forward's output O for delta
    const bf16_t* ao = (const bf16_t*)out;
    switch (nkt_for(N)) {
      case 4: return launch_bwd_fused<4>(q, mask, d, ao, lse, o, B, N, H, s);
      case 8: return launch_bwd_fused<8>(q, mask, d, ao, lse, o, B, N, H, s);
      default: return launch_bwd_fused<12>(q, mask, d, ao, lse, o, B, N, H, s);
    }
  }
  switch (nkt_for(N)) {
    case 4: return launch_bwd<4>(q, mask, d, lse, delta, o, B, N, H, s);
    case 8: return launch_bwd<8>(q, mask, d, lse, delta, o, B, N, H, s);
    case 12: return launch_bwd<12>(q, mask, d, lse, delta, o, B, N, H, s);
    default: return launch_bwd<16>(q, mask, d, lse, delta, o, B, N, H, s);
  }
}
