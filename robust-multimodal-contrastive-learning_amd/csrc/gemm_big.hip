// bf16 MFMA GEMM, 256x256x64 block tile (8 waves as 2(M) x 4(N), each 128x64 = 8x4 MFMA tiles, 128
// accumulator registers), 2 LDS stages of 64 KiB, one workgroup per CU.
//
// Why: the global->LDS (LDS-DMA) path delivers ~62 GB/s per CU (measured with loads-only runs of the
// 128x128 kernel); 256x256 tiles move half the bytes per FLOP.  Operand layouts, swizzles and the
// swapped-operand epilogue are those of gemm_fast.hip.  Inside a k-tile the work is cut into 4 phases
// (k-step x row-half): the fragment reads of phase p+1 are issued before the MFMAs of phase p (register
// double buffering) and a quarter of the next k-tile's global_load_lds instructions is issued per phase.
// Used where it measures faster than the 128x128 kernel (N = 768 outputs with K >= 2304; see
// rmcl_launch_gemm_fast).  A 4-buffer half-tile ring with counted vmcnt across raw barriers was tried on
// this tile as well and measured 10-25 % SLOWER than this simple form (GEMM notes).
#include "rmcl_common.h"
#include "kernels.h"

#define GBM2 256
#define GBN2 256
#define GBK2 64
#define G_A_BYTES (GBM2 * GBK2 * 2)                 // 32 KiB
#define G_STAGE_BYTES ((GBM2 + GBN2) * GBK2 * 2)    // 64 KiB
#define GROUP_M2 4

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int kswz2(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// one quarter (q = 0..3) of this wave's global_load_lds work for one 256 x 64 operand tile (32 instr / 8 waves = 4 per wave)
template <bool KC>
__device__ __forceinline__ void big_stage_quarter(const bf16_t* __restrict__ base, long ld, int r0, int Rmax, int k0, char* lds_tile,
                                                  int wave, int lane, int q) {
  const int inst = wave * 4 + q;
  const bf16_t* src;
  if (KC) {
    const int row = inst * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    src = base + (long)min(r0 + row, Rmax - 1) * ld + k0 + chunk * 8;
  } else {
    const int k = inst * 2 + (lane >> 5);            // 512-B k-rows: 2 per instruction
    const int chunk = (lane & 31) ^ kswz2(k);
    src = base + (long)(k0 + k) * ld + r0 + chunk * 8;
  }
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
}

template <bool KC>
__device__ __forceinline__ bf16x8 big_load_frag(const char* lds_tile, int t0, int s, int lane) {
  if (KC) {
    const int row = t0 + (lane & 15);
    const int chunk = (4 * s + (lane >> 4)) ^ (row & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + chunk * 16);
  } else {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int c8 = (t0 >> 2) + p;
    union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * s + 8 * (lane >> 4) + 4 * h + q;
      const int c16 = (c8 >> 1) ^ kswz2(k);
      const char* a = lds_tile + k * 512 + c16 * 16 + (c8 & 1) * 8;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    return u.v;
  }
}

template <bool A_KC, bool B_KC, typename TO>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int m0, n0;
  {
    const int per_group = GROUP_M2 * tiles_n, group = bid / per_group, first_m = group * GROUP_M2;
    const int gsz = min(tiles_m - first_m, GROUP_M2), r = bid - group * per_group;
    m0 = (first_m + r % gsz) * GBM2;
    n0 = (r / gsz) * GBN2;
  }
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  int kbeg = 0, kend = g.K;
  long zoff = 0;
  if (g.splitk > 1) {
    const int per = ((g.K / GBK2 + g.splitk - 1) / g.splitk) * GBK2;
    kbeg = blockIdx.y * per;
    kend = min(g.K, kbeg + per);
    zoff = (long)blockIdx.y * g.M * g.ldc;
  }
  const int nk = max(0, (kend - kbeg) / GBK2);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      big_stage_quarter<A_KC>(A, g.lda, m0, g.M, kbeg, smem, wave, lane, q);
      big_stage_quarter<B_KC>(B, g.ldb, n0, g.N, kbeg, smem + G_A_BYTES, wave, lane, q);
    }
  }
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const char* at = smem + (it & 1) * G_STAGE_BYTES;
    const char* bt = at + G_A_BYTES;
    char* nx = smem + ((it + 1) & 1) * G_STAGE_BYTES;
    const bool pre = it + 1 < nk;
    const int kn = kbeg + (it + 1) * GBK2;
    // phases p = (s, half): k-step s, row half (4 of the wave's 8 row tiles)
    bf16x8 af[2][4], bf[2][4];                               // af: per phase (double buffered); bf: per k-step
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = big_load_frag<A_KC>(at, wm * 128 + i * 16, 0, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = big_load_frag<B_KC>(bt, wn * 64 + j * 16, 0, lane);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int cb = p & 1, nb = cb ^ 1, s0 = p >> 1;
      if (pre) {                                             // a quarter of the next k-tile's DMA per phase
        big_stage_quarter<A_KC>(A, g.lda, m0, g.M, kn, nx, wave, lane, p);
        big_stage_quarter<B_KC>(B, g.ldb, n0, g.N, kn, nx + G_A_BYTES, wave, lane, p);
      }
      if (p < 3) {                                           // fragments of phase p+1
        const int s1 = (p + 1) >> 1, h1 = (p + 1) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[nb][i] = big_load_frag<A_KC>(at, wm * 128 + (h1 * 4 + i) * 16, s1, lane);
        if (h1 == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[s1][j] = big_load_frag<B_KC>(bt, wn * 64 + j * 16, s1, lane);
        }
      }
      const int h0 = p & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[h0 * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s0][j], af[cb][i], acc[h0 * 4 + i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C) + zoff;
  TO* C2 = reinterpret_cast<TO*>(g.C2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      float v[4] = {g.alpha * acc[i][j][0], g.alpha * acc[i][j][1], g.alpha * acc[i][j][2], g.alpha * acc[i][j][3]};
      if (epi & EPI_BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (epi & EPI_DGELU) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + (long)m * g.ld_aux + n);
        v[0] *= gelu_fast_grad(__uint_as_float(u.x << 16)); v[1] *= gelu_fast_grad(__uint_as_float(u.x & 0xffff0000u));
        v[2] *= gelu_fast_grad(__uint_as_float(u.y << 16)); v[3] *= gelu_fast_grad(__uint_as_float(u.y & 0xffff0000u));
      }
      const long ci = (long)m * g.ldc + n;
      if (epi & EPI_SAVE_PREACT) {
        if constexpr (sizeof(TO) == 2) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(C2 + ci) = pk;
        } else {
          *reinterpret_cast<float4*>(C2 + ci) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      if (epi & EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
      }
      if (epi & EPI_RESIDUAL) {
        const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + (long)m * g.ld_aux + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      if constexpr (sizeof(TO) == 2) {
        uint2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(C + ci) = pk;
      } else {
        if (epi & EPI_ACCUM) {
          const float4 o = *reinterpret_cast<const float4*>(C + ci);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        *reinterpret_cast<float4*>(C + ci) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

bool rmcl_gemm_big_supported(const GemmArgs& g, int a_kc, int b_kc) {
  if (g.N % GBN2 != 0 || g.K % GBK2 != 0 || g.K < GBK2) return false;
  if (!a_kc && g.M % GBM2 != 0) return false;
  if (g.epi & (EPI_DROPOUT | EPI_DROP_BWD)) return false;
  return true;
}

template <bool A_KC, bool B_KC>
static int launch_big(const GemmArgs& g, int dt_out, hipStream_t s) {
  static RmclLdsOnce once_f, once_b;
  RMCL_TRY(rmcl_set_max_lds(once_f, reinterpret_cast<const void*>(gemm_big_kernel<A_KC, B_KC, float>), 2 * G_STAGE_BYTES));
  RMCL_TRY(rmcl_set_max_lds(once_b, reinterpret_cast<const void*>(gemm_big_kernel<A_KC, B_KC, bf16_t>), 2 * G_STAGE_BYTES));
  const int tm = cdiv(g.M, GBM2), tn = g.N / GBN2;
  dim3 grid(tm * tn, g.splitk > 1 ? g.splitk : 1);
  if (dt_out == RMCL_F32) RMCL_LAUNCH((gemm_big_kernel<A_KC, B_KC, float>), grid, dim3(512), 2 * G_STAGE_BYTES, s, g, tm, tn);
  else RMCL_LAUNCH((gemm_big_kernel<A_KC, B_KC, bf16_t>), grid, dim3(512), 2 * G_STAGE_BYTES, s, g, tm, tn);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_launch_gemm_big(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s) {
  if (a_kc && b_kc) return launch_big<true, true>(g, dt_out, s);
  if (a_kc && !b_kc) return launch_big<true, false>(g, dt_out, s);
  return launch_big<false, false>(g, dt_out, s);
}
