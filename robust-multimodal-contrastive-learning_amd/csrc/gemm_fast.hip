// bf16 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) - the workhorse of the
// encoder forward (NT), data-gradient (NN) and weight-gradient (TN) passes.
//
// Block tile (WM*64) x (WN*64) x 64, one wave per 64x64 sub-tile (4x4 MFMA tiles, 64 accumulator
// registers).  Staging: global -> LDS directly with 16-byte global_load_lds (no VGPR round trip)
// into a ring of STAGES LDS buffers.  STAGES == 2: one __syncthreads per k-tile (the loads of tile
// t+1 overlap the MFMAs of tile t).  STAGES == 3: raw s_barrier + COUNTED s_waitcnt vmcnt(N) so the
// loads of tile t+2 stay in flight across the barrier (guide "Pipelining across barriers").
// The LDS image is lane-linear as the instruction requires; bank conflicts are removed by permuting
// the per-lane SOURCE address and applying the same involution on the fragment reads (rule 21 / T2):
//   * k-contiguous operands ([rows][64 k], 128-B rows): 16-B chunk index ^= (row & 7);
//     fragments are one ds_read_b128 per lane (A[row=l&15][k=8(l>>4)+j]).
//   * m/n-contiguous operands ([64 k][R cols]; the W of dX = dY W and both operands of
//     dW = dY^T X): 16-B chunk index ^= f(k) << 1, f(k) = (k&3) | ((k>>3)&1)<<2;
//     fragments are two ds_read_b64_tr_b16 (hardware transpose) per lane - no transposed
//     copies of weights or activations exist anywhere.
// The MFMA is issued with the operands swapped (D = W_frag x X_frag) so each lane ends up with 4
// CONSECUTIVE output columns of one row: the epilogue (bias, GELU, GELU', residual, pre-activation
// save) runs on 16-byte fp32 / 8-byte bf16 vectors.
// Weight gradients use split-K over tokens into fp32 slabs + an ordered reduce (bitwise
// reproducible, no float atomics).
#include "rmcl_common.h"
#include "kernels.h"

#define FBK 64

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int kswz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// Issues this wave's global_load_lds instructions for one operand tile of R rows/cols x 64 k.
//   KC: rows r0.. (clamped to Rmax-1), k at k0;   MC: k-rows k0.., columns r0..r0+R-1
template <bool KC, int R, int NW>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, long ld, int r0, int Rmax, int k0, char* lds_tile,
                                           int wave, int lane) {
  constexpr int NINST = R * FBK * 2 / 1024;   // 1 KiB per wave-instruction
  static_assert(NINST % NW == 0, "tile must split evenly over the waves");
#pragma unroll
  for (int i = 0; i < NINST / NW; ++i) {
    const int inst = wave * (NINST / NW) + i;
    const bf16_t* src;
    if (KC) {
      const int row = inst * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      src = base + (long)min(r0 + row, Rmax - 1) * ld + k0 + chunk * 8;
    } else {
      constexpr int LPR = R / 8;              // lanes (16-B chunks) per k-row
      constexpr int KPI = 64 / LPR;           // k-rows per instruction
      const int k = inst * KPI + lane / LPR;
      const int chunk = (lane % LPR) ^ kswz(k);
      src = base + (long)(k0 + k) * ld + r0 + chunk * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
  }
}

// fragment of the 16-row (or 16-col) sub-tile starting at t0, k-step s (32 deep), from a staged tile
template <bool KC, int R>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int t0, int s, int lane) {
  if (KC) {
    const int row = t0 + (lane & 15);
    const int chunk = (4 * s + (lane >> 4)) ^ (row & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + chunk * 16);
  } else {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int c8 = (t0 >> 2) + p;                       // 8-byte chunk index inside the k-row
    union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * s + 8 * (lane >> 4) + 4 * h + q;
      const int c16 = (c8 >> 1) ^ kswz(k);
      const char* a = lds_tile + k * (R * 2) + c16 * 16 + (c8 & 1) * 8;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    return u.v;
  }
}

template <bool A_KC, bool B_KC, typename TO, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64) void gemm_fast_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = WM * 64, BN = WN * 64, NW = WM * WN;
  constexpr int A_BYTES = BM * FBK * 2, STAGE_BYTES = (BM + BN) * FBK * 2;
  constexpr int LOADS = STAGE_BYTES / 1024 / NW;       // global_load_lds per wave per stage
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order (bijective form): blocks that share an XCD walk a contiguous run of tiles,
  // n fastest, so the A row-panel stays in that XCD's L2 while its column tiles are computed.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  int kbeg = 0, kend = g.K;
  long zoff = 0;
  if (g.splitk > 1) {
    const int per = ((g.K / FBK + g.splitk - 1) / g.splitk) * FBK;
    kbeg = blockIdx.y * per;
    kend = min(g.K, kbeg + per);
    zoff = (long)blockIdx.y * g.M * g.ldc;              // slab z of the split-K partial buffer
  }
  const int nk = max(0, (kend - kbeg) / FBK);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int buf) {
    char* st = smem + buf * STAGE_BYTES;
    stage_tile<A_KC, BM, NW>(A, g.lda, m0, g.M, kbeg + kt * FBK, st, wave, lane);
    stage_tile<B_KC, BN, NW>(B, g.ldb, n0, g.N, kbeg + kt * FBK, st + A_BYTES, wave, lane);
  };
  auto compute = [&](int buf) {
    const char* at = smem + buf * STAGE_BYTES;
    const char* bt = at + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<A_KC, BM>(at, wm * 64 + i * 16, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = load_frag<B_KC, BN>(bt, wn * 64 + j * 16, s, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
    }
  };

  if constexpr (STAGES == 2) {
    if (nk > 0) stage(0, 0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
      if (it + 1 < nk) stage(it + 1, (it + 1) & 1);
      compute(it & 1);
      __syncthreads();
    }
  } else {
    // 3-deep ring: tile it+2 is issued right after the barrier that retires tile it-1's reads; the
    // counted wait leaves tile it+1's loads in flight across the barrier.
    if (nk > 0) stage(0, 0);
    if (nk > 1) stage(1, 1);
    int buf = 0;
    for (int it = 0; it < nk; ++it) {
      if (it + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it + 2 < nk) stage(it + 2, buf >= 1 ? buf - 1 : 2);
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }

  // epilogue: lane owns row m = ..+(lane&15), columns n..n+3 with n = ..+4*(lane>>4)
  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C) + zoff;
  TO* C2 = reinterpret_cast<TO*>(g.C2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      float v[4] = {g.alpha * acc[i][j][0], g.alpha * acc[i][j][1], g.alpha * acc[i][j][2], g.alpha * acc[i][j][3]};
      if (epi & EPI_BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (epi & EPI_RESIDUAL) {
        const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + (long)m * g.ld_aux + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
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

// out[i] += sum_z slab[z][i]   (ordered: bitwise reproducible)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long n4, int nz) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4*>(out)[i];
    for (int z = 0; z < nz; ++z) {
      const float4 b = reinterpret_cast<const float4*>(slab)[(long)z * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

int rmcl_slab_reduce(const float* slab, float* out, long n, int nz, hipStream_t s) {
  RMCL_REQUIRE(n % 4 == 0, "slab_reduce: n%4");
  RMCL_LAUNCH(slab_reduce_kernel, dim3(std::min<long>(cdiv(n / 4, 256), 2048)), dim3(256), 0, s, slab, out, n / 4, nz);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---- configuration -----------------------------------------------------------------------------
// cfg 0: 128x128 tile, 4 waves, 2 stages (64 KiB LDS, 2 blocks/CU)
// cfg 1: 128x128 tile, 4 waves, 3 stages (96 KiB LDS, 1 block/CU), counted vmcnt
// cfg 2: 256x128 tile, 8 waves, 3 stages (144 KiB LDS, 1 block/CU), counted vmcnt
// cfg 3: 256x128 tile, 8 waves, 2 stages (96 KiB LDS, 1 block/CU)
static int g_gemm_cfg = -1;   // -1: choose per shape
void rmcl_gemm_fast_set_cfg(int cfg) { g_gemm_cfg = cfg; }

bool rmcl_gemm_fast_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc) {
  if (dt_in != RMCL_BF16) return false;
  if (g.nb1 * g.nb2 != 1) return false;                        // batched GEMMs: exact kernel
  if (!a_kc && b_kc) return false;
  if (g.K < FBK || g.K % FBK != 0 || g.N % 128 != 0) return false;
  if (!a_kc && g.M % 256 != 0) return false;                   // m-contiguous A is read in full rows
  if (g.lda % 8 || g.ldb % 8 || g.ldc % 4 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return false;
  if (g.epi & (EPI_TANH | EPI_ATOMIC)) return false;
  if ((g.epi & EPI_ACCUM) && dt_out != RMCL_F32) return false;
  if ((g.epi & (EPI_RESIDUAL | EPI_DGELU)) && (g.ld_aux % 4 || ((uintptr_t)g.aux & 15))) return false;
  if ((g.epi & EPI_SAVE_PREACT) && ((uintptr_t)g.C2 & 15)) return false;
  if (g.splitk > 1) return false;                              // split-K goes through rmcl_launch_gemm_fast_slab
  return true;
}

template <bool A_KC, bool B_KC, int WM, int WN, int STAGES>
static int launch_cfg(const GemmArgs& g, int dt_out, hipStream_t s) {
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int LDS = STAGES * (BM + BN) * FBK * 2;
  const int tm = cdiv(g.M, BM), tn = g.N / BN;
  dim3 grid(tm * tn, g.splitk > 1 ? g.splitk : 1);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fast_kernel<A_KC, B_KC, float, WM, WN, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fast_kernel<A_KC, B_KC, bf16_t, WM, WN, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  if (dt_out == RMCL_F32) RMCL_LAUNCH((gemm_fast_kernel<A_KC, B_KC, float, WM, WN, STAGES>), grid, dim3(WM * WN * 64), LDS, s, g, tm, tn);
  else RMCL_LAUNCH((gemm_fast_kernel<A_KC, B_KC, bf16_t, WM, WN, STAGES>), grid, dim3(WM * WN * 64), LDS, s, g, tm, tn);
  RMCL_CHECK_LAUNCH();
  return 0;
}

template <bool A_KC, bool B_KC>
static int launch_layout(const GemmArgs& g, int dt_out, int cfg, hipStream_t s) {
  switch (cfg) {
    case 1: return launch_cfg<A_KC, B_KC, 2, 2, 3>(g, dt_out, s);
    case 2: return launch_cfg<A_KC, B_KC, 4, 2, 3>(g, dt_out, s);
    case 3: return launch_cfg<A_KC, B_KC, 4, 2, 2>(g, dt_out, s);
    default: return launch_cfg<A_KC, B_KC, 2, 2, 2>(g, dt_out, s);
  }
}

int rmcl_launch_gemm_fast(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s) {
  int cfg = g_gemm_cfg;
  if (cfg < 0) cfg = 0;
  if (a_kc && b_kc) return launch_layout<true, true>(g, dt_out, cfg, s);
  if (a_kc && !b_kc) return launch_layout<true, false>(g, dt_out, cfg, s);
  return launch_layout<false, false>(g, dt_out, cfg, s);
}

// dW[M=Nout, N=Kin] += A^T B over K tokens with split-K partial slabs (slab: splitk*M*N floats)
int rmcl_launch_gemm_fast_slab(const GemmArgs& g0, float* slab, float* out, hipStream_t s) {
  GemmArgs g = g0;
  g.C = slab;
  g.epi = 0;
  RMCL_TRY(rmcl_launch_gemm_fast(g, RMCL_F32, 0, 0, s));
  return rmcl_slab_reduce(slab, out, (long)g.M * g.N, g.splitk, s);
}
