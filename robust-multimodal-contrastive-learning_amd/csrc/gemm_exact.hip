// Exact-f32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// This is the arithmetic of the fp32 parity path (bit-for-bit a k-ordered fmaf chain per output,
// MI355X guide "FP32-input MFMA") and the any-shape fallback of the bf16 path (bf16 operands are
// widened to f32 when they are staged into LDS).  128x128x16 block tile, 4 waves (2x2), each wave
// 64x64 = 2x2 MFMA tiles of 32x32; LDS images are k-major [16][128+4] f32 so that the one-float
// A/B fragments (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) are conflict-free ds_read_b32.
// Register-staged double buffering: tile t+1 is loaded to VGPRs before the MFMAs of tile t.
#include "rmcl_common.h"
#include "gemm.h"

#define GBM 128
#define GBN 128
#define GBK 16
#define GLD (GBM + 4)

template <typename TI> struct VecTraits;
template <> struct VecTraits<float> { static constexpr int V = 4; };
template <> struct VecTraits<bf16_t> { static constexpr int V = 8; };

// Loads this thread's share (8 elements) of a [128 rows x 16 k] operand tile into f32 registers.
// KC: element (r,k) at base[r*ld + k];  MC: element (r,k) at base[k*ld + r].
template <typename TI, bool KC>
__device__ __forceinline__ void tile_load(float (&reg)[8], const TI* __restrict__ base, long ld, int r0, int R,
                                          int k0, int K, int t) {
  constexpr int V = VecTraits<TI>::V;
  constexpr int NV = 8 / V;  // vectors per thread
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = t + 256 * i;
    int r, k;
    if (KC) {
      r = v / (GBK / V);
      k = (v % (GBK / V)) * V;
    } else {
      k = v / (GBM / V);
      r = (v % (GBM / V)) * V;
    }
    const bool ok = (r0 + r < R) && (k0 + k < K);
    const TI* p = KC ? base + (long)(r0 + r) * ld + (k0 + k) : base + (long)(k0 + k) * ld + (r0 + r);
    if (ok) {
      if constexpr (V == 4) {
        const float4 x = *reinterpret_cast<const float4*>(p);
        reg[4 * i + 0] = x.x; reg[4 * i + 1] = x.y; reg[4 * i + 2] = x.z; reg[4 * i + 3] = x.w;
      } else {
        const uint4 x = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          reg[2 * j + 0] = __uint_as_float(w[j] << 16);
          reg[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      }
      // zero the tail of a vector that straddles the end of the contiguous dimension
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool in = KC ? (k0 + k + j < K) : (r0 + r + j < R);
        if (!in) reg[V * i + j] = 0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) reg[V * i + j] = 0.0f;
    }
  }
}

template <typename TI, bool KC>
__device__ __forceinline__ void tile_store(const float (&reg)[8], float* __restrict__ S, int t) {
  constexpr int V = VecTraits<TI>::V;
  constexpr int NV = 8 / V;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = t + 256 * i;
    if (KC) {
      const int r = v / (GBK / V), k = (v % (GBK / V)) * V;
#pragma unroll
      for (int j = 0; j < V; ++j) S[(k + j) * GLD + r] = reg[V * i + j];
    } else {
      const int k = v / (GBM / V), r = (v % (GBM / V)) * V;
#pragma unroll
      for (int j = 0; j < V; j += 4)
        *reinterpret_cast<float4*>(&S[k * GLD + r + j]) =
            make_float4(reg[V * i + j], reg[V * i + j + 1], reg[V * i + j + 2], reg[V * i + j + 3]);
    }
  }
}

template <typename TI, typename TO, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_exact_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * GBK * GLD];
  float* As = smem;
  float* Bs = smem + 2 * GBK * GLD;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * GBN, m0 = blockIdx.y * GBM;

  const TI* A = reinterpret_cast<const TI*>(g.A);
  const TI* B = reinterpret_cast<const TI*>(g.B);
  TO* C = reinterpret_cast<TO*>(g.C);
  TO* C2 = reinterpret_cast<TO*>(g.C2);
  const char* aux = reinterpret_cast<const char*>(g.aux);
  int kbeg = 0, kend = g.K;
  if (g.splitk > 1) {
    const int per = ((g.K + g.splitk - 1) / g.splitk + GBK - 1) / GBK * GBK;
    kbeg = blockIdx.z * per;
    kend = min(g.K, kbeg + per);
  } else {
    const int b1 = blockIdx.z / g.nb2, b2 = blockIdx.z % g.nb2;
    A += b1 * g.sA1 + b2 * g.sA2;
    B += b1 * g.sB1 + b2 * g.sB2;
    const long co = b1 * g.sC1 + b2 * g.sC2;
    C += co;
    if (C2) C2 += co;
    if (aux) aux += co * ((g.epi & EPI_RESIDUAL) ? sizeof(float) : sizeof(TI));
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float ra[8], rb[8];
  const int nk = (kend - kbeg + GBK - 1) / GBK;
  if (nk > 0) {
    tile_load<TI, A_KC>(ra, A, g.lda, m0, g.M, kbeg, kend, t);
    tile_load<TI, B_KC>(rb, B, g.ldb, n0, g.N, kbeg, kend, t);
    tile_store<TI, A_KC>(ra, As, t);
    tile_store<TI, B_KC>(rb, Bs, t);
  }
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    if (it + 1 < nk) {
      tile_load<TI, A_KC>(ra, A, g.lda, m0, g.M, kbeg + (it + 1) * GBK, kend, t);
      tile_load<TI, B_KC>(rb, B, g.ldb, n0, g.N, kbeg + (it + 1) * GBK, kend, t);
    }
    const float* as = As + cur * GBK * GLD + wm * 64 + (lane & 31);
    const float* bs = Bs + cur * GBK * GLD + wn * 64 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int kr = (kk + (lane >> 5)) * GLD;
      const float a0 = as[kr], a1 = as[kr + 32];
      const float b0 = bs[kr], b1 = bs[kr + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (it + 1 < nk) {
      tile_store<TI, A_KC>(ra, As + (cur ^ 1) * GBK * GLD, t);
      tile_store<TI, B_KC>(rb, Bs + (cur ^ 1) * GBK * GLD, t);
    }
    __syncthreads();
  }

  // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int epi = g.epi;
  const bool first_slice = (g.splitk <= 1) || (blockIdx.z == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (row < g.M && col < g.N) {
          float v = g.alpha * acc[i][j][r];
          if ((epi & EPI_BIAS) && first_slice) v += g.bias[col];
          const long drow = (long)row * (g.drop_row_mul > 0 ? g.drop_row_mul : 1);
          if (epi & EPI_DROP_BWD) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ld_aux + col), g.drop_thresh, g.drop_inv_keep);
          if (epi & EPI_DGELU) v *= gelu_erf_grad(to_f32<TI>(reinterpret_cast<const TI*>(aux)[(long)row * g.ld_aux + col]));
          const long ci = (long)row * g.ldc + col;
          if (epi & EPI_SAVE_PREACT) C2[ci] = from_f32<TO>(v);
          if (epi & EPI_GELU) v = gelu_erf(v);
          if (epi & EPI_TANH) v = tanhf(v);
          if (epi & EPI_DROPOUT) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ldc + col), g.drop_thresh, g.drop_inv_keep);
          if (epi & EPI_RESIDUAL) v += reinterpret_cast<const float*>(aux)[(long)row * g.ld_aux + col];
          if (epi & EPI_ATOMIC) {
            if constexpr (sizeof(TO) == 4) atomicAdd(reinterpret_cast<float*>(C) + ci, v);
          } else if (epi & EPI_ACCUM) {
            C[ci] = from_f32<TO>(to_f32<TO>(C[ci]) + v);
          } else {
            C[ci] = from_f32<TO>(v);
          }
        }
      }
}

template <typename TI, typename TO>
static int launch_layout(const GemmArgs& g, int a_kc, int b_kc, dim3 grid, hipStream_t s) {
  if (a_kc && b_kc) RMCL_LAUNCH((gemm_exact_kernel<TI, TO, true, true>), grid, dim3(256), 0, s, g);
  else if (a_kc && !b_kc) RMCL_LAUNCH((gemm_exact_kernel<TI, TO, true, false>), grid, dim3(256), 0, s, g);
  else if (!a_kc && !b_kc) RMCL_LAUNCH((gemm_exact_kernel<TI, TO, false, false>), grid, dim3(256), 0, s, g);
  else RMCL_LAUNCH((gemm_exact_kernel<TI, TO, false, true>), grid, dim3(256), 0, s, g);
  RMCL_CHECK_LAUNCH();
  return 0;
}

bool rmcl_gemm_skinny_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc);
int rmcl_launch_gemm_skinny(const GemmArgs& g, int b_kc, hipStream_t s);
static int g_skinny_form = -1;      // rmcl_tune_set key 6: 0 = the first forms only (row-split skinny kernel, LDS-staged TN kernel)

// Outer-product-like fp32 GEMM: C[M, N] (+)= alpha * A^T B with A [K, M], B [K, N] and a SHORT reduction (K = the batch rows:
// the weight gradients of the heads, of the encoder's cls-only tail and of the Barlow-Twins head, and its cross-correlation
// matrix).  The 128x128x16 LDS-staged kernel above spends its time in four barrier-separated k-tiles for 64 k; here every
// wave owns a 64 x 64 output tile and streams both operands straight from L2 (lane (c, g) of a 16x16x4 MFMA reads A[k0+g][m0+c]
// and B[k0+g][n0+c]: 64-byte rows), 16 MFMAs per 8 loads, no LDS, no barrier.
__global__ __launch_bounds__(256) void gemm_tn_shortk_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, gq = lane >> 4;
  const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128 + (wave & 1) * 64;
  if (m0 >= g.M || n0 >= g.N) return;
  const float* A = reinterpret_cast<const float*>(g.A);
  const float* B = reinterpret_cast<const float*>(g.B);
  int am[4], bn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    am[t] = min(m0 + t * 16 + c, g.M - 1);
    bn[t] = min(n0 + t * 16 + c, g.N - 1);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // EPI_COLSUM: column sums of A over k (= the bias gradient that goes with this weight gradient) as a by-product of the operand
  // loads, written by the waves of the first column block (n0 == 0: one writer per column m) - no separate column-sum launch
  const bool want_cs = (g.epi & EPI_COLSUM) && n0 == 0;
  float sa[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < g.K; k0 += 8) {                    // two 4-k steps per trip, all 16 loads in flight before the MFMAs
    float a[2][4], b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = min(k0 + 4 * u + gq, g.K - 1);
      const bool live = k0 + 4 * u + gq < g.K;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[u][t] = live ? A[(long)k * g.lda + am[t]] : 0.f;
        b[u][t] = live ? B[(long)k * g.ldb + bn[t]] : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) sa[t] += a[0][t] + a[1][t];     // (rows past K were loaded as 0)
  }
  if (want_cs) {                                             // lane (c, gq) holds the rows k = gq (mod 4): sum over the four lane groups
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = sa[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m0 + t * 16 + c;
      if (gq == 0 && m < g.M) g.colsum[m] += v;
    }
  }
  float* C = reinterpret_cast<float*>(g.C);
  // C += : every old value of the tile is loaded BEFORE the first store.  "load, add, store" per element cannot be reordered by the
  // compiler (the stores may alias the next loads for all it knows) and ran as 64 serial round trips per lane (tools/st_trace.py
  // found the same pattern in the weight-gradient launch's epilogue)
  float oldc[4][4][4];
  const bool accum = (g.epi & EPI_ACCUM) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + i * 16 + 4 * gq + r, col = n0 + j * 16 + c;
        oldc[i][j][r] = (accum && row < g.M && col < g.N) ? C[(long)row * g.ldc + col] : 0.f;
      }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + i * 16 + 4 * gq + r, col = n0 + j * 16 + c;
        if (row < g.M && col < g.N) C[(long)row * g.ldc + col] = g.alpha * acc[i][j][r] + oldc[i][j][r];
      }
}

// true when rmcl_launch_gemm_exact(g, f32, f32, a_kc = 0, b_kc = 0) runs the short-K kernel, the only one that knows EPI_COLSUM
bool rmcl_gemm_tn_shortk_takes(const GemmArgs& g) {
  return g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K >= 1 && g.K <= 256 && g.splitk <= 1 && g.nb1 * g.nb2 == 1 &&
         (g.epi & ~(EPI_ACCUM | EPI_COLSUM)) == 0 && g_skinny_form != 0;
}

int rmcl_launch_gemm_exact(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc, hipStream_t s) {
  if (g.A && g.B && g.C && g.M > 0 && g.N > 0 && rmcl_gemm_skinny_supported(g, dt_in, dt_out, a_kc)) return rmcl_launch_gemm_skinny(g, b_kc, s);
  if (g.A && g.B && g.C && g.M > 0 && g.N > 0 && !a_kc && !b_kc && dt_in == RMCL_F32 && dt_out == RMCL_F32 && g.K >= 1 && g.K <= 256 &&
      g.splitk <= 1 && g.nb1 * g.nb2 == 1 && (g.epi & ~(EPI_ACCUM | EPI_COLSUM)) == 0 && (!(g.epi & EPI_COLSUM) || g.colsum) && g_skinny_form != 0) {
    RMCL_LAUNCH(gemm_tn_shortk_kernel, dim3(cdiv(g.N, 128), cdiv(g.M, 128)), dim3(256), 0, s, g);
    RMCL_CHECK_LAUNCH();
    return 0;
  }
  const int V = dt_in == RMCL_F32 ? 4 : 8;
  RMCL_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0, "gemm: bad dims");
  RMCL_REQUIRE(g.lda % V == 0 && g.ldb % V == 0, "gemm: lda/ldb must be a multiple of the 16-byte vector width");
  // a vector may straddle the end of the contiguous dimension only if the row pitch covers it
  const int Kr = (g.K + V - 1) / V * V, Mr = (g.M + V - 1) / V * V, Nr = (g.N + V - 1) / V * V;
  RMCL_REQUIRE(a_kc ? (g.lda >= Kr) : (g.lda >= Mr), "gemm: lda too small for vector loads");
  RMCL_REQUIRE(b_kc ? (g.ldb >= Kr) : (g.ldb >= Nr), "gemm: ldb too small for vector loads");
  RMCL_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm: A/B must be 16-byte aligned");
  RMCL_REQUIRE(!(g.epi & EPI_ATOMIC) || dt_out == RMCL_F32, "gemm: atomic epilogue needs fp32 C");
  RMCL_REQUIRE(g.splitk <= 1 || ((g.epi & EPI_ATOMIC) && g.nb1 * g.nb2 == 1), "gemm: split-K needs EPI_ATOMIC and no batching");
  const int z = g.splitk > 1 ? g.splitk : g.nb1 * g.nb2;
  RMCL_REQUIRE(z >= 1 && z <= 65535, "gemm: bad batch count");
  dim3 grid(cdiv(g.N, GBN), cdiv(g.M, GBM), z);
  if (dt_in == RMCL_F32 && dt_out == RMCL_F32) return launch_layout<float, float>(g, a_kc, b_kc, grid, s);
  if (dt_in == RMCL_BF16 && dt_out == RMCL_BF16) return launch_layout<bf16_t, bf16_t>(g, a_kc, b_kc, grid, s);
  if (dt_in == RMCL_BF16 && dt_out == RMCL_F32) return launch_layout<bf16_t, float>(g, a_kc, b_kc, grid, s);
  if (dt_in == RMCL_F32 && dt_out == RMCL_BF16) return launch_layout<float, bf16_t>(g, a_kc, b_kc, grid, s);
  rmcl_set_error("gemm: unsupported dtype combination");
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// Skinny exact-f32 GEMM for the heads (pooler / MoCo head / ITM): M <= 64 rows per block, K long.
// The 128x128 kernel above would run such a problem on N/128 = 6 workgroups, latency-bound at
// ~120 us; here every workgroup owns 64 rows x 16 columns (N/16 workgroups), each wave one 16x16
// MFMA tile (v_mfma_f32_16x16x4_f32), operands streamed straight from L2 with 16-byte loads, no LDS.
// k order inside a 16-wide chunk is permuted (lane group g, element j <-> k = 16c + 4g + j) on both
// operands, so one float4 per lane feeds four MFMAs.
// ---------------------------------------------------------------------------------------------------
template <bool B_KC>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gq = lane >> 4;
  const int m = blockIdx.y * 64 + wave * 16 + (lane & 15), n = blockIdx.x * 16 + (lane & 15);
  const float* A = reinterpret_cast<const float*>(g.A) + (long)min(m, g.M - 1) * g.lda + 4 * gq;
  const float* B = reinterpret_cast<const float*>(g.B);
  const int nc = min(n, g.N - 1);
  B += B_KC ? (long)nc * g.ldb + 4 * gq : (long)(4 * gq) * g.ldb + nc;
  // four independent accumulator chains (the 16x16x4 MFMA is 8 passes deep: one chain would serialise on its own latency)
  // and 8 k-steps of loads in flight; the chains are summed in k order at the end.  NOTE: this changes the summation
  // order relative to a single chain only in the last bits (heads: fp32, tolerance 1e-5).
  f32x4 ac[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // 64 k per iteration with every load of the iteration issued before the first MFMA (the compiler otherwise waits
  // vmcnt(0) after each scalar B load: four serial L2 round trips per 16 k); 16-k tail for K % 64
  int k0 = 0;
  for (; k0 + 64 <= g.K; k0 += 64) {
    float4 av[4];
    float bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = *reinterpret_cast<const float4*>(A + k0 + 16 * u);
      if (B_KC) {
        const float4 t = *reinterpret_cast<const float4*>(B + k0 + 16 * u);
        bv[u][0] = t.x; bv[u][1] = t.y; bv[u][2] = t.z; bv[u][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[u][j] = B[(long)(k0 + 16 * u + j) * g.ldb];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                       // all 20 loads are in flight before the first MFMA waits
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ac[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u][0], ac[0], 0, 0, 0);
      ac[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u][1], ac[1], 0, 0, 0);
      ac[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u][2], ac[2], 0, 0, 0);
      ac[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u][3], ac[3], 0, 0, 0);
    }
  }
  for (; k0 < g.K; k0 += 16) {
    const float4 a = *reinterpret_cast<const float4*>(A + k0);
    float b[4];
    if (B_KC) {
      const float4 t = *reinterpret_cast<const float4*>(B + k0);
      b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = B[(long)(k0 + j) * g.ldb];
    }
    ac[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], ac[0], 0, 0, 0);
    ac[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], ac[1], 0, 0, 0);
    ac[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], ac[2], 0, 0, 0);
    ac[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], ac[3], 0, 0, 0);
  }
  f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r]);
  float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = blockIdx.y * 64 + wave * 16 + 4 * gq + r;
    if (row < g.M && n < g.N) {
      float v = g.alpha * acc[r];
      if (g.epi & EPI_BIAS) v += g.bias[n];
      const long ci = (long)row * g.ldc + n;
      if (g.epi & EPI_SAVE_PREACT) reinterpret_cast<float*>(g.C2)[ci] = v;
      if (g.epi & EPI_GELU) v = gelu_erf(v);
      if (g.epi & EPI_TANH) v = tanhf(v);
      if (g.epi & EPI_DGELU) v *= gelu_erf_grad(reinterpret_cast<const float*>(g.aux)[(long)row * g.ld_aux + n]);
      const long drow = (long)row * (g.drop_row_mul > 0 ? g.drop_row_mul : 1);                       // dense row of this compact row (gemm.h)
      if (g.epi & EPI_DROP_BWD) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ld_aux + n), g.drop_thresh, g.drop_inv_keep);
      if (g.epi & EPI_DROPOUT) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ldc + n), g.drop_thresh, g.drop_inv_keep);
      if (g.epi & EPI_RESIDUAL) v += reinterpret_cast<const float*>(g.aux)[(long)row * g.ld_aux + n];
      if (g.epi & EPI_ACCUM) v += C[ci];
      C[ci] = v;
      if (g.epi & EPI_DUP) reinterpret_cast<float*>(g.C2)[ci] = v;
    }
  }
}

// Second form of the skinny GEMM: the four waves of a workgroup split K instead of the rows.  Every wave covers all 64 rows
// (4 row tiles) x NT columns over its quarter of K, so a B fragment feeds 4 (x NT/16) MFMAs instead of 1, the serial k-loop
// of a workgroup is 4x shorter (N = 768, K = 3072: 48 workgroups were looping 3072 deep), and the partial tiles meet in LDS
// where the epilogue runs.  NT = 32 halves the re-reads of the A panel for very wide outputs (Barlow-Twins head, N = 8192).
// Same operand trick as above: lane (c, g) holds k = 16 s + 4 g + j in component j of one float4, on both operands.
template <bool B_KC, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_ksplit_kernel(GemmArgs g) {
  constexpr int CT = NT / 16;
  __shared__ float red[NW][64][NT + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, gq = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * NT;
  const int kq = g.K / NW, k_lo = wave * kq;                               // kq % 16 == 0 (launcher)
  const float* Ap[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) Ap[rt] = reinterpret_cast<const float*>(g.A) + (long)min(m0 + rt * 16 + c, g.M - 1) * g.lda + 4 * gq;
  const float* Bp[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int nc = min(n0 + ct * 16 + c, g.N - 1);
    Bp[ct] = reinterpret_cast<const float*>(g.B) + (B_KC ? (long)nc * g.ldb + 4 * gq : (long)(4 * gq) * g.ldb + nc);
  }
  f32x4 acc[4][CT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  // register ring over 16-k steps, PD - 1 steps of loads in flight ahead of the MFMAs: the waves are few and every step is an L2 /
  // HBM round trip (64-byte row pieces of A), so with one step of lookahead the k-loop was a chain of 12 - 24 such round trips
  // (K = 3072, 8 waves: 41.8 us for 9 MB of weights).  The ring index is a compile-time constant (PD steps per trip).
  constexpr int PD = 4;                                       // (round 4: 8 steps of lookahead measured SLOWER in the step: 33.9 vs 33.6 ms, three alternating runs)
  float4 av[PD][4];
  float bv[PD][CT][4];
  auto load = [&](int buf, int k) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) av[buf][rt] = *reinterpret_cast<const float4*>(Ap[rt] + k);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (B_KC) {
        const float4 t = *reinterpret_cast<const float4*>(Bp[ct] + k);
        bv[buf][ct][0] = t.x; bv[buf][ct][1] = t.y; bv[buf][ct][2] = t.z; bv[buf][ct][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[buf][ct][j] = Bp[ct][(long)(k + j) * g.ldb];
      }
    }
  };
  auto mac = [&](int buf) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][rt].x, bv[buf][ct][0], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][rt].y, bv[buf][ct][1], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][rt].z, bv[buf][ct][2], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][rt].w, bv[buf][ct][3], acc[rt][ct], 0, 0, 0);
      }
  };
  const int nsteps = kq / 16;
#pragma unroll
  for (int i = 0; i < PD - 1; ++i)
    if (i < nsteps) load(i, k_lo + 16 * i);
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const int step = s0 + j;
      if (step < nsteps) {
        if (step + PD - 1 < nsteps) load((j + PD - 1) % PD, k_lo + 16 * (step + PD - 1));
        __builtin_amdgcn_sched_barrier(0);
        mac(j);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][rt * 16 + 4 * gq + r][ct * 16 + c] = acc[rt][ct][r];
  __syncthreads();
  float* C = reinterpret_cast<float*>(g.C);
  // every global operand of the epilogue (bias, aux, old C) is loaded for ALL of the thread's elements before the first store: in
  // "load, compute, store" order per element the loads of element i + 1 cannot pass the store of element i (possible alias)
  constexpr int ITER = NT / NW;                                // 64 * NT elements over 64 * NW threads
  float bia[ITER], aux[ITER], oldc[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = threadIdx.x + it * 64 * NW, rl = i / NT, cl = i % NT, row = m0 + rl, n = n0 + cl;
    const bool live = row < g.M && n < g.N;
    bia[it] = (live && (g.epi & EPI_BIAS)) ? g.bias[n] : 0.f;
    aux[it] = (live && (g.epi & (EPI_DGELU | EPI_RESIDUAL))) ? reinterpret_cast<const float*>(g.aux)[(long)row * g.ld_aux + n] : 0.f;
    oldc[it] = (live && (g.epi & EPI_ACCUM)) ? C[(long)row * g.ldc + n] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = threadIdx.x + it * 64 * NW, rl = i / NT, cl = i % NT, row = m0 + rl, n = n0 + cl;
    if (row >= g.M || n >= g.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][rl][cl];                                              // k order: wave 0 .. NW-1
    v *= g.alpha;
    if (g.epi & EPI_BIAS) v += bia[it];
    const long ci = (long)row * g.ldc + n;
    if (g.epi & EPI_SAVE_PREACT) reinterpret_cast<float*>(g.C2)[ci] = v;
    if (g.epi & EPI_GELU) v = gelu_erf(v);
    if (g.epi & EPI_TANH) v = tanhf(v);
    if (g.epi & EPI_DGELU) v *= gelu_erf_grad(aux[it]);
    const long drow = (long)row * (g.drop_row_mul > 0 ? g.drop_row_mul : 1);                         // dense row of this compact row (gemm.h)
    if (g.epi & EPI_DROP_BWD) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ld_aux + n), g.drop_thresh, g.drop_inv_keep);
    if (g.epi & EPI_DROPOUT) v *= drop_scale(g.drop_seed, (uint32_t)(drow * g.ldc + n), g.drop_thresh, g.drop_inv_keep);
    if (g.epi & EPI_RESIDUAL) v += aux[it];
    if (g.epi & EPI_ACCUM) v += oldc[it];
    C[ci] = v;
    if (g.epi & EPI_DUP) reinterpret_cast<float*>(g.C2)[ci] = v;
  }
}

bool rmcl_gemm_skinny_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc) {
  return a_kc && dt_in == RMCL_F32 && dt_out == RMCL_F32 && g.M <= 1024 && g.K % 16 == 0 && g.K >= 64 && g.splitk <= 1 &&
         g.nb1 * g.nb2 == 1 && (g.epi & ~(EPI_BIAS | EPI_TANH | EPI_ACCUM | EPI_GELU | EPI_SAVE_PREACT | EPI_DGELU | EPI_RESIDUAL | EPI_DUP | EPI_DROPOUT | EPI_DROP_BWD)) == 0 &&
         (!(g.epi & EPI_DROP_BWD) || (g.epi & EPI_DGELU)) &&
         !((g.epi & EPI_DUP) && (g.epi & EPI_SAVE_PREACT)) &&
         !((g.epi & EPI_DGELU) && (g.epi & EPI_RESIDUAL)) && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
         ((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0;
}

void rmcl_gemm_skinny_set_form(int v) { g_skinny_form = v; }

int rmcl_launch_gemm_skinny(const GemmArgs& g, int b_kc, hipStream_t s) {
  if (g_skinny_form != 0 && g.K % 64 == 0 && g.K >= 128) {
    // 8 waves (K split 8 ways) where the reduction is long and the grid small; 32-column tiles for very wide outputs
    // (round 4: 8 waves from K = 512 on measured -0.05 ms per step over six alternating runs, but the different fp32 summation order moved the
    // 4-sample BatchNorm golden of the Barlow-Twins step from 0.9 % to 1.03 % on one gradient norm - bound 1 % - so the threshold stays)
    const bool w8 = g.K % 128 == 0 && g.K >= 2048 && g.N < 4096;
    if (g.N >= 4096) {
      dim3 grid(cdiv(g.N, 32), cdiv(g.M, 64));
      if (b_kc) RMCL_LAUNCH((gemm_skinny_ksplit_kernel<true, 32, 4>), grid, dim3(256), 0, s, g);
      else RMCL_LAUNCH((gemm_skinny_ksplit_kernel<false, 32, 4>), grid, dim3(256), 0, s, g);
    } else if (w8) {
      dim3 grid(cdiv(g.N, 16), cdiv(g.M, 64));
      if (b_kc) RMCL_LAUNCH((gemm_skinny_ksplit_kernel<true, 16, 8>), grid, dim3(512), 0, s, g);
      else RMCL_LAUNCH((gemm_skinny_ksplit_kernel<false, 16, 8>), grid, dim3(512), 0, s, g);
    } else {
      dim3 grid(cdiv(g.N, 16), cdiv(g.M, 64));
      if (b_kc) RMCL_LAUNCH((gemm_skinny_ksplit_kernel<true, 16, 4>), grid, dim3(256), 0, s, g);
      else RMCL_LAUNCH((gemm_skinny_ksplit_kernel<false, 16, 4>), grid, dim3(256), 0, s, g);
    }
    RMCL_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid(cdiv(g.N, 16), cdiv(g.M, 64));
  if (b_kc) RMCL_LAUNCH(gemm_skinny_kernel<true>, grid, dim3(256), 0, s, g);
  else RMCL_LAUNCH(gemm_skinny_kernel<false>, grid, dim3(256), 0, s, g);
  RMCL_CHECK_LAUNCH();
  return 0;
}
