// bf16 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) - the workhorse of the
// encoder forward (NT), data-gradient (NN) and weight-gradient (TN) passes.
//
// PERSISTENT workgroups (2 per CU, 4 waves each) walk the 128x128 output tiles; one wave per 64x64
// sub-tile (4x4 MFMA tiles, 64 accumulator registers), 64-deep k-tiles.
// Staging: global -> LDS directly with 16-byte global_load_lds (no VGPR round trip) into two LDS
// stages; the k-tiles of ALL tiles a workgroup owns form one stream, so the first k-tile of the
// next output tile is already in flight while the current tile's last MFMAs and its epilogue run
// (with K = 768 a tile has only 12 k-steps: an exposed prologue per tile costs ~30 %).
// The LDS image is lane-linear as the instruction requires; bank conflicts are removed by permuting
// the per-lane SOURCE address and applying the same involution on the fragment reads (rule 21 / T2):
//   * k-contiguous operands ([rows][64 k], 128-B rows): 16-B chunk index ^= (row & 7);
//     fragments are one ds_read_b128 per lane (A[row=l&15][k=8(l>>4)+j]).
//   * m/n-contiguous operands ([64 k][128 cols]; the W of dX = dY W and both operands of
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

#define FBM 128
#define FBN 128
#define FBK 64
#define GROUP_M 8
#define A_BYTES (FBM * FBK * 2)
#define STAGE_BYTES ((FBM + FBN) * FBK * 2)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int kswz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// Issues this wave's 4 global_load_lds instructions for one 128 x 64 operand tile (16 KiB).
//   KC: rows r0.. (clamped to Rmax-1), k at k0;   MC: k-rows k0.., columns r0..r0+127
template <bool KC>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, long ld, int r0, int Rmax, int k0, char* lds_tile,
                                           int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const bf16_t* src;
    if (KC) {
      const int row = inst * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      src = base + (long)min(r0 + row, Rmax - 1) * ld + k0 + chunk * 8;
    } else {
      const int k = inst * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ kswz(k);
      src = base + (long)(k0 + k) * ld + r0 + chunk * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
  }
}

// fragment of the 16-row (or 16-col) sub-tile starting at t0, k-step s (32 deep), from a staged tile
template <bool KC>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int t0, int s, int lane) {
  if (KC) {
    const int row = t0 + (lane & 15);
    const int chunk = (4 * s + (lane >> 4)) ^ (row & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + chunk * 16);
  } else {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int c8 = (t0 >> 2) + p;                       // 8-byte chunk index inside the 256-B k-row
    union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * s + 8 * (lane >> 4) + 4 * h + q;
      const int c16 = (c8 >> 1) ^ kswz(k);
      const char* a = lds_tile + k * 256 + c16 * 16 + (c8 & 1) * 8;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    return u.v;
  }
}

struct TileCoord {
  int m0, n0, kbeg, nk;
  long zoff;
};

// virtual tile id -> coordinates.  vt = z * nwg + t; within a slice the XCD-aware bijective remap
// gives every XCD (blocks with equal id & 7; the persistent grid is a multiple of 8) a contiguous run
// of tiles, n fastest, so an A row-panel stays in that XCD's L2 while its column tiles are computed.
__device__ __forceinline__ TileCoord decode_tile(const GemmArgs& g, int vt, int tiles_n, int nwg) {
  TileCoord c;
  const int z = vt / nwg;
  int t = vt - z * nwg;
  {
    const int xcd = t & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  {
    // grouped (super-tile) order inside the XCD's run: 8 row-panels x all column tiles per group, rows
    // fastest, so the ~64 tiles an XCD works on at a time touch 8 A-panels + 8 B-panels (3 MB at K=768,
    // fits the 4 MiB L2) instead of 3 A-panels + every B-panel (measured: 28 % of the LDS-DMA bytes of
    // the fc1 GEMM were missing L2 with the plain n-fastest order).
    const int tiles_m = nwg / tiles_n, per_group = GROUP_M * tiles_n;
    const int group = t / per_group, first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M), r = t - group * per_group;
    c.m0 = (first_m + r % gsz) * FBM;
    c.n0 = (r / gsz) * FBN;
  }
  c.kbeg = 0;
  int kend = g.K;
  c.zoff = 0;
  if (g.splitk > 1) {
    const int per = ((g.K / FBK + g.splitk - 1) / g.splitk) * FBK;
    c.kbeg = z * per;
    kend = min(g.K, c.kbeg + per);
    c.zoff = (long)z * g.M * g.ldc;                     // slab z of the split-K partial buffer
  }
  c.nk = (kend - c.kbeg) / FBK;                         // >= 1 (launcher guarantees non-empty slices)
  return c;
}

// DROP: dropout epilogues compiled in (a separate instantiation: their extra live state costs >100 spilled
// VGPRs in the common no-dropout kernels otherwise).
template <bool A_KC, bool B_KC, typename TO, bool DROP>
__global__ __launch_bounds__(256, 2) void gemm_fast_kernel(GemmArgs g, int tiles_n, int nwg, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  const int G = gridDim.x;
  const bool dbg_noload = g.tag & (1 << 30), dbg_nomma = g.tag & (1 << 29);   // ablation switches (tools/gemm_bench.py)

  auto stage = [&](const TileCoord& c, int kt, int buf) {
    char* st = smem + buf * STAGE_BYTES;
    stage_tile<A_KC>(A, g.lda, c.m0, g.M, c.kbeg + kt * FBK, st, wave, lane);
    stage_tile<B_KC>(B, g.ldb, c.n0, g.N, c.kbeg + kt * FBK, st + A_BYTES, wave, lane);
  };

  int vt = blockIdx.x;
  if (vt >= total) return;
  TileCoord cur = decode_tile(g, vt, tiles_n, nwg);
  stage(cur, 0, 0);
  __syncthreads();
  int buf = 0;
  for (; vt < total; vt += G) {
    const bool has_next = vt + G < total;
    TileCoord nxt = cur;
    if (has_next) nxt = decode_tile(g, vt + G, tiles_n, nwg);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int it = 0; it < cur.nk; ++it) {
      const bool last = it + 1 == cur.nk;
      if (!dbg_noload) {
        if (!last) stage(cur, it + 1, buf ^ 1);
        else if (has_next) stage(nxt, 0, buf ^ 1);          // next tile's first k-tile rides under this tile's tail
      }
      if (!dbg_nomma) {
        const char* at = smem + buf * STAGE_BYTES;
        const char* bt = at + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 af[4], bf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = load_frag<A_KC>(at, wm * 64 + i * 16, s, lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[j] = load_frag<B_KC>(bt, wn * 64 + j * 16, s, lane);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
        }
      }
      if (last) {
        // epilogue (registers -> global only, so it overlaps the in-flight prefetch):
        // lane owns row m = ..+(lane&15), columns n..n+3 with n = ..+4*(lane>>4)
        const int epi = g.epi;
        TO* C = reinterpret_cast<TO*>(g.C) + cur.zoff;
        TO* C2 = reinterpret_cast<TO*>(g.C2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = cur.m0 + wm * 64 + i * 16 + (lane & 15);
          if (m >= g.M) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = cur.n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
            float v[4] = {g.alpha * acc[i][j][0], g.alpha * acc[i][j][1], g.alpha * acc[i][j][2], g.alpha * acc[i][j][3]};
            if (epi & EPI_BIAS) {
              const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
              v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (DROP && (epi & EPI_DROP_BWD)) {
              const uint32_t di = (uint32_t)((long)m * g.ld_aux + n);
              drop_scale4(g.drop_seed, di, g.drop_thresh, g.drop_inv_keep, v[0], v[1], v[2], v[3]);
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
            if (DROP && (epi & EPI_DROPOUT)) {
              drop_scale4(g.drop_seed, (uint32_t)ci, g.drop_thresh, g.drop_inv_keep, v[0], v[1], v[2], v[3]);
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
            // keep the 16 per-tile epilogues from being interleaved: hoisting every tile's bias / residual /
            // aux loads to the top costs > 160 live VGPRs and spills (121 spilled registers measured)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
      buf ^= 1;
    }
    cur = nxt;
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

// cfg (rmcl_tune_set key 0): persistent grid size in workgroups per CU (default 2); 10 / 11 = ablations
static int g_gemm_cfg = -1;
int rmcl_gemm_fast_get_cfg() { return g_gemm_cfg; }
extern int g_st_xflags;
void rmcl_gemm_fast_set_cfg(int cfg) {
  if (cfg > 60 && cfg < 70) { g_st_xflags = cfg - 60; cfg = 60; } else { g_st_xflags = 0; }
  g_gemm_cfg = cfg;
}

bool rmcl_gemm_fast_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc) {
  if (dt_in != RMCL_BF16) return false;
  if (g.nb1 * g.nb2 != 1) return false;                        // batched GEMMs: exact kernel
  if (!a_kc && b_kc) return false;
  if (g.K < FBK || g.K % FBK != 0 || g.N % FBN != 0) return false;
  if (!a_kc && g.M % FBM != 0) return false;                   // m-contiguous A is read in full 256-B rows
  if (g.lda % 8 || g.ldb % 8 || g.ldc % 4 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return false;
  if (g.epi & (EPI_TANH | EPI_ATOMIC)) return false;
  if ((g.epi & EPI_ACCUM) && dt_out != RMCL_F32) return false;
  if ((g.epi & (EPI_RESIDUAL | EPI_DGELU)) && (g.ld_aux % 4 || ((uintptr_t)g.aux & 15))) return false;
  if ((g.epi & EPI_SAVE_PREACT) && ((uintptr_t)g.C2 & 15)) return false;
  if (g.splitk > 1) return false;                              // split-K goes through rmcl_launch_gemm_fast_slab
  return true;
}

template <bool A_KC, bool B_KC, bool DROP>
static int launch_fast(const GemmArgs& g, int dt_out, hipStream_t s) {
  static RmclLdsOnce once_f, once_b;
  RMCL_TRY(rmcl_set_max_lds(once_f, reinterpret_cast<const void*>(gemm_fast_kernel<A_KC, B_KC, float, DROP>), 2 * STAGE_BYTES));
  RMCL_TRY(rmcl_set_max_lds(once_b, reinterpret_cast<const void*>(gemm_fast_kernel<A_KC, B_KC, bf16_t, DROP>), 2 * STAGE_BYTES));
  const int tm = cdiv(g.M, FBM), tn = g.N / FBN, nwg = tm * tn;
  const int total = nwg * (g.splitk > 1 ? g.splitk : 1);
  int per_cu = (g_gemm_cfg >= 1 && g_gemm_cfg <= 4) ? g_gemm_cfg : 2;
  const int grid = std::min(total, 256 * per_cu);             // multiple of 8 whenever it is < total
  if (dt_out == RMCL_F32) RMCL_LAUNCH((gemm_fast_kernel<A_KC, B_KC, float, DROP>), dim3(grid), dim3(256), 2 * STAGE_BYTES, s, g, tn, nwg, total);
  else RMCL_LAUNCH((gemm_fast_kernel<A_KC, B_KC, bf16_t, DROP>), dim3(grid), dim3(256), 2 * STAGE_BYTES, s, g, tn, nwg, total);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_launch_gemm_pp(const GemmArgs& g, int dt_out, hipStream_t s);
int rmcl_launch_gemm_sw(const GemmArgs& g, int dt_out, int b_kc, hipStream_t s);
bool rmcl_gemm_sw_supported(const GemmArgs& g, int a_kc, int b_kc);
double rmcl_gemm_sw_fill(const GemmArgs& g, int cus);
int rmcl_launch_gemm_st(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s);
double rmcl_gemm_st_fill(const GemmArgs& g, int cus);
int g_gemm_share = 1;               // rmcl_tune_set key 10: independent chains sharing the chip (half-batch lanes: 2) - a launch is sized against 1 / share of the CUs
bool rmcl_gemm_st_supported(const GemmArgs& g, int a_kc, int b_kc);
bool rmcl_gemm_pp_supported(const GemmArgs& g, int a_kc, int b_kc);
bool rmcl_gemm_big_supported(const GemmArgs& g, int a_kc, int b_kc);
int rmcl_launch_gemm_big(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s);

// true when rmcl_launch_gemm_fast sends this GEMM to the 192x384 or the 192x192 kernel (the only ones that implement the
// LayerNorm-folded epilogues EPI_LNFOLD / EPI_ROWSTAT)
// Which kernel family rmcl_launch_gemm_fast gives a GEMM to (the ONE place the routing is decided; rmcl_gemm_route in the C ABI
// reports it so that a parity test can assert which kernels it compared): 0 = 128x128 (gemm_fast), 1 = 192x192 one workgroup per CU
// (gemm_st), 2 = 192x384 (gemm_sw), 3 = 192x192x32 two workgroups per CU (gemm_dp), 4 = 256x256 ping-pong, 5 = 256x256
int rmcl_gemm_route_code(const GemmArgs& g, int dt_out, int a_kc, int b_kc) {
  if (g_gemm_cfg == 80 && rmcl_gemm_dp_supported(g, a_kc, b_kc) && !((g.epi & EPI_RESIDUAL) && dt_out != RMCL_F32)) return 3;
  // 192x384 tiles where they make exact rounds (N = 3072 at M = 64*185: 496 tiles = 2 x 248)
  if ((g_gemm_cfg == 70 || (g_gemm_cfg < 0 && rmcl_gemm_sw_fill(g, 248 / g_gemm_share) >= 0.95)) && rmcl_gemm_sw_supported(g, a_kc, b_kc) &&
      !((g.epi & (EPI_DROPOUT | EPI_DROP_BWD)) && dt_out != RMCL_BF16)) return 2;      // (its dropout epilogues write bf16)
  // 192x192 ping-pong tiles for the activation GEMMs (M = B*185 rows) whenever they fill the CU rounds
  if ((g_gemm_cfg == 60 || (g_gemm_cfg < 0 && rmcl_gemm_st_fill(g, 256 / g_gemm_share) >= 0.7)) && rmcl_gemm_st_supported(g, a_kc, b_kc)) return 1;
  if (g_gemm_cfg == 50 && rmcl_gemm_pp_supported(g, a_kc, b_kc)) return 4;
  // 256x256 tiles where they measure faster (MI355X, M = 11840): narrow outputs with a long reduction
  const bool big_wins = a_kc && g.splitk <= 1 && g.N <= 1024 && g.K >= 2048 && cdiv(g.M, 256) * (g.N / 256) >= 128;
  if ((g_gemm_cfg == 30 || (g_gemm_cfg < 0 && big_wins)) && rmcl_gemm_big_supported(g, a_kc, b_kc)) return 5;
  return 0;
}

// true when the GEMM runs on one of the 192-row tile kernels (the only ones that implement the LayerNorm-folded epilogues
// EPI_LNFOLD / EPI_ROWSTAT)
bool rmcl_gemm_routes_to_tile192(const GemmArgs& g, int a_kc, int b_kc) {
  const int r = rmcl_gemm_route_code(g, (g.epi & EPI_ROWSTAT) ? RMCL_F32 : RMCL_BF16, a_kc, b_kc);
  return r == 1 || r == 2 || r == 3;
}

int rmcl_launch_gemm_fast(const GemmArgs& g0, int dt_out, int a_kc, int b_kc, hipStream_t s) {
  GemmArgs g = g0;
  const int route = rmcl_gemm_route_code(g, dt_out, a_kc, b_kc);
  RMCL_REQUIRE(!(g.epi & (EPI_LNFOLD | EPI_ROWSTAT)) || (route >= 1 && route <= 3),
               "gemm: the LayerNorm-folded epilogues exist in the 192-row tile kernels only");
  if (route == 3) return rmcl_launch_gemm_dp(g, dt_out, s);
  if (route == 2) return rmcl_launch_gemm_sw(g, dt_out, b_kc, s);
  if (route == 1) return rmcl_launch_gemm_st(g, dt_out, a_kc, b_kc, s);
  if (route == 4) return rmcl_launch_gemm_pp(g, dt_out, s);
  if (route == 5) return rmcl_launch_gemm_big(g, dt_out, a_kc, b_kc, s);
  if (g_gemm_cfg == 10) g.tag |= 1 << 30;
  if (g_gemm_cfg == 11) g.tag |= 1 << 29;
  if (g.splitk > 1) {                                          // no empty K slices (the k-tile stream assumes nk >= 1)
    const int kt = g.K / FBK, per = (kt + g.splitk - 1) / g.splitk;
    g.splitk = (kt + per - 1) / per;
  }
  const bool drop = g.epi & (EPI_DROPOUT | EPI_DROP_BWD);
  if (a_kc && b_kc) return drop ? launch_fast<true, true, true>(g, dt_out, s) : launch_fast<true, true, false>(g, dt_out, s);
  if (a_kc && !b_kc) return drop ? launch_fast<true, false, true>(g, dt_out, s) : launch_fast<true, false, false>(g, dt_out, s);
  return launch_fast<false, false, false>(g, dt_out, s);
}

// dW[M=Nout, N=Kin] += A^T B over K tokens with split-K partial slabs (slab: splitk*M*N floats)
int rmcl_launch_gemm_fast_slab(const GemmArgs& g0, float* slab, float* out, hipStream_t s) {
  GemmArgs g = g0;
  g.C = slab;
  g.epi = 0;
  {
    const int kt = g.K / FBK, per = (kt + g.splitk - 1) / g.splitk;
    g.splitk = std::max(1, (kt + per - 1) / per);
  }
  RMCL_TRY(rmcl_launch_gemm_fast(g, RMCL_F32, 0, 0, s));
  return rmcl_slab_reduce(slab, out, (long)g.M * g.N, g.splitk, s);
}
