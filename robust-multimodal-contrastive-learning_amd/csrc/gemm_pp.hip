// bf16 MFMA GEMM, 256x256x64 block tile, 8 waves in two groups that run half a phase apart ("ping-pong"):
// while the four waves of one group (one per SIMD) issue their MFMA cluster, the four waves of the other
// group issue the LDS fragment reads and the global_load_lds of their next cluster.  One workgroup per CU.
//
//   * LDS: 2 buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 64 k (16 KiB each) = 128 KiB, one array.
//     Image per half-tile: [row][128 B], 16-byte chunk c of row r stored at chunk c ^ (r & 7) (applied on the
//     SOURCE address of the LDS-DMA, and on the ds_read address) -> conflict-free ds_read_b128.
//   * wave (wm, wn) owns rows {h*128 + wm*64 + 0..63 : h = 0,1} and columns {h*128 + wn*32 + 0..31 : h = 0,1}
//     of the block tile, so that a phase touches ONE A half-tile and ONE B half-tile:
//        P1: read A0 (8 x b128) + B0 (4)   MFMA rows(A0) x cols(B0)
//        P2: read B1 (4)                   MFMA rows(A0) x cols(B1)
//        P3: read A1 (8)                   MFMA rows(A1) x cols(B1)
//        P4: (B0 fragments kept)           MFMA rows(A1) x cols(B0)
//     16 MFMA 16x16x32 per phase per wave.
//   * every phase also issues ONE half-tile of LDS-DMA (2 global_load_lds per wave), always into a half-tile
//     whose last ds_read is >= 2 phases old, and at least 4 phases before its first read:
//        (t,P1): B1(t+1)   (t,P2): A1(t+1)   (t,P3): B0(t+2)   (t,P4): A0(t+2)
//     so `s_waitcnt vmcnt(8)` (the 4 youngest half-tiles may be in flight) before the phase's first barrier is
//     the only wait the k-loop needs; it is never 0 until the last k-tile.
//   * barriers are raw s_barrier (a __syncthreads would drain the DMA queue); group 1 executes one extra barrier
//     at the start and group 0 one at the end, which is what skews the two groups.
#include "rmcl_common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define PP_HALF 16384
#define PP_A0 0
#define PP_A1 PP_HALF
#define PP_B0 (2 * PP_HALF)
#define PP_B1 (3 * PP_HALF)
#define PP_BUF (4 * PP_HALF)
#define PP_GROUP_M 4

template <int N>
__device__ __forceinline__ void pp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PPStage {
  const bf16_t* A;
  const bf16_t* B;
  uint32_t oa[2][2], ob[2][2];   // element offsets [half][q] of this lane's 16-byte source chunk (k = 0)
  int wave;
};

// one half-tile (2 LDS-DMA instructions per wave)
__device__ __forceinline__ void pp_stage(const bf16_t* base, const uint32_t (&off)[2], int k0, char* lds_half, int wave) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
    __builtin_amdgcn_global_load_lds((glb_void*)(base + k0 + off[q]), (lds_void*)(lds_half + (wave * 2 + q) * 1024), 16, 0, 0);
}

__device__ __forceinline__ bf16x8 pp_ld(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

#define PP_PHASE_BEGIN()                                  \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1)
#define PP_PHASE_END()                                    \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0)

// one k-tile (4 phases).  cur = buffer holding tile t, oth = the other buffer.  k1 / k2 = k offsets of tiles t+1 / t+2.
template <int W0, int W1, int W3, bool ST01, bool ST23>
__device__ __forceinline__ void pp_tile(f32x4 (&acc)[8][4], const PPStage& st, char* cur, char* oth, int k1, int k2,
                                        const int (&aoff)[2], const int (&boff)[2]) {
  bf16x8 a[4][2], b[4][2];
  // ---- P1
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) b[j][s] = pp_ld(cur + PP_B0 + j * 2048 + boff[s]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[i][s] = pp_ld(cur + PP_A0 + i * 2048 + aoff[s]);
  if (ST01) pp_stage(st.B, st.ob[1], k1, oth + PP_B1, st.wave);
  pp_wait_vm<W0>();
  PP_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[i][j], 0, 0, 0);
  PP_PHASE_END();
  // ---- P2
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) b[2 + j][s] = pp_ld(cur + PP_B1 + j * 2048 + boff[s]);
  if (ST01) pp_stage(st.A, st.oa[1], k1, oth + PP_A1, st.wave);
  pp_wait_vm<W1>();
  PP_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 2; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[i][j], 0, 0, 0);
  PP_PHASE_END();
  // ---- P3
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[i][s] = pp_ld(cur + PP_A1 + i * 2048 + aoff[s]);
  if (ST23) pp_stage(st.B, st.ob[0], k2, cur + PP_B0, st.wave);
  PP_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 2; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[4 + i][j], 0, 0, 0);
  PP_PHASE_END();
  // ---- P4
  if (ST23) pp_stage(st.A, st.oa[0], k2, cur + PP_A0, st.wave);
  pp_wait_vm<W3>();
  PP_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[4 + i][j], 0, 0, 0);
  PP_PHASE_END();
}

template <typename TO>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int m0, n0;
  {
    const int per_group = PP_GROUP_M * tiles_n, group = bid / per_group, first_m = group * PP_GROUP_M;
    const int gsz = min(tiles_m - first_m, PP_GROUP_M), r = bid - group * per_group;
    m0 = (first_m + r % gsz) * 256;
    n0 = (r / gsz) * 256;
  }
  const int nk = g.K / 64;

  PPStage st;
  st.A = reinterpret_cast<const bf16_t*>(g.A);
  st.B = reinterpret_cast<const bf16_t*>(g.B);
  st.wave = wave;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wave * 2 + q) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      st.oa[h][q] = (uint32_t)min(m0 + h * 128 + row, g.M - 1) * (uint32_t)g.lda + chunk * 8;
      st.ob[h][q] = (uint32_t)min(n0 + h * 128 + row, g.N - 1) * (uint32_t)g.ldb + chunk * 8;
    }
  int aoff[2], boff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sw = ((4 * s + (lane >> 4)) ^ (lane & 7)) * 16;
    aoff[s] = (wm * 64 + (lane & 15)) * 128 + sw;
    boff[s] = (wn * 32 + (lane & 15)) * 128 + sw;
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  char* buf0 = smem;
  char* buf1 = smem + PP_BUF;
  // prologue: all of tile 0, then B0/A0 of tile 1 (the slots phases (-1,P3) and (-1,P4) would have issued)
  pp_stage(st.B, st.ob[0], 0, buf0 + PP_B0, wave);
  pp_stage(st.A, st.oa[0], 0, buf0 + PP_A0, wave);
  pp_stage(st.B, st.ob[1], 0, buf0 + PP_B1, wave);
  pp_stage(st.A, st.oa[1], 0, buf0 + PP_A1, wave);
  pp_stage(st.B, st.ob[0], 64, buf1 + PP_B0, wave);
  pp_stage(st.A, st.oa[0], 64, buf1 + PP_A0, wave);
  pp_wait_vm<4>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wm == 1) __builtin_amdgcn_s_barrier();                 // skew: group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);

  int it = 0;
  for (; it + 2 < nk; ++it) {
    char* cur = (it & 1) ? buf1 : buf0;
    char* oth = (it & 1) ? buf0 : buf1;
    pp_tile<8, 8, 8, true, true>(acc, st, cur, oth, (it + 1) * 64, (it + 2) * 64, aoff, boff);
  }
  {
    char* cur = (it & 1) ? buf1 : buf0;
    char* oth = (it & 1) ? buf0 : buf1;
    pp_tile<8, 8, 4, true, false>(acc, st, cur, oth, (it + 1) * 64, 0, aoff, boff);
    pp_tile<2, 0, 0, false, false>(acc, st, oth, cur, 0, 0, aoff, boff);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();

  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + 4 * (lane >> 4);
      float v[4] = {g.alpha * acc[i][j][0], g.alpha * acc[i][j][1], g.alpha * acc[i][j][2], g.alpha * acc[i][j][3]};
      if (epi & EPI_BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      const long ci = (long)m * g.ldc + n;
      if constexpr (sizeof(TO) == 2) {
        uint2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(C + ci) = pk;
      } else {
        *reinterpret_cast<float4*>(C + ci) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

bool rmcl_gemm_pp_supported(const GemmArgs& g, int a_kc, int b_kc) {
  if (!a_kc || !b_kc || g.splitk > 1 || g.nb1 > 1 || g.nb2 > 1) return false;
  if (g.N % 256 != 0 || g.K % 64 != 0 || g.K < 128) return false;
  if ((g.epi & ~EPI_BIAS) != 0) return false;
  if ((long)g.M * g.lda >= (1L << 31) || (long)g.N * g.ldb >= (1L << 31)) return false;
  return true;
}

int rmcl_launch_gemm_pp(const GemmArgs& g, int dt_out, hipStream_t s) {
  static RmclLdsOnce once_f, once_b;
  RMCL_TRY(rmcl_set_max_lds(once_f, reinterpret_cast<const void*>(gemm_pp_kernel<float>), 2 * PP_BUF));
  RMCL_TRY(rmcl_set_max_lds(once_b, reinterpret_cast<const void*>(gemm_pp_kernel<bf16_t>), 2 * PP_BUF));
  const int tm = cdiv(g.M, 256), tn = g.N / 256;
  if (dt_out == RMCL_F32) RMCL_LAUNCH((gemm_pp_kernel<float>), dim3(tm * tn), dim3(512), 2 * PP_BUF, s, g, tm, tn);
  else RMCL_LAUNCH((gemm_pp_kernel<bf16_t>), dim3(tm * tn), dim3(512), 2 * PP_BUF, s, g, tm, tn);
  RMCL_CHECK_LAUNCH();
  return 0;
}
