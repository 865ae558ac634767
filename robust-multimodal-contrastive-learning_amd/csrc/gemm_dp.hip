// bf16 MFMA GEMM with 192x192x32 block tiles, TWO persistent 4-wave workgroups per CU ("dual-persistent").
//
// Why: in gemm_st.hip / gemm_sw.hip (one 8-wave workgroup per CU) the k-loop and the epilogue of a tile are serial on every CU -
// the epilogue (bias / LayerNorm fold / GELU / stash / residual + the HBM-bound stores, 4-20 us per tile) runs with the MFMAs
// idle, and `vmcnt` retires stores and LDS-DMA in issue order, so the next tile's first counted wait also waits for the stores
// (DESIGN.md section 3, items 12-13).  Here a workgroup needs 72 KiB of LDS and one wave per SIMD, so two of them are resident per
// CU and drift apart by themselves: while one is in its epilogue or waits for its stores, the other one's k-loop owns the MFMAs.
//
//   * tile 192 x 192 (one sample's 185 tokens per row tile, like gemm_st.hip), k-step 32: a stage = A [192][32] + B [192][32]
//     bf16 = 24 KiB, three stages = 72 KiB; LDS-DMA (16 B per lane) two stages ahead, counted vmcnt, ONE s_barrier per k-step.
//   * LDS image of an operand unit: 96 "virtual rows" of 128 B, virtual row v = tile rows 2v, 2v+1 (64 B = 32 k each); the
//     16-byte chunk lc = (row & 1) * 4 + k / 8 of virtual row v sits at chunk lc ^ (v & 7) - the swizzle is applied to the
//     SOURCE address of the DMA (the LDS destination of a wave-instruction is linear) and to the ds_read_b128 address; a
//     16-lane read group then touches 16 different 16-byte slots of the 256-byte bank row: no conflicts.
//   * wave w owns columns w*48 .. w*48+47 of all 192 rows: 12 x 3 fragments of 16x16 (144 accumulator registers), 15
//     ds_read_b128 + 36 v_mfma_f32_16x16x32_bf16 per k-step.  The epilogue is gemm_st.hip's (gemm_st_epi.h), run for the two row
//     halves one after the other.
//   * persistent: a workgroup walks its tiles as one stream of k-steps; the last two k-steps of a tile stage the first two of
//     the next one.
// Form: A [M,K] x B [N,K]^T (forward GEMMs; data gradients through the transposed weight shadows).
#include "gemm_st_epi.h"

#define DP_UNIT 12288                       // one operand unit: 192 rows x 32 bf16
#define DP_STAGE (2 * DP_UNIT)              // 24 KiB
#define DP_LDS (3 * DP_STAGE)               // 72 KiB

template <int N>
__device__ __forceinline__ void dp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct DPCtx {
  const bf16_t* A;
  const bf16_t* B;
  uint32_t oa[3], ob[3];
  long ksa, ksb;                            // element stride of one k-step: 32 ([rows][K] operand) or rows * 32 (k-blocked operand)
  int aoff, boff;                           // lane part of the fragment read address (A: + i * 1024; B: wave column included, + j * 1024)
  int wave;
};

// one operand unit = 12 wave-instructions of 1 KiB (8 virtual rows); this wave issues instructions wave*3 .. wave*3+2
__device__ __forceinline__ void dp_stage_unit(const bf16_t* base, const uint32_t (&off)[3], long k0, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 3; ++q)
    __builtin_amdgcn_global_load_lds((glb_void*)(base + k0 + off[q]), (lds_void*)(lds_unit + (wave * 3 + q) * 1024), 16, 0, 0);
}

// One k-step.  `stage`: the stream continues two k-steps ahead (same tile or the next one): stage it first, and wait with
// vmcnt(6) after the MFMAs (k-step s+1 has landed, s+2 may be in flight); otherwise vmcnt(0).  ONE code path for every k-step of
// the stream: the 144 accumulator registers then flow through a single loop body (separate tail steps on the two sides of a branch
// made the register allocator copy the whole accumulator tile through scratch).
__device__ __forceinline__ void dp_step(f32x4 (&acc)[2][6][3], const DPCtx& c, const char* cur, char* nxt2, int k2, bool stage) {   // k2: index of the k-step to stage
  if (stage) {
    // the slot of k-step s+2 held k-step s-1: every wave retired its reads of it before the barrier that ended k-step s-1
    dp_stage_unit(c.A, c.oa, k2 * c.ksa, nxt2, c.wave);
    dp_stage_unit(c.B, c.ob, k2 * c.ksb, nxt2 + DP_UNIT, c.wave);
  }
  __builtin_amdgcn_sched_barrier(0);
  // A fragments stream through a few registers (144 accumulator registers leave room for ~60 more): the three B fragments and
  // the first three A fragments are read up front, then every A fragment's three MFMAs are followed by the read of the A
  // fragment three ahead (sched_group_barrier pins that interleave, and with it the live ranges)
  bf16x8 a[12], b[3];
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const bf16x8*>(cur + DP_UNIT + c.boff + j * 1024);
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = *reinterpret_cast<const bf16x8*>(cur + c.aoff + i * 1024);
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i / 6][i % 6][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i / 6][i % 6][j], 0, 0, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);           // DS read x 6
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);         // MFMA x 3
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read x 1
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  if (stage) dp_wait_vm<6>();
  else dp_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (all fragments were consumed by the MFMAs above; explicit for the WAR below)
  __builtin_amdgcn_s_barrier();                                // k-step s+1 visible to all waves; all reads of k-step s retired
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void dp_tile_setup(STTile& T, const GemmArgs& g, int id, int tiles_m, int tiles_n, int rows_per_tile, int wave, int lane) {
  asm volatile("" : "+v"(lane));   // recompute the lane-derived offsets per tile instead of keeping them live across the k-loops
  int tr, tc;
  if (tiles_n % 4 == 0) {          // bands of 4 column tiles, row tiles inside a band, the band's column tiles innermost (gemm_st.hip)
    const int band = id / (tiles_m * 4), rem = id - band * (tiles_m * 4);
    tr = rem >> 2;
    tc = band * 4 + (rem & 3);
  } else {
    tr = id / tiles_n;
    tc = id - tr * tiles_n;
  }
  T.m0 = tr * rows_per_tile;
  T.n0 = tc * ST_T;
  T.m_end = min(g.M, T.m0 + rows_per_tile);
  T.kt0 = 0;
  T.nk = g.K / 32;
  T.zoff = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int v = (wave * 3 + q) * 8 + (lane >> 3);            // virtual row of the unit
    const int lc = (lane & 7) ^ (v & 7);                       // logical chunk stored at physical chunk lane & 7
    const int row = 2 * v + (lc >> 2), kc = lc & 3;
    T.oa[q] = (uint32_t)min(T.m0 + row, g.M - 1) * ((g.kblk & 1) ? 32u : (uint32_t)g.lda) + kc * 8;
    T.ob[q] = (uint32_t)min(T.n0 + row, g.N - 1) * ((g.kblk & 2) ? 32u : (uint32_t)g.ldb) + kc * 8;
  }
}

template <int AUX, typename TO, int LNF>
__global__ __launch_bounds__(256, 2) void gemm_dp_kernel(GemmArgs g, int tiles_m, int tiles_n, int rows_per_tile, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = tiles_m * tiles_n, G = gridDim.x, bidx = blockIdx.x;

  DPCtx c;
  c.A = reinterpret_cast<const bf16_t*>(g.A);
  c.B = reinterpret_cast<const bf16_t*>(g.B);
  c.wave = wave;
  c.ksa = (g.kblk & 1) ? (long)g.M * 32 : 32;
  c.ksb = (g.kblk & 2) ? (long)g.N * 32 : 32;
  {
    const int vr = (lane & 15) >> 1;                           // virtual row & 7 of this lane's fragment row
    const int lp = ((((lane & 1) << 2) | (lane >> 4)) ^ vr) * 16 + vr * 128;
    c.aoff = lp;
    c.boff = wave * 3072 + lp;                                 // 48 rows = 24 virtual rows per wave column
  }

  int id = st_tile_id(bidx, 0, G, ntiles);
  if (id < 0) return;                                          // (whole workgroup: id is uniform)
  // Two workgroups that start together on a CU run the same number of k-steps and would reach their epilogues together: the second
  // half of the grid (observed placement: block b and b + #CUs share a CU; speed only) starts `stagger` ticks of the 100 MHz clock late
  if (stagger > 0 && bidx >= (G + 1) / 2) {
    const long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < stagger) __builtin_amdgcn_s_sleep(16);
  }
  STTile cur, nxt;
  dp_tile_setup(cur, g, id, tiles_m, tiles_n, rows_per_tile, wave, lane);
#pragma unroll
  for (int q = 0; q < 3; ++q) { c.oa[q] = cur.oa[q]; c.ob[q] = cur.ob[q]; }

  // prologue: the first two k-steps of the first tile
  dp_stage_unit(c.A, c.oa, 0, smem, wave);
  dp_stage_unit(c.B, c.ob, 0, smem + DP_UNIT, wave);
  dp_stage_unit(c.A, c.oa, c.ksa, smem + DP_STAGE, wave);
  dp_stage_unit(c.B, c.ob, c.ksb, smem + DP_STAGE + DP_UNIT, wave);
  dp_wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int sc = 0, sn = 2;                                          // LDS slot of the current k-step / of the k-step two ahead
  for (int r = 0;; ++r) {
    const int nid = st_tile_id(bidx, r + 1, G, ntiles);
    f32x4 acc[2][6][3];                                        // [row half][row fragment][column fragment]
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i / 6][i % 6][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = cur.nk;
    for (int it = 0; it < nk; ++it) {
      const int ahead = it + 2 - nk;                           // >= 0: k-step `ahead` of the NEXT tile is the one to stage
      if (ahead == 0 && nid >= 0) {                            // the stream continues with the next tile
        dp_tile_setup(nxt, g, nid, tiles_m, tiles_n, rows_per_tile, wave, lane);
#pragma unroll
        for (int q = 0; q < 3; ++q) { c.oa[q] = nxt.oa[q]; c.ob[q] = nxt.ob[q]; }
      }
      dp_step(acc, c, smem + sc * DP_STAGE, smem + sn * DP_STAGE, ahead < 0 ? it + 2 : ahead, ahead < 0 || nid >= 0);
      sc = sc == 2 ? 0 : sc + 1;
      sn = sn == 2 ? 0 : sn + 1;
    }
    // the slot the tile's last k-step was read from (sc has already moved on when the stream continues): 24 KiB of scratch for
    // the epilogue's row images; the next tile's k-steps 0 / 1 sit in the other two slots
    const int s_free = sc == 0 ? 2 : sc - 1;
    char* scratch = smem + s_free * DP_STAGE;
    float* rowstat = reinterpret_cast<float*>(smem + DP_LDS);
    st_epilogue_lds<AUX, TO, false, LNF>(acc[0], g, cur, 0, wave, lane, wave, scratch, rowstat);
    st_epilogue_lds<AUX, TO, false, LNF>(acc[1], g, cur, 1, wave, lane, wave, scratch, rowstat);
    if (nid < 0) break;
    cur = nxt;
  }
}

bool rmcl_gemm_dp_supported(const GemmArgs& g, int a_kc, int b_kc) {
  if (!a_kc || !b_kc || g.nb1 > 1 || g.nb2 > 1 || g.splitk > 1) return false;
  if (g.N % ST_T != 0 || g.K % 32 != 0 || g.K < 64) return false;
  if ((long)g.M * g.lda >= (1L << 31) || (long)g.N * g.ldb >= (1L << 31) || (long)g.M * g.ldc >= (1L << 31)) return false;
  if (g.epi & ~(EPI_BIAS | EPI_GELU | EPI_SAVE_PREACT | EPI_RESIDUAL | EPI_DGELU | EPI_LNFOLD | EPI_ROWSTAT)) return false;
  if ((g.epi & EPI_RESIDUAL) && (g.epi & EPI_DGELU)) return false;
  if (g.epi & EPI_LNFOLD) {
    if ((g.epi & ~(EPI_LNFOLD | EPI_GELU | EPI_SAVE_PREACT)) || !g.ln_s || !g.ln_c || !g.ln_part || g.ln_nparts % 2 || g.ln_nparts <= 0 || g.ln_cols <= 0) return false;
  }
  if (g.epi & EPI_ROWSTAT) {
    if ((g.epi & ~(EPI_ROWSTAT | EPI_BIAS | EPI_RESIDUAL)) || !(g.epi & EPI_RESIDUAL) || !g.ln_part || !g.C2 || g.ln_nparts != 4 * (g.N / ST_T)) return false;
  }
  return true;
}

// tiles per workgroup slot (2 slots per CU): how full the rounds of a launch are
double rmcl_gemm_dp_fill(const GemmArgs& g, int cus) {
  const long tiles = (long)cdiv(g.M, ST_T) * (g.N / ST_T), slots = 2L * cus;
  return (double)tiles / (double)(cdiv(tiles, slots) * slots);
}

extern int g_st_reserve_cus;
int g_dp_stagger = 0;                 // rmcl_tune_set key 7: start delay of the second workgroup of every CU, 100 MHz ticks

template <int AUX, typename TO, int LNF>
static int launch_dp3(const GemmArgs& g, hipStream_t s) {
  constexpr int LDS = DP_LDS + 2048;
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>((gemm_dp_kernel<AUX, TO, LNF>)), LDS));
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  const int tm = cdiv(g.M, ST_T), tn = g.N / ST_T, rows = cdiv(g.M, tm);
  const int grid = std::min(tm * tn, 2 * std::max(8, ncu - g_st_reserve_cus));
  RMCL_LAUNCH((gemm_dp_kernel<AUX, TO, LNF>), dim3(grid), dim3(256), LDS, s, g, tm, tn, rows, g_dp_stagger);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_launch_gemm_dp(const GemmArgs& g, int dt_out, hipStream_t s) {
  if (g.epi & EPI_LNFOLD) {
    RMCL_REQUIRE(dt_out == RMCL_BF16, "gemm_dp: the LayerNorm-folded form writes bf16");
    return launch_dp3<ST_AUX_NONE, bf16_t, 1>(g, s);
  }
  if (g.epi & EPI_ROWSTAT) {
    RMCL_REQUIRE(dt_out == RMCL_F32, "gemm_dp: the row-statistics producer writes the fp32 residual stream");
    return launch_dp3<ST_AUX_RES, float, 2>(g, s);
  }
  if (g.epi & EPI_RESIDUAL) {
    RMCL_REQUIRE(dt_out == RMCL_F32, "gemm_dp: the residual epilogue writes fp32");
    return launch_dp3<ST_AUX_RES, float, 0>(g, s);
  }
  if (g.epi & EPI_DGELU) return dt_out == RMCL_F32 ? launch_dp3<ST_AUX_DGELU, float, 0>(g, s) : launch_dp3<ST_AUX_DGELU, bf16_t, 0>(g, s);
  return dt_out == RMCL_F32 ? launch_dp3<ST_AUX_NONE, float, 0>(g, s) : launch_dp3<ST_AUX_NONE, bf16_t, 0>(g, s);
}
