// bf16 MFMA GEMM with 192x192x64 block tiles, one PERSISTENT workgroup (8 waves) per CU, three 48 KiB LDS stages.
//
// Why this tile: the activations of one RMCL step have M = B * 185 rows (185 tokens per sample) and N = 768,
// 2304 or 3072 = 4, 12, 16 x 192 columns.  With `rows_per_tile` = 185 (each row tile = one sample, 7 of the 192
// tile rows idle; the launcher uses the fewest row tiles, 62 of 191 rows) B = 64 gives 62 x {4, 12, 16} = 248 x {1, 3, 4}
// tiles = exact rounds of a 248-workgroup grid, where 128x128 / 256x256 tiles leave 13-45 % of the chip idle in the
// last round.  Forms: A [M,K] x B [N,K] (forward), A [M,K] x B [K,N] (dX; B through ds_read_b64_tr_b16),
// A [K,M] x B [K,N] over split-K work items (weight gradients -> fp32 slabs); epilogues incl. dropout as separate
// instantiations.
//
// Schedule ("ping-pong", cf. gemm_pp.hip): waves 0-3 (group 0, wave rows 0..95) and waves 4-7 (group 1, rows
// 96..191) run half a phase apart - group 1 executes one extra s_barrier first - so that one group's 18-MFMA
// cluster overlaps the other's 9 ds_read_b128 + LDS-DMA issue.  A k-tile is two phases (k-steps of 32):
//     P1: read fragments of k-step 0; stage the A half of tile t+2 (3 LDS-DMA per wave) into stage (t+2)%3, whose last
//         reads, (t-1,P2), were retired (lgkmcnt(0)) before that phase's first barrier;  barrier, 18 MFMA, barrier
//     P2: read fragments of k-step 1; stage the B half of tile t+2; s_waitcnt vmcnt(6) -> tile t+1 has landed, first read
//         one phase later.
// Barriers are raw s_barrier, so the DMA of tile t+2 stays in flight across them.  The transposed-read forms use ONE
// phase per k-tile instead (PH = 1, see st_tile).
//
// Persistence: a workgroup walks its output tiles (round r: tile r * grid + xcd-aware slot) as ONE stream of k-tiles -
// the last two k-tiles of an output tile already stage the first two k-tiles of the next one, so the epilogue
// (bias / GELU / residual / stores) of tile n runs while the operands of tile n+1 land.
#include "rmcl_common.h"
#include "kernels.h"

#define ST_OP_BYTES (ST_T * 128)            // one operand tile: 192 rows x 64 bf16 = 24 KiB
#define ST_STAGE (2 * ST_OP_BYTES)          // 48 KiB
#define ST_LDS (3 * ST_STAGE)               // 144 KiB


// In-kernel phase trace (developer builds only: hipcc ... -DST_TRACE): lane 0 of waves 0 and 4 of workgroup ST_TRACE_WG stamps
// the 100 MHz wall clock at phase boundaries; tools/st_trace.py reads the stamps through rmcl_debug_st_trace.
#ifdef ST_TRACE
#ifndef ST_TRACE_WG
#define ST_TRACE_WG 100
#endif
__device__ long long g_st_trace[2][32];
#define ST_STAMP(i)                                                                                              \
  if (blockIdx.x == ST_TRACE_WG && (threadIdx.x & 255) == 0) g_st_trace[threadIdx.x >> 8][i] = wall_clock64()
extern "C" int rmcl_debug_st_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_trace), sizeof(long long) * 64);
}
__device__ long long g_dw_trace[2][32];
#define DW_STAMP(i)                                                                                              \
  if (blockIdx.x == ST_TRACE_WG && (threadIdx.x & 255) == 0) g_dw_trace[threadIdx.x >> 8][i] = wall_clock64()
extern "C" int rmcl_debug_dw_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_trace), sizeof(long long) * 64);
}
#else
#define ST_STAMP(i)
#define DW_STAMP(i)
#endif
#include "gemm_st_epi.h"

template <int N>
__device__ __forceinline__ void st_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one operand tile = 24 wave-instructions of 1 KiB; this wave issues instructions wave*3 .. wave*3+2
// (koff = uniform element offset of the k-tile: k0 for a [rows][K] operand, k0 * ld for a [K][cols] operand)
__device__ __forceinline__ void st_stage_op(const bf16_t* base, const uint32_t (&off)[3], long koff, char* lds_op, int wave) {
#pragma unroll
  for (int q = 0; q < 3; ++q)
    __builtin_amdgcn_global_load_lds((glb_void*)(base + koff + off[q]), (lds_void*)(lds_op + (wave * 3 + q) * 1024), 16, 0, 0);
}

// B stored [K][N] (n contiguous): LDS image [64 k][384 B]; the 32-byte column chunk c of k-row k sits at chunk c ^ g(k),
// g(k) = bit1(k) | bit3(k) << 1.  With 384-byte rows the bank of a row start alternates 0 / 32 with k & 1, so the eight
// 32-byte pieces one half-wave of ds_read_b64_tr_b16 touches (k-rows q, q+8; q = 0..3) land on eight different 8-bank sets.
__device__ __forceinline__ int st_gk(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

template <bool KC>
__device__ __forceinline__ bf16x8 st_ld_b(const char* lds_b, int off, int s) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(lds_b + off);          // (off already holds the k-step swizzle)
  } else {
    // transposed reads by inline asm (rmcl_common.h lds_read_tr16_asm): the builtin made hipcc drain the in-flight LDS-DMA with
    // vmcnt(0) at the head of every k-tile.  Every caller retires them with an explicit lgkmcnt(0) + sched_barrier before the MFMAs.
    union { bf16x8 v; s16x4 h[2]; } u;
    const uint32_t a = lds_addr(lds_b + off);
    if (s == 0) { u.h[0] = lds_read_tr16_asm<0>(a); u.h[1] = lds_read_tr16_asm<4 * 384>(a); }
    else { u.h[0] = lds_read_tr16_asm<32 * 384>(a); u.h[1] = lds_read_tr16_asm<32 * 384 + 4 * 384>(a); }
    return u.v;
  }
}

#define ST_PHASE_BEGIN()                                  \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1)
#define ST_PHASE_END()                                    \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0)

struct STCtx {
  const bf16_t* A;
  const bf16_t* B;
  uint32_t oa[3], ob[3];
  int aoff[2], aoffi[6], boff[2], boffj[3];
  long kstep_a, kstep_b;          // element stride of one k-tile: 64 ([rows][K] operand) or 64 * ld ([K][cols] operand)
  int wave;
};

// MODE 0: steady state (stage tile t+2, vmcnt(6)); 1: second-to-last tile (no stage, vmcnt(0)); 2: last tile (no stage, no wait)
// PH = 2: one phase per k-step (18 MFMA between barriers); PH = 1: ONE phase per k-tile - both k-steps' fragments are read
// up front (72 fragment registers) and 36 MFMAs run between barriers, halving the barrier / role-switch overhead per FLOP.
// With PH = 1 the DMA into stage (t+2)%3 is issued one phase after that stage's last ds_reads, so those reads are retired
// (lgkmcnt(0)) BEFORE the phase's first barrier rather than after it.
template <int MODE, bool A_KC, bool B_KC, int PH>
__device__ __forceinline__ void st_tile(f32x4 (&acc)[6][3], const STCtx& c, const char* cur, char* nxt2, int t2) {
  if constexpr (PH == 2) {
    bf16x8 a[6], b[3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = B_KC ? st_ld_b<true>(cur + ST_OP_BYTES, j * 2048 + c.boff[s], s) : st_ld_b<false>(cur + ST_OP_BYTES, c.boffj[j], s);
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i] = A_KC ? st_ld_b<true>(cur, i * 2048 + c.aoff[s], s) : st_ld_b<false>(cur, c.aoffi[i], s);
      if (MODE == 0) {
        // half of tile t+2 per phase (A with k-step 0, B with k-step 1): the A half is issued ONE phase after the stage's last
        // reads, which is why those reads are retired before the barrier below
        if (s == 0) st_stage_op(c.A, c.oa, (long)t2 * c.kstep_a, nxt2, c.wave);
        else { st_stage_op(c.B, c.ob, (long)t2 * c.kstep_b, nxt2 + ST_OP_BYTES, c.wave); st_wait_vm<6>(); }
      } else if (MODE == 1 && s == 1) {
        st_wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ST_PHASE_BEGIN();
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      ST_PHASE_END();
    }
  } else {
    bf16x8 a[2][6], b[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 3; ++j) b[s][j] = B_KC ? st_ld_b<true>(cur + ST_OP_BYTES, j * 2048 + c.boff[s], s) : st_ld_b<false>(cur + ST_OP_BYTES, c.boffj[j], s);
#pragma unroll
      for (int i = 0; i < 6; ++i) a[s][i] = A_KC ? st_ld_b<true>(cur, i * 2048 + c.aoff[s], s) : st_ld_b<false>(cur, c.aoffi[i], s);
    }
    if (MODE == 0) {
      st_stage_op(c.A, c.oa, (long)t2 * c.kstep_a, nxt2, c.wave);
      st_stage_op(c.B, c.ob, (long)t2 * c.kstep_b, nxt2 + ST_OP_BYTES, c.wave);
      st_wait_vm<6>();
    } else if (MODE == 1) {
      st_wait_vm<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this stage's reads are done before anyone may overwrite it
    ST_PHASE_BEGIN();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[s][j], a[s][i], acc[i][j], 0, 0, 0);
    ST_PHASE_END();
  }
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void st_tile_setup(STTile& T, const GemmArgs& g, int item, int tiles_m, int tiles_n, int rows_per_tile, int wave, int lane,
                                              int xflags) {
  asm volatile("" : "+v"(lane));   // recompute the lane-derived offsets per tile instead of keeping them live across the k-loops
  // split-K: item = split * tiles + tile; each split owns `per` k-tiles (the last one the remainder, >= 2 by construction)
  // (the integer divisions of this set-up sit in front of a launch's first LDS-DMA: the common cases - no split-K, one band - skip theirs)
  int id = item;
  T.kt0 = 0;
  T.nk = g.K / 64;
  T.zoff = 0;
  if (g.splitk > 1) {
    const int ntile = tiles_m * tiles_n, split = item / ntile;
    id = item - split * ntile;
    const int kts = g.K / 64, per = (kts + g.splitk - 1) / g.splitk;
    T.kt0 = split * per;
    T.nk = min(per, kts - T.kt0);
    T.zoff = (long)split * g.M * g.ldc;
  }
  // tile order: bands of 4 column tiles, row tiles inside a band, the band's 4 column tiles innermost - the 32 consecutive
  // tiles one XCD works on in a round are 8 A panels x 4 B panels (12 x 192 x K x 2 B: fits its 4 MiB L2 for K = 768)
  int tr, tc;
  if (!(xflags & 2) && tiles_n == 4) {
    tr = id >> 2;
    tc = id & 3;
  } else if (!(xflags & 2) && tiles_n % 4 == 0) {
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
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = (wave * 3 + q) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const int lin = (wave * 3 + q) * 64 + lane;              // 16-byte chunk index in the [64][24] image of a [K][cols] operand
    const int k = lin / 24, c16 = (lin - k * 24) ^ (st_gk(k) << 1);
    if (A_KC) T.oa[q] = (uint32_t)min(T.m0 + row, g.M - 1) * (uint32_t)g.lda + chunk * 8;
    else T.oa[q] = (uint32_t)k * (uint32_t)g.lda + T.m0 + c16 * 8;
    if (B_KC) T.ob[q] = (uint32_t)min(T.n0 + row, g.N - 1) * (uint32_t)g.ldb + chunk * 8;
    else T.ob[q] = (uint32_t)k * (uint32_t)g.ldb + T.n0 + c16 * 8;
  }
}

// epilogue (order: alpha, bias, gelu'(aux), save pre-activation, gelu, residual, accumulate; as gemm_fast.hip).  All
// loads of the tile are issued first (rows past the tile end read a clamped row), only the stores are predicated.
// DROP: dropout epilogues (counter-based masks, rmcl_common.h) compiled in - a separate instantiation so that the
// parity / headline configuration (drop_rate 0) carries none of their live state.
template <int AUX, typename TO, bool DROP>
__device__ __forceinline__ void st_epilogue(const f32x4 (&acc)[6][3], const GemmArgs& g, const STTile& T, int wm, int wn, int lane) {
  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C) + T.zoff;
  TO* C2 = reinterpret_cast<TO*>(g.C2);
  const int nb = T.n0 + wn * 48 + 4 * (lane >> 4);
  const int mb = T.m0 + wm * 96 + (lane & 15);
  float4 bias[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bias[j] = (epi & EPI_BIAS) ? *reinterpret_cast<const float4*>(g.bias + nb + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 res[AUX == ST_AUX_RES ? 6 : 1][3];
  uint2 pre[AUX == ST_AUX_DGELU ? 6 : 1][3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const long mr = min(mb + i * 16, g.M - 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (AUX == ST_AUX_RES) res[i][j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + mr * g.ld_aux + nb + j * 16);
      if (AUX == ST_AUX_DGELU) pre[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + mr * g.ld_aux + nb + j * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int m = mb + i * 16;
    const bool live = m < T.m_end;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float v[4] = {g.alpha * acc[i][j][0] + bias[j].x, g.alpha * acc[i][j][1] + bias[j].y, g.alpha * acc[i][j][2] + bias[j].z,
                    g.alpha * acc[i][j][3] + bias[j].w};
      if (DROP && (epi & EPI_DROP_BWD)) {                    // mask of the forward's hidden dropout, indexed like the stash
        const uint32_t di = (uint32_t)((long)m * g.ld_aux + nb + j * 16);
        drop_scale4(g.drop_seed, di, g.drop_thresh, g.drop_inv_keep, v[0], v[1], v[2], v[3]);
      }
      if (AUX == ST_AUX_DGELU) {
        const uint2 u = pre[i][j];
        v[0] *= gelu_poly_grad(__uint_as_float(u.x << 16)); v[1] *= gelu_poly_grad(__uint_as_float(u.x & 0xffff0000u));
        v[2] *= gelu_poly_grad(__uint_as_float(u.y << 16)); v[3] *= gelu_poly_grad(__uint_as_float(u.y & 0xffff0000u));
      }
      const long ci = (long)m * g.ldc + nb + j * 16;
      if ((epi & EPI_SAVE_PREACT) && live) st_store4<TO>(C2 + ci, v);
      if (epi & EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_poly(v[r]);
      }
      if (DROP && (epi & EPI_DROPOUT)) {
        drop_scale4(g.drop_seed, (uint32_t)ci, g.drop_thresh, g.drop_inv_keep, v[0], v[1], v[2], v[3]);
      }
      if (AUX == ST_AUX_RES) { v[0] += res[i][j].x; v[1] += res[i][j].y; v[2] += res[i][j].z; v[3] += res[i][j].w; }
      if constexpr (sizeof(TO) == 4) {
        if ((epi & EPI_ACCUM) && live) {
          const float4 o = *reinterpret_cast<const float4*>(C + ci);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
      }
      if (live) st_store4<TO>(C + ci, v);
    }
  }
}


template <bool A_KC, bool B_KC, int AUX, typename TO, bool DROP, int PH, int LNF = 0>
__global__ __launch_bounds__(512) void gemm_st_kernel(GemmArgs g, int tiles_m, int tiles_n, int rows_per_tile, int xflags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = tiles_m * tiles_n * max(g.splitk, 1), G = gridDim.x, bidx = blockIdx.x;   // work items

  STCtx c;
  c.A = reinterpret_cast<const bf16_t*>(g.A);
  c.B = reinterpret_cast<const bf16_t*>(g.B);
  c.wave = wave;
  c.kstep_a = A_KC ? 64 : 64 * g.lda;
  c.kstep_b = B_KC ? 64 : 64 * g.ldb;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int k = 8 * (lane >> 4) + q;
    const int c8 = (wm * 96 + i * 16) / 4 + p;
    c.aoffi[i] = k * 384 + (((c8 >> 1) ^ (st_gk(k) << 1)) * 16) + (c8 & 1) * 8;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sw = ((4 * s + (lane >> 4)) ^ (lane & 7)) * 16;
    c.aoff[s] = (wm * 96 + (lane & 15)) * 128 + sw;
    c.boff[s] = (wn * 48 + (lane & 15)) * 128 + sw;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int k = 8 * (lane >> 4) + q;                       // + 32 s + 4 h: neither changes g(k)
    const int c8 = (wn * 48 + j * 16) / 4 + p;
    c.boffj[j] = k * 384 + (((c8 >> 1) ^ (st_gk(k) << 1)) * 16) + (c8 & 1) * 8;
  }

  ST_STAMP(0);
  int id = st_tile_id(bidx, 0, G, ntiles);
  if (id < 0) return;                                        // (whole workgroup: id is uniform)
  STTile cur, nxt;
  st_tile_setup<A_KC, B_KC>(cur, g, id, tiles_m, tiles_n, rows_per_tile, wave, lane, xflags);
#pragma unroll
  for (int q = 0; q < 3; ++q) { c.oa[q] = cur.oa[q]; c.ob[q] = cur.ob[q]; }

  // prologue: the first two k-tiles of the first work item
  st_stage_op(c.A, c.oa, (long)cur.kt0 * c.kstep_a, smem, wave);
  st_stage_op(c.B, c.ob, (long)cur.kt0 * c.kstep_b, smem + ST_OP_BYTES, wave);
  st_stage_op(c.A, c.oa, (long)(cur.kt0 + 1) * c.kstep_a, smem + ST_STAGE, wave);
  st_stage_op(c.B, c.ob, (long)(cur.kt0 + 1) * c.kstep_b, smem + ST_STAGE + ST_OP_BYTES, wave);
  st_wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wm == 1) __builtin_amdgcn_s_barrier();                 // skew: group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);

  ST_STAMP(1);
  int sc = 0, sn = 2;                                        // LDS stage of the current k-tile / of the k-tile two ahead
  for (int r = 0;; ++r) {
    const int nid = st_tile_id(bidx, r + 1, G, ntiles);
    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = cur.nk;
    for (int it = 0; it + 2 < nk; ++it) {
      st_tile<0, A_KC, B_KC, PH>(acc, c, smem + sc * ST_STAGE, smem + sn * ST_STAGE, cur.kt0 + it + 2);
      sc = sc == 2 ? 0 : sc + 1;
      sn = sn == 2 ? 0 : sn + 1;
    }
    if (nid >= 0) {                                          // the stream continues with the next work item
      st_tile_setup<A_KC, B_KC>(nxt, g, nid, tiles_m, tiles_n, rows_per_tile, wave, lane, xflags);
#pragma unroll
      for (int q = 0; q < 3; ++q) { c.oa[q] = nxt.oa[q]; c.ob[q] = nxt.ob[q]; }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        st_tile<0, A_KC, B_KC, PH>(acc, c, smem + sc * ST_STAGE, smem + sn * ST_STAGE, nxt.kt0 + e);
        sc = sc == 2 ? 0 : sc + 1;
        sn = sn == 2 ? 0 : sn + 1;
      }
    } else {
      st_tile<1, A_KC, B_KC, PH>(acc, c, smem + sc * ST_STAGE, nullptr, 0);
      sc = sc == 2 ? 0 : sc + 1;
      st_tile<2, A_KC, B_KC, PH>(acc, c, smem + sc * ST_STAGE, nullptr, 0);
    }
    ST_STAMP(2);
    // the stage the tile's last k-tile was read from (sc has already moved on when the stream continues): 24 KiB per group
    const int s_free = nid >= 0 ? (sc == 0 ? 2 : sc - 1) : sc;
    st_epilogue_lds<AUX, TO, DROP, LNF>(acc, g, cur, wm, wn, lane, wave, smem + s_free * ST_STAGE + wm * (ST_STAGE / 2),
                                        reinterpret_cast<float*>(smem + ST_LDS));
    if (nid < 0) break;
    cur = nxt;
  }
  ST_STAMP(20);
  if (wm == 0) __builtin_amdgcn_s_barrier();
#ifdef ST_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ST_STAMP(21);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// All weight gradients of ONE encoder layer in ONE launch: dW_g[Nout, Kin] += dY_g^T X_g for the layer's four linear
// layers (fc2, fc1, proj, qkv), the four bias gradients, and the finish of the layer's LayerNorm dgamma / dbeta replicas.
// One workgroup per 192x192 output tile (4x16 + 16x4 + 4x4 + 12x4 = 192 tiles at D = 768), each streaming ALL tokens as
// its K dimension ([K][M] x [K][N] form, both operands through ds_read_b64_tr_b16, one phase per k-tile) and adding its
// tile into the gradient arena in the epilogue: no split-K slabs, no ordered-reduce kernel, no column-sum kernels
// (round 1: 4 GEMM + 4 slab-reduce + 4 colsum launches per layer).  Bitwise reproducible for the weight matrices: a tile
// is accumulated by one workgroup in k order.
// Bias gradient = column sums of dY = the A operand: one extra MFMA per A fragment against a fragment of ones, only on the
// k-tiles with kt % tiles_n == tn (every (row-tile, k-tile) pair is covered exactly once by the tiles_n tiles of that row
// band, so the extra work is spread evenly) and only by one of the four waves that hold the same fragment; float atomics
// (<= tiles_n addends per element).
__global__ __launch_bounds__(512) void gemm_dw_group_kernel(DwGroupArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  DW_STAMP(0);
  // ---- LayerNorm replica finish: workgroups 0 .. 3*nslots-1, 256 columns each --------------------------------------
  if ((int)blockIdx.x < a.nslots * ((a.D + 255) / 256) && t < 256) {
    const int per = (a.D + 255) / 256, si = blockIdx.x / per, c = (blockIdx.x % per) * 256 + t;
    if (c < a.D) {
      const float* base = a.rep + (long)a.slot[si] * (32 * 2 * 1024);
      float sg = 0.f, sb = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) { sg += base[r * 2048 + c]; sb += base[r * 2048 + 1024 + c]; }
      a.G[a.g_gamma[si] + c] += sg;
      a.G[a.g_beta[si] + c] += sb;
    }
  }
  // ---- tile of this workgroup: XCD-aware slot -> tile id (slots of one XCD are consecutive tile ids) ------------------
  const int ntiles = a.tile_base[4];
  const int id = st_tile_id(blockIdx.x, 0, gridDim.x, ntiles);
  if (id < 0) return;
  int gi = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q) gi += (id >= a.tile_base[q]) ? 1 : 0;
  const int local = id - a.tile_base[gi], tn_cnt = a.tiles_n[gi];
  // inside a GEMM: bands of 4 column tiles x all row tiles, like st_tile_setup
  int tr, tc;
  {
    const int tm_cnt = (a.tile_base[gi + 1] - a.tile_base[gi]) / tn_cnt;
    const int band = local / (tm_cnt * 4), rem = local - band * (tm_cnt * 4);
    tr = rem >> 2;
    tc = band * 4 + (rem & 3);
  }
  const int m0 = tr * ST_T, n0 = tc * ST_T;
  GemmArgs g{};
  g.C = a.C[gi]; g.ldc = a.ldc[gi]; g.M = 1 << 30; g.N = 1 << 30; g.alpha = 1.0f; g.epi = EPI_ACCUM; g.lda = a.lda[gi]; g.ldb = a.ldb[gi];

  STCtx c;
  c.A = a.A[gi];
  c.B = a.B[gi];
  c.wave = wave;
  c.kstep_a = 64 * (long)g.lda;
  c.kstep_b = 64 * (long)g.ldb;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int k = 8 * (lane >> 4) + q;
    const int c8 = (wm * 96 + i * 16) / 4 + p;
    c.aoffi[i] = k * 384 + (((c8 >> 1) ^ (st_gk(k) << 1)) * 16) + (c8 & 1) * 8;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int k = 8 * (lane >> 4) + q;
    const int c8 = (wn * 48 + j * 16) / 4 + p;
    c.boffj[j] = k * 384 + (((c8 >> 1) ^ (st_gk(k) << 1)) * 16) + (c8 & 1) * 8;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int lin = (wave * 3 + q) * 64 + lane;              // 16-byte chunk index in the [64][24] image of a [K][cols] operand
    const int k = lin / 24, c16 = (lin - k * 24) ^ (st_gk(k) << 1);
    c.oa[q] = (uint32_t)k * (uint32_t)g.lda + m0 + c16 * 8;
    c.ob[q] = (uint32_t)k * (uint32_t)g.ldb + n0 + c16 * 8;
  }
  const int nk = a.K / 64;
  st_stage_op(c.A, c.oa, 0, smem, wave);
  st_stage_op(c.B, c.ob, 0, smem + ST_OP_BYTES, wave);
  st_stage_op(c.A, c.oa, c.kstep_a, smem + ST_STAGE, wave);
  st_stage_op(c.B, c.ob, c.kstep_b, smem + ST_STAGE + ST_OP_BYTES, wave);
  st_wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wm == 1) __builtin_amdgcn_s_barrier();                 // skew: group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);

  DW_STAMP(1);
  f32x4 acc[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // bias: fragment index pair handled by this wave (the 4 wn waves of a row group hold identical A fragments)
  const int bi0 = wn, bi1 = wn + 4;                            // i = bi0, and i = bi1 when < 6
  bf16x8 ones;
  {
    union { bf16x8 v; uint32_t w[4]; } u;
    u.w[0] = u.w[1] = u.w[2] = u.w[3] = 0x3f803f80u;
    ones = u.v;
  }
  auto bias_step = [&](const char* cur) {                      // column sums of the A tile in LDS stage `cur` (both k-steps)
    bf16x8 f0[2], f1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i == bi0) f0[s] = st_ld_b<false>(cur, c.aoffi[i], s);
        if (i == bi1) f1[s] = st_ld_b<false>(cur, c.aoffi[i], s);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (asm reads: no compiler-inserted wait)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f0[s], accb[0], 0, 0, 0);
      if (bi1 < 6) accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f1[s], accb[1], 0, 0, 0);
    }
  };
  int sc = 0, sn = 2, kt = 0;
  int bias_in = tc % tn_cnt;                                   // k-tiles until this tile's next bias turn (kt % tn_cnt == tc)
  for (; kt + 2 < nk; ++kt) {
    if (a.bias[gi] && bias_in == 0) bias_step(smem + sc * ST_STAGE);
    st_tile<0, false, false, 1>(acc, c, smem + sc * ST_STAGE, smem + sn * ST_STAGE, kt + 2);
    bias_in = bias_in == 0 ? tn_cnt - 1 : bias_in - 1;
    sc = sc == 2 ? 0 : sc + 1;
    sn = sn == 2 ? 0 : sn + 1;
  }
  if (a.bias[gi] && bias_in == 0) bias_step(smem + sc * ST_STAGE);
  st_tile<1, false, false, 1>(acc, c, smem + sc * ST_STAGE, nullptr, 0);
  bias_in = bias_in == 0 ? tn_cnt - 1 : bias_in - 1;
  sc = sc == 2 ? 0 : sc + 1;
  if (a.bias[gi] && bias_in == 0) bias_step(smem + sc * ST_STAGE);
  st_tile<2, false, false, 1>(acc, c, smem + sc * ST_STAGE, nullptr, 0);

  DW_STAMP(2);
  STTile T{};
  T.m0 = m0; T.n0 = n0; T.m_end = m0 + ST_T; T.zoff = 0;
  st_epilogue_lds<ST_AUX_NONE, float, false, 0, true>(acc, g, T, wm, wn, lane, wave, smem + sc * ST_STAGE + wm * (ST_STAGE / 2));
  DW_STAMP(3);
  if (a.bias[gi] && lane < 16) {
    float* bp = a.bias[gi] + m0 + wm * 96 + lane;
    atomicAdd(bp + bi0 * 16, accb[0][0]);
    if (bi1 < 6) atomicAdd(bp + bi1 * 16, accb[1][0]);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
#ifdef ST_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DW_STAMP(4);
#endif
}

int rmcl_launch_dw_group(const DwGroupArgs& a, hipStream_t s) {
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(gemm_dw_group_kernel), ST_LDS));
  const int ntiles = a.tile_base[4];
  RMCL_REQUIRE(a.K % 64 == 0 && a.K >= 256, "dw_group: tokens must be a multiple of 64 (>= 256)");
  RMCL_REQUIRE(ntiles >= a.nslots * ((a.D + 255) / 256), "dw_group: too few tiles for the LayerNorm finish");
  RMCL_LAUNCH(gemm_dw_group_kernel, dim3(ntiles), dim3(512), ST_LDS, s, a);
  RMCL_CHECK_LAUNCH();
  return 0;
}

bool rmcl_gemm_st_supported(const GemmArgs& g, int a_kc, int b_kc) {
  if (g.nb1 > 1 || g.nb2 > 1) return false;
  if (g.N % ST_T != 0 || g.K % 64 != 0 || g.K < 128) return false;
  if ((long)(a_kc ? g.M : g.K) * g.lda >= (1L << 31) || (long)(b_kc ? g.N : g.K) * g.ldb >= (1L << 31)) return false;
  if (g.splitk > 1) return false;                                // (split-K: rmcl_launch_gemm_st_slab)
  if ((long)g.M * g.ldc >= (1L << 31)) return false;             // (the LDS epilogue's row stores use 32-bit element offsets)
  if (!a_kc) {                                                   // [K][M] x [K][N] (weight gradients): plain or slab output only
    if (b_kc || g.M % ST_T != 0 || (g.epi & ~EPI_ACCUM)) return false;
    return true;
  }
  if (g.epi & ~(EPI_BIAS | EPI_GELU | EPI_SAVE_PREACT | EPI_RESIDUAL | EPI_DGELU | EPI_ACCUM | EPI_DROPOUT | EPI_DROP_BWD | EPI_LNFOLD | EPI_ROWSTAT)) return false;
  if ((g.epi & EPI_RESIDUAL) && (g.epi & EPI_DGELU)) return false;
  if (g.epi & EPI_LNFOLD) {                                       // consumer: bias-less bf16 GEMM on [rows][K] x [cols][K] (+ GELU / pre-activation stash)
    if (!b_kc || (g.epi & ~(EPI_LNFOLD | EPI_GELU | EPI_SAVE_PREACT)) || ((g.epi & EPI_SAVE_PREACT) && !g.C2) || !g.ln_s || !g.ln_c || !g.ln_part ||
        g.ln_nparts % 2 || g.ln_nparts <= 0 || g.ln_cols <= 0)
      return false;
  }
  if (g.epi & EPI_ROWSTAT) {                                      // producer: fp32 residual-stream output of [rows][K] x [cols][K]
    if (!b_kc || (g.epi & ~(EPI_ROWSTAT | EPI_BIAS | EPI_RESIDUAL | EPI_DROPOUT)) || !(g.epi & EPI_RESIDUAL) || !g.ln_part || !g.C2 || g.ln_nparts != 4 * (g.N / ST_T))
      return false;
  }
  return true;
}

// how full the CU rounds of a launch are (1 = every round uses all 256 CUs)
double rmcl_gemm_st_fill(const GemmArgs& g, int cus) {
  const long tiles = (long)cdiv(g.M, ST_T) * (g.N / ST_T) * (g.splitk > 1 ? g.splitk : 1);
  return (double)tiles / (double)(cdiv(tiles, (long)cus) * cus);
}

int g_st_reserve_cus = 8;            // CUs left to other kernels (rmcl_tune_set key 1; RCCL channels, side-stream kernels): free at M = 64*185
int g_st_xflags = 0;                 // experiment switch (rmcl_tune_set key 0, values 61.. -> xflags = value - 60)

static int st_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

template <bool A_KC, bool B_KC, int AUX, typename TO, bool DROP = false>
static int launch_st3(const GemmArgs& g, hipStream_t s) {
  static RmclLdsOnce once2, once1;
  RMCL_TRY(rmcl_set_max_lds(once2, reinterpret_cast<const void*>(gemm_st_kernel<A_KC, B_KC, AUX, TO, DROP, 2>), ST_LDS));
  RMCL_TRY(rmcl_set_max_lds(once1, reinterpret_cast<const void*>(gemm_st_kernel<A_KC, B_KC, AUX, TO, DROP, 1>), ST_LDS));
  // a tile costs the same whether 185 or 192 of its rows are live, so the fewest row tiles win; they share M evenly
  const int tm = cdiv(g.M, ST_T), tn = g.N / ST_T, rows = A_KC ? cdiv(g.M, tm) : ST_T;
  const int items = tm * tn * (g.splitk > 1 ? g.splitk : 1);
  // M = 64 * 185 gives 62 row tiles: 248 x {1, 3, 4} tiles for N = 768 / 2304 / 3072, so a 248-workgroup grid loses nothing
  // the reserved CUs are given up only while that costs no extra round (the patch-embedding data gradient has 768 tiles: 3 rounds of
  // 256 workgroups, 4 of 248)
  const int ncu = st_num_cus(), gres = max(8, ncu - (A_KC ? g_st_reserve_cus : 0));
  const int grid = min(items, cdiv(items, gres) > cdiv(items, ncu) ? ncu : gres);
  // one phase per k-tile measures 4-5 % faster with transposed-read operands (dX, dW), two phases 1.5 % faster for [rows][K] x [cols][K]
  const bool one_phase = ((g_st_xflags & 4) != 0) != (!A_KC || !B_KC);
  if (one_phase) RMCL_LAUNCH((gemm_st_kernel<A_KC, B_KC, AUX, TO, DROP, 1>), dim3(grid), dim3(512), ST_LDS, s, g, tm, tn, rows, g_st_xflags);
  else RMCL_LAUNCH((gemm_st_kernel<A_KC, B_KC, AUX, TO, DROP, 2>), dim3(grid), dim3(512), ST_LDS, s, g, tm, tn, rows, g_st_xflags);
  RMCL_CHECK_LAUNCH();
  return 0;
}

template <bool B_KC, int AUX>
static int launch_st2(const GemmArgs& g, int dt_out, hipStream_t s) {
  if (g.epi & (EPI_DROPOUT | EPI_DROP_BWD))
    return dt_out == RMCL_F32 ? launch_st3<true, B_KC, AUX, float, true>(g, s) : launch_st3<true, B_KC, AUX, bf16_t, true>(g, s);
  return dt_out == RMCL_F32 ? launch_st3<true, B_KC, AUX, float>(g, s) : launch_st3<true, B_KC, AUX, bf16_t>(g, s);
}

template <bool B_KC>
static int launch_st(const GemmArgs& g, int dt_out, hipStream_t s) {
  if (g.epi & EPI_RESIDUAL) return launch_st2<B_KC, ST_AUX_RES>(g, dt_out, s);
  if (g.epi & EPI_DGELU) return launch_st2<B_KC, ST_AUX_DGELU>(g, dt_out, s);
  return launch_st2<B_KC, ST_AUX_NONE>(g, dt_out, s);
}

template <int AUX, typename TO, int LNF, bool DROP = false>
static int launch_st_lnf(const GemmArgs& g, hipStream_t s) {
  constexpr int LDS = ST_LDS + (LNF != 0 ? 2048 : 0);     // + the epilogue's row statistics (consumer) / row centres (producer)
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>((gemm_st_kernel<true, true, AUX, TO, DROP, 2, LNF>)), LDS));
  const int tm = cdiv(g.M, ST_T), tn = g.N / ST_T, rows = cdiv(g.M, tm);
  const int grid = min(tm * tn, max(8, st_num_cus() - g_st_reserve_cus));
  RMCL_LAUNCH((gemm_st_kernel<true, true, AUX, TO, DROP, 2, LNF>), dim3(grid), dim3(512), LDS, s, g, tm, tn, rows, g_st_xflags);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_launch_gemm_st(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s) {
  if (g.epi & EPI_LNFOLD) {
    RMCL_REQUIRE(a_kc && b_kc && dt_out == RMCL_BF16, "gemm_st: the LayerNorm-folded form is [rows][K] x [cols][K] with bf16 output");
    return launch_st_lnf<ST_AUX_NONE, bf16_t, 1>(g, s);
  }
  if (g.epi & EPI_ROWSTAT) {
    RMCL_REQUIRE(a_kc && b_kc && dt_out == RMCL_F32, "gemm_st: the row-statistics producer writes the fp32 residual stream");
    // (dropout on the branch output, before the residual add: its own instantiation - training-realistic passes only)
    return (g.epi & EPI_DROPOUT) ? launch_st_lnf<ST_AUX_RES, float, 2, true>(g, s) : launch_st_lnf<ST_AUX_RES, float, 2>(g, s);
  }
  if (!a_kc) {
    RMCL_REQUIRE(dt_out == RMCL_F32 && !b_kc, "gemm_st: the [K][M] x [K][N] form writes fp32");
    return launch_st3<false, false, ST_AUX_NONE, float>(g, s);
  }
  return b_kc ? launch_st<true>(g, dt_out, s) : launch_st<false>(g, dt_out, s);
}

// dW-style GEMM: split K into `splitk` slices written as fp32 slabs [splitk][M][N], then reduced (ordered) into `out` (+=).
// Every slice gets at least 2 k-tiles (the k-tile stream's minimum).
int rmcl_launch_gemm_st_slab(const GemmArgs& g0, float* slab, float* out, hipStream_t s) {
  GemmArgs g = g0;
  const int kts = g.K / 64;
  int sk = max(1, min(g.splitk, kts / 2));
  for (;; --sk) {                                                // no empty / single-k-tile last slice
    const int per = cdiv(kts, sk), used = cdiv(kts, per);
    if (used == sk && kts - (sk - 1) * per >= 2) break;
    if (sk == 1) break;
  }
  g.splitk = sk;
  g.C = slab;
  g.ldc = g.N;
  g.epi = 0;
  RMCL_TRY(rmcl_launch_gemm_st(g, RMCL_F32, 0, 0, s));
  return rmcl_slab_reduce(slab, out, (long)g.M * g.N, sk, s);
}


// ---------------------------------------------------------------------------------------------------------------------
// PROTOTYPE (round 4, review item 2): a ROW TILE CARRIED THROUGH SEVERAL GEMMs INSIDE ONE LAUNCH ("chain").
//
// Everything between two attention calls of an encoder layer is row-wise: proj (+residual) -> fc1 (+GELU) -> fc2 (+residual) -> the next
// layer's qkv.  As separate launches every GEMM fills the chip by itself: all workgroups run prologue, k-loop and (HBM-bound) epilogue in
// lockstep, a kernel boundary (drain + ~1.5 us + the write-back of everything the launch left dirty in the L2s) sits between them, and a
// consumer finds its operand in the Infinity Cache at best.  Here a GROUP of four workgroups on ONE XCD (blocks b, b + 8, b + 16, b + 24:
// the dispatcher deals blocks round-robin over the 8 XCDs) owns a row tile (191 rows: one sample) for the whole chain; member c computes
// column tiles c, c + 4, ... of every stage, and the stages are ordered by a global-memory ticket per (stage, row tile): a member adds 1
// after its stores have drained, the next stage starts when the ticket shows all four.  Groups need nothing from each other, so they drift
// apart (one group's store-heavy epilogue under another group's k-loop), the producer -> consumer panel (<= 1.2 MB per group) is read out
// of the XCD's own L2, and there is no grid-wide drain between the GEMMs.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility).  Producer: plain stores -> every storing wave's vmcnt(0) -> workgroup
// barrier -> ONE lane: [flags & 1: agent-scope release = write-back of the XCD L2's dirty lines, placement-independent] -> relaxed
// agent-scope add to the ticket.  Consumer: ONE lane polls the ticket (relaxed, L1-bypassing), then ONE agent-scope acquire (invalidates
// this CU's L1: the operand arrives by LDS-DMA through it) -> vmcnt(0) -> workgroup barrier.  With flags & 1 == 0 the group relies on its
// four members sharing an XCD (their L2 is then the coherence point: no write-back); the launcher's caller verifies the placement
// (`xcc` output) - a wrong guess would be WRONG, not slow, which is why this form stays behind the prototype entry point.
// Spins are bounded (give-up word ticket[CHAIN_ERR]); tickets only grow (target = 4 x launch epoch), so nothing is re-zeroed per launch.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_tile_setup_rc(STTile& T, const GemmArgs& g, int tr, int tc, int rows_per_tile, int wave, int lane) {
  asm volatile("" : "+v"(lane));
  T.kt0 = 0;
  T.nk = g.K / 64;
  T.zoff = 0;
  T.m0 = tr * rows_per_tile;
  T.n0 = tc * ST_T;
  T.m_end = min(g.M, T.m0 + rows_per_tile);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = (wave * 3 + q) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    T.oa[q] = (uint32_t)min(T.m0 + row, g.M - 1) * (uint32_t)g.lda + chunk * 8;
    T.ob[q] = (uint32_t)min(T.n0 + row, g.N - 1) * (uint32_t)g.ldb + chunk * 8;
  }
}

__global__ __launch_bounds__(512) void gemm_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int b = blockIdx.x;
  const int grp = (b & 7) + 8 * (b >> 5), mem = (b >> 3) & 3;   // blocks b, b + 8, b + 16, b + 24 (one XCD): one row tile
  if (a.xcc && t == 0) {                                         // placement record: which XCD this block really runs on
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    a.xcc[b] = (int)(id & 0xf);
  }
  if (grp >= a.tiles_m) return;
  const bool stamp = a.stamps != nullptr && b == a.stamp_wg && t == 0;
  STCtx c;
  c.wave = wave;
  c.kstep_a = 64;
  c.kstep_b = 64;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sw = ((4 * s + (lane >> 4)) ^ (lane & 7)) * 16;
    c.aoff[s] = (wm * 96 + (lane & 15)) * 128 + sw;
    c.boff[s] = (wn * 48 + (lane & 15)) * 128 + sw;
  }
  float* rowstat = reinterpret_cast<float*>(smem + ST_LDS);
  for (int st = 0; st < a.n; ++st) {
    const GemmArgs& g = a.g[st];
    if (stamp) a.stamps[st * 4 + 0] = wall_clock64();
    if (st > 0) {
      if (t == 0) {
        const unsigned* tk = a.ticket + (st - 1) * CHAIN_TICKETS + grp;
        int spins = 0;
        while (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.target) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 22)) {                            // (~1 s: a member never arrived - give up loudly instead of hanging the queue)
            __hip_atomic_fetch_or(a.ticket + CHAIN_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    if (stamp) a.stamps[st * 4 + 1] = wall_clock64();
    c.A = reinterpret_cast<const bf16_t*>(g.A);
    c.B = reinterpret_cast<const bf16_t*>(g.B);
    const int tn = g.N / ST_T;
    int tc = mem;
    STTile cur, nxt;
    st_tile_setup_rc(cur, g, grp, tc, a.rows_per_tile, wave, lane);
#pragma unroll
    for (int q = 0; q < 3; ++q) { c.oa[q] = cur.oa[q]; c.ob[q] = cur.ob[q]; }
    st_stage_op(c.A, c.oa, 0, smem, wave);
    st_stage_op(c.B, c.ob, 0, smem + ST_OP_BYTES, wave);
    st_stage_op(c.A, c.oa, c.kstep_a, smem + ST_STAGE, wave);
    st_stage_op(c.B, c.ob, c.kstep_b, smem + ST_STAGE + ST_OP_BYTES, wave);
    st_wait_vm<6>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();                 // skew: group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);
    int sc = 0, sn = 2;
    for (;;) {
      const int ntc = tc + 4;
      const bool more = ntc < tn;
      f32x4 acc[6][3];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nk = cur.nk;
      for (int it = 0; it + 2 < nk; ++it) {
        st_tile<0, true, true, 2>(acc, c, smem + sc * ST_STAGE, smem + sn * ST_STAGE, it + 2);
        sc = sc == 2 ? 0 : sc + 1;
        sn = sn == 2 ? 0 : sn + 1;
      }
      if (more) {                                              // the k-tile stream continues with this member's next column tile
        st_tile_setup_rc(nxt, g, grp, ntc, a.rows_per_tile, wave, lane);
#pragma unroll
        for (int q = 0; q < 3; ++q) { c.oa[q] = nxt.oa[q]; c.ob[q] = nxt.ob[q]; }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          st_tile<0, true, true, 2>(acc, c, smem + sc * ST_STAGE, smem + sn * ST_STAGE, e);
          sc = sc == 2 ? 0 : sc + 1;
          sn = sn == 2 ? 0 : sn + 1;
        }
      } else {
        st_tile<1, true, true, 2>(acc, c, smem + sc * ST_STAGE, nullptr, 0);
        sc = sc == 2 ? 0 : sc + 1;
        st_tile<2, true, true, 2>(acc, c, smem + sc * ST_STAGE, nullptr, 0);
      }
      const int s_free = more ? (sc == 0 ? 2 : sc - 1) : sc;
      char* scratch = smem + s_free * ST_STAGE + wm * (ST_STAGE / 2);
      if (g.epi & EPI_ROWSTAT) st_epilogue_lds<ST_AUX_RES, float, false, 2>(acc, g, cur, wm, wn, lane, wave, scratch, rowstat);
      else if (g.epi & EPI_LNFOLD) st_epilogue_lds<ST_AUX_NONE, bf16_t, false, 1>(acc, g, cur, wm, wn, lane, wave, scratch, rowstat);
      else st_epilogue_lds<ST_AUX_NONE, bf16_t, false, 0>(acc, g, cur, wm, wn, lane, wave, scratch, rowstat);
      if (!more) break;
      cur = nxt;
      tc = ntc;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                 // un-skew: both wave groups have passed the same number of barriers
    if (stamp) a.stamps[st * 4 + 2] = wall_clock64();
    // publish this member's part of the stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave: its stores have been written
    __syncthreads();
    if (t == 0) {
      if ((a.flags & 1) && st + 1 < a.n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // placement-independent form: write back the XCD L2's dirty lines
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // EVERY stage's ticket advances by one per member and launch (also the last stage's and those of unused stages below), so that
      // `target = 4 x launches` holds for every stage whatever the stage count of the earlier launches was
      __hip_atomic_fetch_add(a.ticket + st * CHAIN_TICKETS + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp) a.stamps[st * 4 + 3] = wall_clock64();
  }
  if (t == 0)
    for (int st = a.n; st < 4; ++st) __hip_atomic_fetch_add(a.ticket + st * CHAIN_TICKETS + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int rmcl_launch_gemm_chain(const ChainArgs& a0, hipStream_t s) {
  ChainArgs a = a0;
  RMCL_REQUIRE(a.n >= 1 && a.n <= 4 && a.ticket, "gemm_chain: 1..4 stages and a ticket buffer");
  const int M = a.g[0].M;
  a.tiles_m = cdiv(M, ST_T);
  a.rows_per_tile = cdiv(M, a.tiles_m);
  RMCL_REQUIRE(a.tiles_m <= CHAIN_TICKETS, "gemm_chain: at most 64 row tiles (M <= 12288)");
  for (int i = 0; i < a.n; ++i) {
    const GemmArgs& g = a.g[i];
    RMCL_REQUIRE(g.A && g.B && g.C && g.M == M && g.N % ST_T == 0 && g.K % 64 == 0 && g.K >= 128, "gemm_chain: bad stage shape");
    RMCL_REQUIRE((long)g.M * g.lda < (1L << 31) && (long)g.N * g.ldb < (1L << 31), "gemm_chain: operand too large for 32-bit offsets");
    RMCL_REQUIRE(!(g.epi & ~(EPI_BIAS | EPI_GELU | EPI_SAVE_PREACT | EPI_RESIDUAL | EPI_ROWSTAT | EPI_LNFOLD)), "gemm_chain: unsupported epilogue");
    RMCL_REQUIRE(!(g.epi & EPI_ROWSTAT) || ((g.epi & EPI_RESIDUAL) && g.aux && g.C2 && g.ln_part && g.ln_nparts == 4 * (g.N / ST_T)), "gemm_chain: producer stage");
    RMCL_REQUIRE(!(g.epi & EPI_RESIDUAL) || (g.epi & EPI_ROWSTAT), "gemm_chain: a residual stage is a row-statistics producer (fp32 output)");
    RMCL_REQUIRE(!(g.epi & EPI_LNFOLD) || (g.ln_s && g.ln_c && g.ln_part && g.ln_nparts > 0 && g.ln_nparts % 2 == 0 && g.ln_cols > 0), "gemm_chain: folded stage");
  }
  constexpr int LDS = ST_LDS + 2048;
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(gemm_chain_kernel), LDS));
  const int grid = 32 * cdiv(a.tiles_m, 8);
  RMCL_LAUNCH(gemm_chain_kernel, dim3(grid), dim3(512), LDS, s, a);
  RMCL_CHECK_LAUNCH();
  return 0;
}
