// bf16 MFMA GEMM with 192x384x64 block tiles ("wide" sibling of gemm_st.hip) for the N = 3072 activation GEMMs of the
// step (fc1 forward, fc2 dX): the A tile is shared by twice as many columns, so the LDS-DMA moves 7.6 instead of 10.2 bytes
// per KFLOP (the 192x192 loop is bound by that path) and half as many tile prologues / epilogues run per FLOP.
// M = 64*185 -> 62 x 8 = 496 tiles = exactly 2 rounds of a 248-workgroup grid.
//
//   * one 8-wave workgroup per CU, per-wave output 96x96 (36 fragments = 144 accumulator registers); waves 0-3 / 4-7 are
//     the two ping-pong groups (rows 0..95 / 96..191), wave column wn owns columns {h*192 + wn*48 + 0..47 : h = 0,1}.
//   * LDS: 2 buffers x {A 192x64, B0 192x64, B1 192x64} bf16 (24 KiB each, images and swizzles of gemm_st.hip) = 144 KiB.
//   * a k-tile is 4 phases of 18 MFMA (3 row fragments x 3 column fragments x 2 k-steps):
//        P1: rows-lo x B0   (reads A-lo 6, B0 6)   stages A(t+1)
//        P2: rows-lo x B1   (reads B1 6)           stages B0(t+1)
//        P3: rows-hi x B1   (reads A-hi 6)         stages B1(t+1)
//        P4: rows-hi x B0   (reads B0 6)           -
//     every unit is staged >= 2 phases after the last read of the buffer it overwrites and waited for (counted vmcnt(3):
//     only the youngest unit may be in flight) one phase before its first read.
#include <type_traits>
#include "rmcl_common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define SW_UNIT 24576                        // 192 rows x 64 bf16
#define SW_BUF (3 * SW_UNIT)                 // A, B0, B1
#define SW_LDS (2 * SW_BUF)                  // 144 KiB


// In-kernel phase trace (developer builds: RMCL_EXTRA_FLAGS=-DST_TRACE, tools/st_trace.py), as in gemm_st.hip
#ifdef ST_TRACE
__device__ long long g_sw_trace[2][32];
#define SW_STAMP(i)                                                                                              \
  if (blockIdx.x == 100 && (threadIdx.x & 255) == 0) g_sw_trace[threadIdx.x >> 8][i] = wall_clock64()
extern "C" int rmcl_debug_sw_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sw_trace), sizeof(long long) * 64);
}
#else
#define SW_STAMP(i)
#endif

template <int N>
__device__ __forceinline__ void sw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void sw_stage(const bf16_t* base, const uint32_t (&off)[3], long koff, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 3; ++q)
    __builtin_amdgcn_global_load_lds((glb_void*)(base + koff + off[q]), (lds_void*)(lds_unit + (wave * 3 + q) * 1024), 16, 0, 0);
}

__device__ __forceinline__ int sw_gk(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

template <bool KC>
__device__ __forceinline__ bf16x8 sw_ld(const char* unit, int off, int s) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(unit + off);          // (off already holds the k-step swizzle)
  } else {
    union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(unit + off + s * (32 * 384) + h * (4 * 384)));
    return u.v;
  }
}

#define SW_PHASE_BEGIN()                                  \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1)
#define SW_PHASE_END()                                    \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0)

enum { SW_AUX_NONE = 0, SW_AUX_RES = 1, SW_AUX_DGELU = 2 };

struct SWCtx {
  const bf16_t* A;
  const bf16_t* B;
  uint32_t oa[3], ob0[3], ob1[3];
  int aoff[2], boff[2], boffj[3];
  long kstep_b;
  int wave;
};

// GELU'(aux) epilogue operand through LDS: one 24 KiB unit (32 image rows x 768 B) of the [96][768 B] image of an aux chunk
// (48 tile rows per wave group: image row r -> tile row (r / 48) * 96 + ch * 48 + r % 48), whole rows, 16 bytes per lane;
// the 16-byte chunk of row r at chunk ^ (r & 7) like the output image (the swizzle is applied to the SOURCE address).
struct SWAux {
  const bf16_t* aux;
  long ld;
  int m0, n0, M;
};
__device__ __forceinline__ void sw_stage_aux(const SWAux& x, int ch, int unit, char* area, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int L = unit * SW_UNIT + (wave * 3 + q) * 1024 + lane * 16;
    const int r = L / 768, p = (L - r * 768) >> 4;
    const int trow = (r / 48) * 96 + ch * 48 + (r % 48);
    const long m = min(x.m0 + trow, x.M - 1);
    const bf16_t* src = x.aux + m * x.ld + x.n0 + ((p ^ (r & 7)) << 3);
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(area + unit * SW_UNIT + (wave * 3 + q) * 1024), 16, 0, 0);
  }
}

// one k-tile; LAST: no staging (the last k-tile of the output tile).  AUXST (with LAST): the buffer that would take tile
// t+1 takes chunk 0 of the epilogue's aux operand instead, one unit per phase like the operand staging it replaces.
template <bool B_KC, bool LAST, bool AUXST = false>
__device__ __forceinline__ void sw_ktile(f32x4 (&acc)[6][6], const SWCtx& c, const char* cur, char* oth, int t1, const SWAux* ax = nullptr, int lane = 0) {
  bf16x8 a[3][2], b[3][2];
  // ---- P1: rows-lo x B0
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) b[j][s] = B_KC ? sw_ld<true>(cur + SW_UNIT, j * 2048 + c.boff[s], s) : sw_ld<false>(cur + SW_UNIT, c.boffj[j], s);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[i][s] = sw_ld<true>(cur, i * 2048 + c.aoff[s], s);
  if (!LAST) { sw_stage(c.A, c.oa, (long)t1 * 64, oth, c.wave); sw_wait_vm<3>(); } else { sw_wait_vm<0>(); if (AUXST) sw_stage_aux(*ax, 0, 0, oth, c.wave, lane); }
  SW_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[i][j], 0, 0, 0);
  SW_PHASE_END();
  // ---- P2: rows-lo x B1
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) b[j][s] = B_KC ? sw_ld<true>(cur + 2 * SW_UNIT, j * 2048 + c.boff[s], s) : sw_ld<false>(cur + 2 * SW_UNIT, c.boffj[j], s);
  if (!LAST) sw_stage(c.B, c.ob0, (long)t1 * c.kstep_b, oth + SW_UNIT, c.wave);
  else if (AUXST) sw_stage_aux(*ax, 0, 1, oth, c.wave, lane);
  SW_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[i][3 + j], 0, 0, 0);
  SW_PHASE_END();
  // ---- P3: rows-hi x B1
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[i][s] = sw_ld<true>(cur, (3 + i) * 2048 + c.aoff[s], s);
  if (!LAST) sw_stage(c.B, c.ob1, (long)t1 * c.kstep_b, oth + 2 * SW_UNIT, c.wave);
  else if (AUXST) sw_stage_aux(*ax, 0, 2, oth, c.wave, lane);
  SW_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[3 + i][3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[3 + i][3 + j], 0, 0, 0);
  SW_PHASE_END();
  // ---- P4: rows-hi x B0
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) b[j][s] = B_KC ? sw_ld<true>(cur + SW_UNIT, j * 2048 + c.boff[s], s) : sw_ld<false>(cur + SW_UNIT, c.boffj[j], s);
  if (!LAST) sw_wait_vm<3>();                                 // A(t+1), B0(t+1) have landed; B1(t+1) may be in flight
  SW_PHASE_BEGIN();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[3 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[3 + i][j], 0, 0, 0);
  SW_PHASE_END();
}

template <typename TO>
__device__ __forceinline__ void sw_store4(TO* p, const float (&v)[4]) {
  if constexpr (sizeof(TO) == 2) {
    uint2 pk;
    pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = pk;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// LNF = 1: LayerNorm folded into this GEMM (EPI_LNFOLD, gemm.h): A = bf16 copy of the residual stream, B = W * gamma;
// v = rstd_m * (acc - mean_m * s_n) + c_n with the row statistics summed from the producer's partials into LDS.
// DROP (bf16 outputs only): the dropout epilogues of the training-realistic configuration (drop_rate 0.1, config.py:57) as separate
// instantiations - EPI_DROPOUT on fc1's GELU output, EPI_DROP_BWD (that mask again) on fc2-dX - so that the dropout-free kernels carry
// none of their state; masks are counter-based (rmcl_common.h), indexed by the dense element index like every other kernel's.
template <bool B_KC, int AUX, typename TO, int LNF = 0, bool DROP = false>
__global__ __launch_bounds__(512) void gemm_sw_kernel(GemmArgs g, int tiles_m, int tiles_n, int rows_per_tile) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  SW_STAMP(0);
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // bands of 4 column tiles, row tiles inside a band, the band's column tiles innermost (as gemm_st.hip)
  int tr, tc;
  if (tiles_n % 4 == 0) {
    const int band = bid / (tiles_m * 4), rem = bid - band * (tiles_m * 4);
    tr = rem >> 2;
    tc = band * 4 + (rem & 3);
  } else {
    tr = bid / tiles_n;
    tc = bid - tr * tiles_n;
  }
  const int m0 = tr * rows_per_tile, n0 = tc * 384;
  const int m_end = min(g.M, m0 + rows_per_tile);
  const int nk = g.K / 64;

  SWCtx c;
  c.A = reinterpret_cast<const bf16_t*>(g.A);
  c.B = reinterpret_cast<const bf16_t*>(g.B);
  c.wave = wave;
  c.kstep_b = B_KC ? 64 : 64 * g.ldb;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = (wave * 3 + q) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const int lin = (wave * 3 + q) * 64 + lane;              // 16-byte chunk index in the [64][24] image of a [K][cols] operand
    const int k = lin / 24, c16 = (lin - k * 24) ^ (sw_gk(k) << 1);
    c.oa[q] = (uint32_t)min(m0 + row, g.M - 1) * (uint32_t)g.lda + chunk * 8;
    if (B_KC) {
      c.ob0[q] = (uint32_t)(n0 + row) * (uint32_t)g.ldb + chunk * 8;
      c.ob1[q] = (uint32_t)(n0 + 192 + row) * (uint32_t)g.ldb + chunk * 8;
    } else {
      c.ob0[q] = (uint32_t)k * (uint32_t)g.ldb + n0 + c16 * 8;
      c.ob1[q] = c.ob0[q] + 192;
    }
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
    const int k = 8 * (lane >> 4) + q;
    const int c8 = (wn * 48 + j * 16) / 4 + p;
    c.boffj[j] = k * 384 + (((c8 >> 1) ^ (sw_gk(k) << 1)) * 16) + (c8 & 1) * 8;
  }

  f32x4 acc[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  char* buf0 = smem;
  char* buf1 = smem + SW_BUF;
  // LayerNorm-folded consumer: the row statistics of the tile's 192 rows come from the PRODUCER's partial sums, which are complete
  // when this kernel starts - fetched here, under the latency of the first k-tile's LDS-DMA, instead of at the head of the
  // epilogue (tools/st_trace.py: chunk 0 of the epilogue 5.7 us against 3.4 us for chunk 1, the difference being this round trip)
  float4 pst[LNF == 1 ? 8 : 1];
  float cen = 0.f;                                              // centre of the row's partial sums (gemm.h ln_center): only the stashed mean needs it
  if constexpr (LNF == 1) {
    if (t < 192) {
      if (n0 == 0 && g.ln_mean && g.ln_center) cen = g.ln_center[min((long)m0 + t, (long)g.M - 1)];
      const float4* pp = reinterpret_cast<const float4*>(g.ln_part + min((long)m0 + t, (long)g.M - 1) * (long)(g.ln_nparts * 2));
      if (g.ln_nparts == 16) {
#pragma unroll
        for (int q = 0; q < 8; ++q) pst[q] = pp[q];
      }
    }
  }
  sw_stage(c.A, c.oa, 0, buf0, wave);
  sw_stage(c.B, c.ob0, 0, buf0 + SW_UNIT, wave);
  sw_stage(c.B, c.ob1, 0, buf0 + 2 * SW_UNIT, wave);
  if constexpr (LNF == 1) {
    if (t < 192) {
      const long m = min((long)m0 + t, (long)g.M - 1);
      float s1 = 0.f, s2 = 0.f;
      if (g.ln_nparts == 16) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1 += pst[q].x + pst[q].z; s2 += pst[q].y + pst[q].w; }
      } else {
        const float4* pp = reinterpret_cast<const float4*>(g.ln_part + m * (long)(g.ln_nparts * 2));
        for (int q = 0; q < g.ln_nparts / 2; ++q) { const float4 v = pp[q]; s1 += v.x + v.z; s2 += v.y + v.w; }
      }
      const float inv = 1.0f / (float)g.ln_cols, mean = s1 * inv;
      const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + g.ln_eps);
      float* rowstat0 = reinterpret_cast<float*>(smem + SW_LDS);
      rowstat0[2 * t] = mean;
      rowstat0[2 * t + 1] = rstd;
      if (n0 == 0 && g.ln_mean && m0 + t < m_end) { g.ln_mean[m] = mean + cen; g.ln_rstd[m] = rstd; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (published by the barrier below; first read in the epilogue)
  }
  sw_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wm == 1) __builtin_amdgcn_s_barrier();                 // skew: group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);

  SW_STAMP(1);
  int it = 0;
  for (; it + 1 < nk; ++it) {
    char* cur = (it & 1) ? buf1 : buf0;
    char* oth = (it & 1) ? buf0 : buf1;
    sw_ktile<B_KC, false>(acc, c, cur, oth, it + 1);
  }
  // GELU'(aux) epilogue with bf16 output: the aux tile (the stashed pre-activation, cold in HBM by the time the backward
  // runs) comes through LDS - chunk 0 is fetched by LDS-DMA into the idle buffer DURING the last k-tile, chunk 1 while chunk
  // 0 is converted and stored; whole 768-byte rows instead of the fragment layout's 32-byte pieces.
  constexpr bool AUX_LDS = AUX == SW_AUX_DGELU && sizeof(TO) == 2;
  char* last_cur = (it & 1) ? buf1 : buf0;
  char* aux_area = (it & 1) ? buf0 : buf1;                     // free during the last k-tile
  SWAux ax{reinterpret_cast<const bf16_t*>(g.aux), (long)g.ld_aux, m0, n0, g.M};
  if constexpr (AUX_LDS) {
    sw_ktile<B_KC, true, true>(acc, c, last_cur, aux_area, 0, &ax, lane);
    sw_wait_vm<0>();
  } else {
    sw_ktile<B_KC, true>(acc, c, last_cur, nullptr, 0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if constexpr (AUX_LDS) {
    __builtin_amdgcn_s_barrier();                              // every wave's aux DMA has landed, every wave is done with the last k-tile
    // chunk 1 of the aux operand goes into the buffer the last k-tile was read from, NOW: each chunk's output image is written IN PLACE
    // over its aux image (a lane reads exactly the 8-byte units it writes), so chunk 0 lives in one operand buffer and chunk 1 in the
    // other and this DMA has the whole of chunk 0 to land (fetched after chunk 0's barrier it cost ~4 us of HBM latency per tile:
    // tools/st_trace.py fc2dx, "chunk0: barrier + read-back + stores issued" 7.4 us against 3.3 for chunk 1)
#pragma unroll
    for (int u = 0; u < 3; ++u) sw_stage_aux(ax, 1, u, last_cur, wave, lane);
  }

  SW_STAMP(2);
  // epilogue (order as gemm_st.hip: alpha, bias, gelu'(aux), stash, gelu, residual)
  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C);
  TO* C2 = reinterpret_cast<TO*>(g.C2);
  const int mb = m0 + wm * 96 + (lane & 15);
  if constexpr (sizeof(TO) == 2) {
    // bf16 outputs go through LDS (all 144 KiB are free now) and leave as whole 768-byte rows, 16 bytes per lane: the
    // fragment layout's 32-byte row segments are written at about half that rate (tools/store_pattern_bench.hip).  Per wave
    // group: 48 rows x 768 B images of the output and, when stashed, of the pre-activation; 16-byte chunk c of row r at
    // c ^ (r & 7).  (Both groups are barrier-aligned here; every wave executes the same 4 barriers.)
    constexpr int ROWB = 768, PIECES = 48, ROWS = 48, IMG = ROWS * ROWB;   // 36 KiB per image
    // (AUX_LDS: no stash image exists - chunk ch's aux image AND, in place, its output image live in one operand buffer each:
    // chunk 0 in the one that was idle during the last k-tile, chunk 1 in the one the last k-tile was read from)
    char* img0 = AUX_LDS ? aux_area + wm * IMG : smem + wm * (2 * IMG);
    const bool stash = !AUX_LDS && (epi & EPI_SAVE_PREACT) != 0;
    float* rowstat = reinterpret_cast<float*>(smem + SW_LDS);  // LNF: 192 x (mean, rstd), beyond the two operand buffers (filled at kernel start)
    // per-column vectors of both column halves and (LNF) the statistics of this lane's six rows: fetched ONCE, ahead of the chunk
    // loop (inside it every (chunk, half) paid an L2 / LDS round trip before its first FMA)
    // (the GELU' form carries no bias - fc2-dX - and has no registers to spare: its zero "bias" stays a constant)
    // (the dropout instantiations keep them per (chunk, half) too: with both halves' vectors live beside the mask state the folded fc1
    // form spilled - 256 VGPRs + scratch - and its epilogue gained nothing from the round-4 rework: 90 us against 76 without dropout)
    constexpr bool HOIST = AUX != SW_AUX_DGELU && !DROP;
    float4 bias2[HOIST ? 2 : 1][3], lns2[HOIST ? 2 : 1][LNF == 1 ? 3 : 1];
    if constexpr (HOIST) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int nbh = n0 + hb * 192 + wn * 48 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (LNF == 1) {
            bias2[hb][j] = *reinterpret_cast<const float4*>(g.ln_c + nbh + j * 16);
            lns2[hb][j] = *reinterpret_cast<const float4*>(g.ln_s + nbh + j * 16);
          } else {
            bias2[hb][j] = (epi & EPI_BIAS) ? *reinterpret_cast<const float4*>(g.bias + nbh + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
    float2 rs6[LNF == 1 ? 6 : 1];
    if constexpr (LNF == 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) rs6[i] = *reinterpret_cast<const float2*>(rowstat + 2 * (wm * 96 + i * 16 + (lane & 15)));
    }
    // Addresses of the epilogue, once per tile (they were recomputed per value group / per store: with the integer divisions and 64-bit
    // pointer arithmetic about a third of the epilogue's VALU instructions, MFMAs idle).  Image writes: row il*16 + lane%16 has the same
    // (row & 7) for every il, so one offset per (column half, j) and the row block as an immediate.  Read-back + row stores: work item
    // q = tg + 256 k -> (row, piece) = (q / 48, q % 48); 256 = 5 * 48 + 16, so k and k + 3 differ by exactly 16 rows: three (row, piece) pairs.
    int woff[2][3];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r = lane & 15, e8 = hb * 48 + wn * 12 + j * 4 + (lane >> 4);     // 8-byte unit (4 bf16) in the 384-column row
        woff[hb][j] = r * ROWB + (((e8 >> 1) ^ (r & 7)) * 16) + (e8 & 1) * 8;
        asm volatile("" : "+v"(woff[hb][j]));
      }
    int rb_row[3], rb_lds[3];
    uint32_t rb_dst[3];
    {
      const int tg = (wave & 3) * 64 + lane, row0 = tg / PIECES, cp0 = tg - row0 * PIECES;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int t16 = cp0 + 16 * kk, wrap = t16 >= PIECES ? 1 : 0, cp = t16 - wrap * PIECES, row = row0 + 5 * kk + wrap;
        rb_row[kk] = row;
        rb_lds[kk] = row * ROWB + cp * 16;
        rb_dst[kk] = (uint32_t)row * (uint32_t)g.ldc + (uint32_t)((cp ^ (row & 7)) * 8);
        asm volatile("" : "+v"(rb_row[kk]), "+v"(rb_lds[kk]), "+v"(rb_dst[kk]));
      }
    }
    // The chunk loop, once per (GELU, stash) combination and chosen here: tested per value group inside the loops the two flags cost a scalar
    // branch + mask set-up per four values and cut the loop body into basic blocks that each re-materialised the polynomial's constants
    // (~600 scalar instructions per tile, which a wave that has its SIMD to itself issues one at a time like the VALU ones).
    // (dropout instantiations: which mask applies - 1 the output's, 2 the forward's hidden one on the way back, 3 decided per group at run
    // time - is a third parameter for the step's own combinations; the element index of a group is a 32-bit add to a per-lane base)
    const uint32_t dbase_aux = DROP ? (uint32_t)mb * (uint32_t)g.ld_aux : 0u, dbase_c = DROP ? (uint32_t)mb * (uint32_t)g.ldc : 0u;
    auto chunk_loop = [&](auto gelu_c, auto stash_c, auto dm_c) {
      constexpr bool GELU = decltype(gelu_c)::value, STASH = decltype(stash_c)::value;
      constexpr int DM = decltype(dm_c)::value;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        char* aux_ch = ch == 0 ? aux_area : last_cur;           // (AUX_LDS only)
        char* img = AUX_LDS ? aux_ch + wm * IMG : img0;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int nb = n0 + hb * 192 + wn * 48 + 4 * (lane >> 4);
          float4 bias_l[3], lns_l[LNF == 1 ? 3 : 1];
          if constexpr (!HOIST) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if constexpr (LNF == 1) {
                bias_l[j] = *reinterpret_cast<const float4*>(g.ln_c + nb + j * 16);
                lns_l[j] = *reinterpret_cast<const float4*>(g.ln_s + nb + j * 16);
              } else {
                bias_l[j] = (epi & EPI_BIAS) ? *reinterpret_cast<const float4*>(g.bias + nb + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
              }
            }
          }
          auto& bias = HOIST ? bias2[HOIST ? hb : 0] : bias_l;
          auto& lns = HOIST ? lns2[HOIST ? hb : 0] : lns_l;
          uint2 pre[AUX == SW_AUX_DGELU ? 3 : 1][3];
#pragma unroll
          for (int il = 0; il < 3; ++il) {
            const long mr = min(mb + (ch * 3 + il) * 16, g.M - 1);
            const int arow = wm * 48 + il * 16 + (lane & 15);      // row of the aux image
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if constexpr (AUX_LDS) {
                const int e8 = hb * 48 + wn * 12 + j * 4 + (lane >> 4);
                pre[il][j] = *reinterpret_cast<const uint2*>(aux_ch + arow * ROWB + (((e8 >> 1) ^ (arow & 7)) * 16) + (e8 & 1) * 8);
              } else if (AUX == SW_AUX_DGELU) {
                pre[il][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + mr * g.ld_aux + nb + j * 16);
              }
            }
          }
#pragma unroll
          for (int il = 0; il < 3; ++il) {
            const int i = ch * 3 + il;
            float mean_m = 0.f, rstd_m = 1.f;
            if constexpr (LNF == 1) { mean_m = rs6[i].x; rstd_m = rs6[i].y; }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              // (all four values of the accumulator at once: rmcl_common.h, "Four-wide forms")
              const f32x4 av = acc[i][hb * 3 + j], bv = f4v(bias[j]);
              f32x4 v;
              if constexpr (LNF == 1) v = rstd_m * (av - mean_m * f4v(lns[j])) + bv;
              else v = g.alpha * av + bv;
              if (AUX == SW_AUX_DGELU) v *= gelu_poly_grad4(bf2f4(pre[il][j]));
              if constexpr (DROP && (DM & 2) != 0) {
                if (DM == 2 || (epi & EPI_DROP_BWD)) {                 // mask of the forward's hidden dropout, indexed like the stash
                  const uint32_t di = dbase_aux + (uint32_t)(i * 16) * (uint32_t)g.ld_aux + (uint32_t)(nb + j * 16);
                  drop_scale4(g.drop_seed, di, g.drop_thresh, g.drop_inv_keep, v);
                }
              }
              const int off = woff[hb][j] + il * 16 * ROWB;
              if constexpr (STASH) *reinterpret_cast<uint2*>(img + IMG + off) = f2bf4(v);
              if constexpr (GELU) v = gelu_poly4(v);
              if constexpr (DROP && (DM & 1) != 0) {
                if (DM == 1 || (epi & EPI_DROPOUT)) {
                  const uint32_t ci = dbase_c + (uint32_t)(i * 16) * (uint32_t)g.ldc + (uint32_t)(nb + j * 16);
                  drop_scale4(g.drop_seed, ci, g.drop_thresh, g.drop_inv_keep, v);
                }
              }
              const uint2 pk = f2bf4(v);
              *reinterpret_cast<uint2*>(img + off) = pk;
            }
          }
        }
        SW_STAMP(4 + 3 * ch);
        __builtin_amdgcn_s_barrier();                          // the group's images of this chunk are complete
        {
          const int mrow = m0 + wm * 96 + ch * ROWS, lim = m_end - mrow;          // (uniform) first row of this group's chunk, live rows in it
          const uint32_t cbase = (uint32_t)mrow * (uint32_t)g.ldc + (uint32_t)n0;
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
              if (rb_row[kk] + 16 * t3 < lim) {
                const size_t dst = (size_t)(cbase + (uint32_t)(16 * t3) * (uint32_t)g.ldc + rb_dst[kk]);
                *reinterpret_cast<float4*>(C + dst) = *reinterpret_cast<const float4*>(img + rb_lds[kk] + t3 * 16 * ROWB);
                if constexpr (STASH) *reinterpret_cast<float4*>(C2 + dst) = *reinterpret_cast<const float4*>(img + IMG + rb_lds[kk] + t3 * 16 * ROWB);
              }
            }
        }
        if constexpr (AUX_LDS) {
          // chunk 1's aux DMA (3 instructions per wave) was issued BEFORE this chunk's 9 row stores and vmcnt retires in issue order:
          // vmcnt(9) proves the DMA without waiting for the stores to drain (tools/st_trace.py: 9.3 us here with vmcnt(0)).  Only
          // when every row of chunk 0 is live - otherwise a wave may skip store instructions and the count would not hold.
          if (ch == 0) { if (m0 + 144 <= m_end) sw_wait_vm<9>(); else sw_wait_vm<0>(); }
        }
        SW_STAMP(5 + 3 * ch);
        __builtin_amdgcn_s_barrier();                          // images consumed (AUX_LDS: and chunk 1's aux has landed)
        SW_STAMP(6 + 3 * ch);
      }
    };
    {
      using T1 = std::integral_constant<bool, true>;
      using T0 = std::integral_constant<bool, false>;
      using D0 = std::integral_constant<int, 0>;
      using D1 = std::integral_constant<int, 1>;
      using D2 = std::integral_constant<int, 2>;
      using D3 = std::integral_constant<int, 3>;
      const bool gelu = (epi & EPI_GELU) != 0;
      const int dm = DROP ? (((epi & EPI_DROPOUT) ? 1 : 0) | ((epi & EPI_DROP_BWD) ? 2 : 0)) : 0;
      if constexpr (!DROP) {
        if constexpr (AUX_LDS) {
          if (gelu) chunk_loop(T1{}, T0{}, D0{}); else chunk_loop(T0{}, T0{}, D0{});
        } else {
          if (gelu && stash) chunk_loop(T1{}, T1{}, D0{});
          else if (gelu) chunk_loop(T1{}, T0{}, D0{});
          else if (stash) chunk_loop(T0{}, T1{}, D0{});
          else chunk_loop(T0{}, T0{}, D0{});
        }
      } else if constexpr (AUX_LDS) {                            // fc2-dX under dropout: the hidden mask on the way back
        if (!gelu && dm == 2) chunk_loop(T0{}, T0{}, D2{});
        else if (gelu) chunk_loop(T1{}, T0{}, D3{});
        else chunk_loop(T0{}, T0{}, D3{});
      } else {                                                   // fc1 under dropout: GELU, then the hidden mask (with / without the stash)
        if (gelu && stash && dm == 1) chunk_loop(T1{}, T1{}, D1{});
        else if (gelu && !stash && dm == 1) chunk_loop(T1{}, T0{}, D1{});
        else if (gelu && stash) chunk_loop(T1{}, T1{}, D3{});
        else if (gelu) chunk_loop(T1{}, T0{}, D3{});
        else if (stash) chunk_loop(T0{}, T1{}, D3{});
        else chunk_loop(T0{}, T0{}, D3{});
      }
    }
  } else {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int nb = n0 + hb * 192 + wn * 48 + 4 * (lane >> 4);
      float4 bias[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) bias[j] = (epi & EPI_BIAS) ? *reinterpret_cast<const float4*>(g.bias + nb + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      uint2 pre[AUX == SW_AUX_DGELU ? 6 : 1][3];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const long mr = min(mb + i * 16, g.M - 1);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (AUX == SW_AUX_DGELU) pre[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + mr * g.ld_aux + nb + j * 16);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int m = mb + i * 16;
        const bool live = m < m_end;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const f32x4 av = acc[i][hb * 3 + j];
          float v[4] = {g.alpha * av[0] + bias[j].x, g.alpha * av[1] + bias[j].y, g.alpha * av[2] + bias[j].z, g.alpha * av[3] + bias[j].w};
          if (AUX == SW_AUX_DGELU) {
            const uint2 u = pre[i][j];
            v[0] *= gelu_poly_grad(__uint_as_float(u.x << 16)); v[1] *= gelu_poly_grad(__uint_as_float(u.x & 0xffff0000u));
            v[2] *= gelu_poly_grad(__uint_as_float(u.y << 16)); v[3] *= gelu_poly_grad(__uint_as_float(u.y & 0xffff0000u));
          }
          const long ci = (long)m * g.ldc + nb + j * 16;
          if ((epi & EPI_SAVE_PREACT) && live) sw_store4<TO>(C2 + ci, v);
          if (epi & EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_poly(v[r]);
          }
          if (live) sw_store4<TO>(C + ci, v);
        }
      }
    }
  }
#ifdef ST_TRACE
  SW_STAMP(10);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SW_STAMP(11);
#endif
}

bool rmcl_gemm_sw_supported(const GemmArgs& g, int a_kc, int b_kc) {
  if (!a_kc || g.nb1 > 1 || g.nb2 > 1 || g.splitk > 1) return false;
  if (g.N % 384 != 0 || g.K % 64 != 0 || g.K < 128) return false;
  if ((long)g.M * g.lda >= (1L << 31) || (long)(b_kc ? g.N : g.K) * g.ldb >= (1L << 31)) return false;
  if ((long)g.M * g.ldc >= (1L << 31)) return false;                          // (the bf16 epilogue's row stores use 32-bit element offsets)
  if (g.epi & ~(EPI_BIAS | EPI_GELU | EPI_SAVE_PREACT | EPI_DGELU | EPI_LNFOLD | EPI_DROPOUT | EPI_DROP_BWD)) return false;   // (residual epilogue: 72 more live registers spill; N = 768 anyway)
  if ((g.epi & EPI_DROP_BWD) && !(g.epi & EPI_DGELU)) return false;          // (dropout epilogues: bf16 outputs only - the router checks dt_out)
  if (g.epi & EPI_LNFOLD) {
    if (!b_kc || (g.epi & (EPI_BIAS | EPI_DGELU)) || !g.ln_s || !g.ln_c || !g.ln_part || g.ln_nparts % 2 || g.ln_nparts <= 0 || g.ln_cols <= 0) return false;
  }
  return true;
}

// fraction of the `cus`-workgroup rounds that carry a tile
double rmcl_gemm_sw_fill(const GemmArgs& g, int cus) {
  const long tiles = (long)cdiv(g.M, 192) * (g.N / 384);
  return (double)tiles / (double)(cdiv(tiles, (long)cus) * cus);
}

template <bool B_KC, int AUX, typename TO, bool DROP = false>
static int launch_sw3(const GemmArgs& g, hipStream_t s) {
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(gemm_sw_kernel<B_KC, AUX, TO, 0, DROP>), SW_LDS));
  const int tm = cdiv(g.M, 192), tn = g.N / 384, rows = cdiv(g.M, tm);
  RMCL_LAUNCH((gemm_sw_kernel<B_KC, AUX, TO, 0, DROP>), dim3(tm * tn), dim3(512), SW_LDS, s, g, tm, tn, rows);
  RMCL_CHECK_LAUNCH();
  return 0;
}

template <bool B_KC, int AUX>
static int launch_sw2(const GemmArgs& g, int dt_out, hipStream_t s) {
  if (g.epi & (EPI_DROPOUT | EPI_DROP_BWD)) {
    RMCL_REQUIRE(dt_out == RMCL_BF16, "gemm_sw: the dropout epilogues write bf16");
    return launch_sw3<B_KC, AUX, bf16_t, true>(g, s);
  }
  return dt_out == RMCL_F32 ? launch_sw3<B_KC, AUX, float>(g, s) : launch_sw3<B_KC, AUX, bf16_t>(g, s);
}

template <bool B_KC>
static int launch_sw(const GemmArgs& g, int dt_out, hipStream_t s) {
  if (g.epi & EPI_DGELU) return launch_sw2<B_KC, SW_AUX_DGELU>(g, dt_out, s);
  return launch_sw2<B_KC, SW_AUX_NONE>(g, dt_out, s);
}

template <bool DROP>
static int launch_sw_lnf(const GemmArgs& g, hipStream_t s) {
  constexpr int LDS = SW_LDS + 2048;
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>((gemm_sw_kernel<true, SW_AUX_NONE, bf16_t, 1, DROP>)), LDS));
  const int tm = cdiv(g.M, 192), tn = g.N / 384, rows = cdiv(g.M, tm);
  RMCL_LAUNCH((gemm_sw_kernel<true, SW_AUX_NONE, bf16_t, 1, DROP>), dim3(tm * tn), dim3(512), LDS, s, g, tm, tn, rows);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_launch_gemm_sw(const GemmArgs& g, int dt_out, int b_kc, hipStream_t s) {
  if (g.epi & EPI_LNFOLD) {
    RMCL_REQUIRE(b_kc && dt_out == RMCL_BF16, "gemm_sw: the LayerNorm-folded form is [rows][K] x [cols][K] with bf16 output");
    return (g.epi & EPI_DROPOUT) ? launch_sw_lnf<true>(g, s) : launch_sw_lnf<false>(g, s);
  }
  return b_kc ? launch_sw<true>(g, dt_out, s) : launch_sw<false>(g, dt_out, s);
}
