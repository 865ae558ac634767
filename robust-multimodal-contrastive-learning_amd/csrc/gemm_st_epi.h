// Shared pieces of the 192-row tile GEMM kernels (gemm_st.hip: one 8-wave workgroup per CU; gemm_dp.hip: two 4-wave
// workgroups per CU): tile descriptor, XCD-aware tile order, and the epilogue that leaves through LDS as whole rows.
#pragma once
#include "rmcl_common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define ST_T 192
#ifndef ST_STAMP
#define ST_STAMP(i)
#endif

template <typename TO>
__device__ __forceinline__ void st_store4(TO* p, const float (&v)[4]) {
  if constexpr (sizeof(TO) == 2) {
    uint2 pk;
    pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = pk;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

enum { ST_AUX_NONE = 0, ST_AUX_RES = 1, ST_AUX_DGELU = 2 };

struct STTile {
  uint32_t oa[3], ob[3];
  int m0, n0, m_end;
  int kt0, nk;                    // k-tile range of this work item (split-K: a slice of K)
  long zoff;                      // element offset of the item's output slab (split-K), else 0
};

// tile of workgroup b in round r (grid G workgroups, T tiles): slots of one XCD (b & 7) are consecutive tile ids
__device__ __forceinline__ int st_tile_id(int b, int r, int G, int T) {
  const int n_r = min(G, T - r * G);
  if (n_r <= 0) return -1;
  const int x = b & 7, s = b >> 3, q = n_r >> 3, e = n_r & 7;
  if (s >= q + (x < e ? 1 : 0)) return -1;
  return r * G + (x < e ? x * (q + 1) : e * (q + 1) + (x - e) * q) + s;
}

// Epilogue through LDS: the MFMA accumulator layout gives each wave-instruction sixteen 32-byte (bf16) / 64-byte (fp32)
// row segments, which the memory system writes at 3.2 TB/s chip-wide against 5.9 TB/s for whole rows
// (tools/store_pattern_bench.hip).  Each wave group (the four waves that share 96 tile rows) therefore writes its
// converted fragments into a [rows][192] image in LDS (16-byte chunk c of row r at chunk c ^ (r & 7): conflict-free for 8
// rows) and stores it back row by row, 16 bytes per lane.  `scratch`: this GROUP's region (24 KiB) inside the LDS stage of
// the tile's last k-tile, which nothing reads any more and the next tile's DMA overwrites only two phases later.
// Barriers: s_barrier is workgroup-wide, the two groups stand at different epilogue steps when it releases (they run one
// barrier apart), and both execute the same number (2 + 2 * chunks), so the skew survives the epilogue.
// LNF = 1: LayerNorm-folded consumer (EPI_LNFOLD): v = rstd_m * (acc - mean_m * s_n) + c_n, row statistics from the
//          producer's partials (this group's 96 rows, one thread each, into `rowstat`: 192 x (mean, rstd) in LDS)
// LNF = 2: producer (EPI_ROWSTAT): bf16 copy of the fp32 output + per-row partial sums of this wave's 48 columns; with
//          g.ln_center both are taken of (x - c_m) (gemm.h: LN is shift-invariant, so any per-row centre is exact) - the centres of
//          the group's 96 rows are parked in `rowstat` (one float per tile row) behind the epilogue's first barrier
template <int AUX, typename TO, bool DROP, int LNF = 0, bool ACCPRE = false>
__device__ __forceinline__ void st_epilogue_lds(const f32x4 (&acc)[6][3], const GemmArgs& g, const STTile& T, int wm, int wn, int lane, int wave,
                                                char* scratch, float* rowstat = nullptr) {
  constexpr int ESZ = sizeof(TO), RI = ESZ == 2 ? 3 : 2, NCH = 6 / RI, ROWS = RI * 16, ROWB = 192 * ESZ, PIECES = ROWB / 16;
  const int epi = g.epi;
  TO* C = reinterpret_cast<TO*>(g.C) + T.zoff;
  TO* C2 = reinterpret_cast<TO*>(g.C2);
  asm volatile("" : "+v"(lane));
  const int nb = T.n0 + wn * 48 + 4 * (lane >> 4);
  const int mb = T.m0 + wm * 96 + (lane & 15);
  // aux operand (fp32 residual / bf16 pre-activation) of chunk ch + 1 is fetched while chunk ch is converted and stored: the
  // residual stream is HBM-cold here, and a load -> wait -> compute sequence per chunk put its latency on every chunk
  float4 resb[2][AUX == ST_AUX_RES ? RI : 1][3];
  uint2 preb[2][AUX == ST_AUX_DGELU ? RI : 1][3];
  auto aux_fetch = [&](int ch, int buf) {
#pragma unroll
    for (int il = 0; il < RI; ++il) {
      const long mr = min(mb + (ch * RI + il) * 16, g.M - 1);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (AUX == ST_AUX_RES) resb[buf][il][j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + mr * g.ld_aux + nb + j * 16);
        if (AUX == ST_AUX_DGELU) preb[buf][il][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + mr * g.ld_aux + nb + j * 16);
      }
    }
  };
  // chunk 0's fetch is issued HERE, ahead of the epilogue's first barrier: group 1 runs one barrier behind group 0, and behind that
  // barrier its fetch went out only when group 0 had already waited for its own (HBM-cold residual: ~5 us) - the two groups paid
  // the latency one after the other (tools/st_trace.py: fc2 epilogue 20 us)
  if constexpr (AUX != ST_AUX_NONE) aux_fetch(0, 0);
  float4 bias[3], lns[LNF == 1 ? 3 : 1];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if constexpr (LNF == 1) {
      bias[j] = *reinterpret_cast<const float4*>(g.ln_c + nb + j * 16);
      lns[j] = *reinterpret_cast<const float4*>(g.ln_s + nb + j * 16);
    } else {
      bias[j] = (epi & EPI_BIAS) ? *reinterpret_cast<const float4*>(g.bias + nb + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if constexpr (LNF == 1) {
    const int tgs = (wave & 3) * 64 + lane;
    if (tgs < 96) {
      const int row = wm * 96 + tgs;
      const long m = min(T.m0 + row, g.M - 1);
      const float4* pp = reinterpret_cast<const float4*>(g.ln_part + m * (long)(g.ln_nparts * 2));
      const bool stash_m = T.n0 == 0 && g.ln_mean && T.m0 + row < T.m_end;
      const float cen = (stash_m && g.ln_center) ? g.ln_center[m] : 0.f;   // the partials are sums of (x - c): the TRUE mean is mean + c
      float s1 = 0.f, s2 = 0.f;
      if (g.ln_nparts == 16) {                                 // (the step's shape: all 8 loads in flight together, not 8 L2 round trips)
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pp[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1 += v[q].x + v[q].z; s2 += v[q].y + v[q].w; }
      } else {
        for (int q = 0; q < g.ln_nparts / 2; ++q) { const float4 v = pp[q]; s1 += v.x + v.z; s2 += v.y + v.w; }
      }
      const float inv = 1.0f / (float)g.ln_cols, mean = s1 * inv;
      const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + g.ln_eps);
      rowstat[2 * row] = mean;
      rowstat[2 * row + 1] = rstd;
      if (stash_m) { g.ln_mean[m] = mean + cen; g.ln_rstd[m] = rstd; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (LNF == 2) {
    const int tgs = (wave & 3) * 64 + lane;
    if (tgs < 96) rowstat[wm * 96 + tgs] = g.ln_center ? g.ln_center[min(T.m0 + wm * 96 + tgs, g.M - 1)] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  ST_STAMP(3);
  __builtin_amdgcn_s_barrier();                              // the other group's last reads of this stage have retired
  ST_STAMP(4);
  float2 rs[LNF == 1 ? 6 : 1];
  if constexpr (LNF == 1) {                                    // mean / rstd of this lane's six rows, read once
#pragma unroll
    for (int i = 0; i < 6; ++i) rs[i] = *reinterpret_cast<const float2*>(rowstat + 2 * (wm * 96 + i * 16 + (lane & 15)));
  }
  float cen6[LNF == 2 ? 6 : 1];
  if constexpr (LNF == 2) {                                    // centres of this lane's six rows
#pragma unroll
    for (int i = 0; i < 6; ++i) cen6[i] = rowstat[wm * 96 + i * 16 + (lane & 15)];
  }
  // ACCPRE (the weight-gradient launch, C += tile): the OLD values of the whole tile are fetched in ONE batch ahead of the chunk loop.
  // Inside the store loop every iteration was "LDS read, global load, wait, add, store": 18 serial HBM round trips per tile = 22 us of
  // epilogue behind a 176 us k-loop (tools/st_trace.py dw); one batch per chunk: 13 us.
  constexpr int NIT = (ROWS * PIECES + 255) / 256;
  float4 oldv[ACCPRE ? NCH : 1][ACCPRE ? NIT : 1];
  if constexpr (ACCPRE && ESZ == 4) {
    const int tg0 = (wave & 3) * 64 + lane;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = it * 256 + tg0, row = q / PIECES, cp = q - row * PIECES;
        const int m = T.m0 + wm * 96 + ch * ROWS + row;
        oldv[ch][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < ROWS * PIECES && m < T.m_end)
          oldv[ch][it] = *reinterpret_cast<const float4*>(C + (long)m * g.ldc + T.n0 + (cp ^ (row & 7)) * 4);
      }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if constexpr (AUX != ST_AUX_NONE) { if (ch + 1 < NCH) aux_fetch(ch + 1, (ch + 1) & 1); }
    auto& res = resb[ch & 1];
    auto& pre = preb[ch & 1];
#pragma unroll
    for (int il = 0; il < RI; ++il) {
      const int i = ch * RI + il;
      const int m = mb + i * 16;
      const bool live = m < T.m_end;
      float mean_m = 0.f, rstd_m = 1.f, ps1 = 0.f, ps2 = 0.f;
      if constexpr (LNF == 1) { mean_m = rs[i].x; rstd_m = rs[i].y; }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float v[4] = {g.alpha * acc[i][j][0] + bias[j].x, g.alpha * acc[i][j][1] + bias[j].y, g.alpha * acc[i][j][2] + bias[j].z,
                      g.alpha * acc[i][j][3] + bias[j].w};
        if constexpr (LNF == 1) {
          v[0] = fmaf(rstd_m, acc[i][j][0] - mean_m * lns[j].x, bias[j].x); v[1] = fmaf(rstd_m, acc[i][j][1] - mean_m * lns[j].y, bias[j].y);
          v[2] = fmaf(rstd_m, acc[i][j][2] - mean_m * lns[j].z, bias[j].z); v[3] = fmaf(rstd_m, acc[i][j][3] - mean_m * lns[j].w, bias[j].w);
        }
        if (DROP && (epi & EPI_DROP_BWD)) {
          const uint32_t di = (uint32_t)((long)m * g.ld_aux + nb + j * 16);
          drop_scale4(g.drop_seed, di, g.drop_thresh, g.drop_inv_keep, v[0], v[1], v[2], v[3]);
        }
        if (AUX == ST_AUX_DGELU) {
          const uint2 u = pre[il][j];
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
        if (AUX == ST_AUX_RES) { v[0] += res[il][j].x; v[1] += res[il][j].y; v[2] += res[il][j].z; v[3] += res[il][j].w; }
        if constexpr (LNF == 2) {
          const float c0 = cen6[i], d0 = v[0] - c0, d1 = v[1] - c0, d2 = v[2] - c0, d3 = v[3] - c0;
          ps1 += (d0 + d1) + (d2 + d3);
          ps2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        // image: row il*16 + lane%16, element column wn*48 + j*16 + 4*(lane/16); 16-byte chunk index XOR (row & 7)
        const int row = il * 16 + (lane & 15);
        if constexpr (ESZ == 2) {
          const int e8 = wn * 12 + j * 4 + (lane >> 4);      // 8-byte unit (4 bf16) in the row
          const int c16 = (e8 >> 1) ^ (row & 7);
          uint2 pk;
          pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(scratch + row * ROWB + c16 * 16 + (e8 & 1) * 8) = pk;
        } else {
          const int c16 = (wn * 12 + j * 4 + (lane >> 4)) ^ (row & 7);
          *reinterpret_cast<float4*>(scratch + row * ROWB + c16 * 16) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      if constexpr (LNF == 2) {                                // this wave's 48 columns of row m: sum over the 4 lane groups
        ps1 += __shfl_xor(ps1, 16, 64); ps1 += __shfl_xor(ps1, 32, 64);
        ps2 += __shfl_xor(ps2, 16, 64); ps2 += __shfl_xor(ps2, 32, 64);
        if (lane < 16 && live)
          *reinterpret_cast<float2*>(g.ln_part + ((long)m * g.ln_nparts + (T.n0 / ST_T) * 4 + wn) * 2) = make_float2(ps1, ps2);
      }
    }
    ST_STAMP(5 + 4 * ch);
    __builtin_amdgcn_s_barrier();                            // the group's image of this chunk is complete
    ST_STAMP(6 + 4 * ch);
    int tg = (wave & 3) * 64 + lane;                         // lane id inside the group
    asm volatile("" : "+v"(tg));                             // (keeps the read-back addresses from being hoisted out of the tile loop as live registers)
#pragma unroll
    for (int q0 = 0; q0 < ROWS * PIECES; q0 += 256) {
      const int q = q0 + tg;
      if (q < ROWS * PIECES) {
        const int row = q / PIECES, cp = q - row * PIECES;   // physical chunk cp holds logical chunk cp ^ (row & 7)
        const int m = T.m0 + wm * 96 + ch * ROWS + row;
        if (m < T.m_end) {
          const float4 w = *reinterpret_cast<const float4*>(scratch + row * ROWB + cp * 16);
          TO* dst = C + (long)m * g.ldc + T.n0 + (cp ^ (row & 7)) * (16 / ESZ);
          if constexpr (ESZ == 4) {
            float4 o = w;
            if constexpr (ACCPRE) {
              const float4 old = oldv[ch][q0 / 256];
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            } else if (epi & EPI_ACCUM) {
              const float4 old = *reinterpret_cast<const float4*>(dst);
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *reinterpret_cast<float4*>(dst) = o;
            if constexpr (LNF == 2) {                          // bf16 copy of the (centred) residual stream: the next GEMM's A operand
              const float c0 = rowstat[wm * 96 + ch * ROWS + row];
              uint2 pk;
              pk.x = (uint32_t)f2bf(o.x - c0) | ((uint32_t)f2bf(o.y - c0) << 16);
              pk.y = (uint32_t)f2bf(o.z - c0) | ((uint32_t)f2bf(o.w - c0) << 16);
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C2) + (long)m * g.ldc + T.n0 + (cp ^ (row & 7)) * 4) = pk;
            }
          } else {
            *reinterpret_cast<float4*>(dst) = w;
          }
        }
      }
    }
    ST_STAMP(7 + 4 * ch);
    __builtin_amdgcn_s_barrier();                            // image consumed: the next chunk may overwrite it
    ST_STAMP(8 + 4 * ch);
  }
  __builtin_amdgcn_s_barrier();                              // the OTHER group (one barrier behind) has consumed its last image too:
}                                                            // the next tile's LDS-DMA may now target this stage
