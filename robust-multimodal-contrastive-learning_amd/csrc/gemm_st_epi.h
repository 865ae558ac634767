// Shared pieces of the 192-row tile GEMM kernels (gemm_st.hip: one 8-wave workgroup per CU; gemm_dp.hip: two 4-wave
// workgroups per CU): tile descriptor, XCD-aware tile order, and the epilogue that leaves through LDS as whole rows.
#pragma once
#include <type_traits>
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
  // Addresses of the epilogue, once per tile (round 4: recomputed per value group / per store they were, with the divisions by PIECES and the
  // 64-bit pointer arithmetic, about a third of the epilogue's VALU instructions - MFMAs idle).  Image writes: row il*16 + lane%16 has the
  // same (row & 7) for every il: one offset per j, the row block as an immediate.  Read-back + row stores: work item q = tg + 256 k ->
  // (row, piece) = (q / PIECES, q % PIECES); 768 = ROWS3 * PIECES, so k and k + 3 differ by exactly ROWS3 rows: three (row, piece) pairs.
  // Computed from the per-call opaque `lane`, so that they are not hoisted out of the persistent tile loop as live registers of the k-loop.
  constexpr int RSTEP = 256 / PIECES, ROWS3 = 768 / PIECES, NT3 = (NIT + 2) / 3;
  int woff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = lane & 15;
    if constexpr (ESZ == 2) {
      const int e8 = wn * 12 + j * 4 + (lane >> 4);          // 8-byte unit (4 bf16) in the row
      woff[j] = r * ROWB + (((e8 >> 1) ^ (r & 7)) * 16) + (e8 & 1) * 8;
    } else {
      woff[j] = r * ROWB + (((wn * 12 + j * 4 + (lane >> 4)) ^ (r & 7)) * 16);
    }
  }
  int rb_row[3], rb_lds[3];
  uint32_t rb_dst[3];
  {
    const int tg = (wave & 3) * 64 + lane, row0 = tg / PIECES, cp0 = tg - row0 * PIECES;   // lane id inside the group
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const int t16 = cp0 + 16 * kk, wrap = (t16 >= PIECES ? 1 : 0) + (t16 >= 2 * PIECES ? 1 : 0), cp = t16 - wrap * PIECES;
      const int row = row0 + RSTEP * kk + wrap;              // physical chunk cp holds logical chunk cp ^ (row & 7)
      rb_row[kk] = row;
      rb_lds[kk] = row * ROWB + cp * 16;
      rb_dst[kk] = (uint32_t)row * (uint32_t)g.ldc + (uint32_t)((cp ^ (row & 7)) * (16 / ESZ));
    }
  }
  float4 oldv[ACCPRE ? NCH : 1][ACCPRE ? NIT : 1];
  if constexpr (ACCPRE && ESZ == 4) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int mrow = T.m0 + wm * 96 + ch * ROWS, lim = min(ROWS, T.m_end - mrow);
      const uint32_t cbase = (uint32_t)mrow * (uint32_t)g.ldc + (uint32_t)T.n0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int kk = it % 3, t3 = it / 3;
        oldv[ch][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb_row[kk] + t3 * ROWS3 < lim)
          oldv[ch][it] = *reinterpret_cast<const float4*>(C + (size_t)(cbase + (uint32_t)(t3 * ROWS3) * (uint32_t)g.ldc + rb_dst[kk]));
      }
    }
  }
  // The chunk loop twice: PLAIN (no GELU, no pre-activation stash, no += into C: every launch of the step but the odd-shaped ones) without
  // the per-value-group tests of those flags - each was a scalar branch + mask set-up per four values and a basic-block boundary.
  // (dropout instantiations: the mask that applies - 1 the output's, 2 the forward's on the way back, 3 both tested at run time - is a second
  // parameter; the element index of a group is a 32-bit add to a per-lane base)
  const uint32_t dbase_aux = DROP ? (uint32_t)mb * (uint32_t)g.ld_aux + (uint32_t)nb : 0u, dbase_c = DROP ? (uint32_t)mb * (uint32_t)g.ldc + (uint32_t)nb : 0u;
  auto chunk_loop = [&](auto plain_c, auto dm_c) {
    constexpr bool PLAIN = decltype(plain_c)::value;
    constexpr int DM = decltype(dm_c)::value;
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
          // (all four values of the accumulator at once: rmcl_common.h, "Four-wide forms")
          const f32x4 av = acc[i][j], bv = f4v(bias[j]);
          f32x4 v;
          if constexpr (LNF == 1) v = rstd_m * (av - mean_m * f4v(lns[j])) + bv;
          else v = g.alpha * av + bv;
          if constexpr (DROP && (DM & 2) != 0) {
            if (DM == 2 || (epi & EPI_DROP_BWD)) {
              const uint32_t di = dbase_aux + (uint32_t)(i * 16) * (uint32_t)g.ld_aux + (uint32_t)(j * 16);
              drop_scale4(g.drop_seed, di, g.drop_thresh, g.drop_inv_keep, v);
            }
          }
          if (AUX == ST_AUX_DGELU) v *= gelu_poly_grad4(bf2f4(pre[il][j]));
          const long ci = (long)m * g.ldc + nb + j * 16;
          if (!PLAIN && (epi & EPI_SAVE_PREACT) && live) {
            const float t[4] = {v.x, v.y, v.z, v.w};
            st_store4<TO>(C2 + ci, t);
          }
          if (!PLAIN && (epi & EPI_GELU)) v = gelu_poly4(v);
          if constexpr (DROP && (DM & 1) != 0) {
            if (DM == 1 || (epi & EPI_DROPOUT))
              drop_scale4(g.drop_seed, dbase_c + (uint32_t)(i * 16) * (uint32_t)g.ldc + (uint32_t)(j * 16), g.drop_thresh, g.drop_inv_keep, v);
          }
          if (AUX == ST_AUX_RES) v += f4v(res[il][j]);
          if constexpr (LNF == 2) {
            const float c0 = cen6[i], d0 = v.x - c0, d1 = v.y - c0, d2 = v.z - c0, d3 = v.w - c0;
            ps1 += (d0 + d1) + (d2 + d3);
            ps2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
          // image: row il*16 + lane%16, element column wn*48 + j*16 + 4*(lane/16); 16-byte chunk index XOR (row & 7)
          if constexpr (ESZ == 2) {
            *reinterpret_cast<uint2*>(scratch + woff[j] + il * 16 * ROWB) = f2bf4(v);
          } else {
            *reinterpret_cast<float4*>(scratch + woff[j] + il * 16 * ROWB) = make_float4(v.x, v.y, v.z, v.w);
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
      {
        const int mrow = T.m0 + wm * 96 + ch * ROWS, lim = min(ROWS, T.m_end - mrow);   // (uniform) first row of this group's chunk, live rows in it
        const uint32_t cbase = (uint32_t)mrow * (uint32_t)g.ldc + (uint32_t)T.n0;
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            if (kk + 3 * t3 < NIT && rb_row[kk] + t3 * ROWS3 < lim) {
              const float4 w = *reinterpret_cast<const float4*>(scratch + rb_lds[kk] + t3 * ROWS3 * ROWB);
              const size_t off = (size_t)(cbase + (uint32_t)(t3 * ROWS3) * (uint32_t)g.ldc + rb_dst[kk]);
              TO* dst = C + off;
              if constexpr (ESZ == 4) {
                float4 o = w;
                if constexpr (ACCPRE) {
                  const float4 old = oldv[ch][kk + 3 * t3];
                  o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                } else if (!PLAIN && (epi & EPI_ACCUM)) {
                  const float4 old = *reinterpret_cast<const float4*>(dst);
                  o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *reinterpret_cast<float4*>(dst) = o;
                if constexpr (LNF == 2) {                        // bf16 copy of the (centred) residual stream: the next GEMM's A operand
                  const float c0 = rowstat[wm * 96 + ch * ROWS + rb_row[kk] + t3 * ROWS3];
                  const uint2 pk = make_uint2(f2bf2(f32x2{o.x, o.y} - c0), f2bf2(f32x2{o.z, o.w} - c0));
                  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C2) + off) = pk;
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
  };
  {
    using P0 = std::integral_constant<bool, false>;
    using P1 = std::integral_constant<bool, true>;
    const bool plain = !(epi & (EPI_GELU | EPI_SAVE_PREACT | EPI_ACCUM));
    if constexpr (!DROP) {
      if (plain) chunk_loop(P1{}, std::integral_constant<int, 0>{}); else chunk_loop(P0{}, std::integral_constant<int, 0>{});
    } else {
      const int dm = ((epi & EPI_DROPOUT) ? 1 : 0) | ((epi & EPI_DROP_BWD) ? 2 : 0);
      if (plain && dm == 1) chunk_loop(P1{}, std::integral_constant<int, 1>{});
      else if (plain && dm == 2) chunk_loop(P1{}, std::integral_constant<int, 2>{});
      else if (plain) chunk_loop(P1{}, std::integral_constant<int, 3>{});
      else chunk_loop(P0{}, std::integral_constant<int, 3>{});
    }
  }
  __builtin_amdgcn_s_barrier();                              // the OTHER group (one barrier behind) has consumed its last image too:
}                                                            // the next tile's LDS-DMA may now target this stage
