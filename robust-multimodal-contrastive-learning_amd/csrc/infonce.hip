// Fused MoCo InfoNCE (objectives.py:328-334,351; pgd_attack_vilt.py:152-158) against the
// momentum queue, forward + gradient wrt the query in ONE pass over the queue, plus the
// queue-distance metrics (objectives.py:337-349) as an epilogue.
//
//   logits_i = [q_i.k_i, q_i @ queue] / T ;  loss = mean_i (logsumexp(logits_i) - logits_i[0])
//   dq_i = gscale/T * ( sum_j softmax_ij * key_j - k_i )
//
// The [B, 1+Kq] logits (16.8 MB at B=64) and the queue clone are never materialised: each
// workgroup streams a 128-column slice of the queue (64 KB, coalesced 512-B rows) into LDS once,
// computes its 64x128 logit tile with the exact-f32 matrix cores, a block-local softmax, and the
// partial dq = P_tile @ slice^T from the same LDS image; a tiny combine kernel merges the
// per-slice (max, sum, dq) partials with the usual log-sum-exp rescaling.
#include "rmcl_common.h"
#include "kernels.h"

#define PD 128       // projection dim (MOCOHead output, heads.py:129-143)
#define SLICE 128    // queue columns per workgroup
#define RT 64        // query rows per workgroup
#define ILD 129      // LDS row pitch (odd -> conflict-free column & row walks with ds_read_b32)
#define NPART 8      // per-(slice,row) scalars: m, z, best, best_idx, dist_sum, cos_sum, dot_sum, unused

__global__ __launch_bounds__(256) void infonce_partial_kernel(const float* __restrict__ q, const float* __restrict__ queue, long Kq,
                                                              int B, float invT, float* __restrict__ part, float* __restrict__ dq_part,
                                                              int Bpad) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                    // [PD][ILD]   Qs[c][j]
  float* qs = Qs + PD * ILD;         // [RT][ILD]   qs[i][c]
  float* Ps = qs + RT * ILD;         // [RT][ILD]   logits then probabilities
  float* c2 = Ps + RT * ILD;         // [SLICE] squared column norms
  float* q2 = c2 + SLICE;            // [RT] squared query norms

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int slice = blockIdx.x, r0 = blockIdx.y * RT;
  const long j0 = (long)slice * SLICE;

  for (int i = 0; i < 16; ++i) {
    const int v = t + 256 * i, c = v >> 5, jq = (v & 31) * 4;
    const float4 x = *reinterpret_cast<const float4*>(queue + (long)c * Kq + j0 + jq);
    float* d = Qs + c * ILD + jq;
    d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
  }
  for (int i = 0; i < 8; ++i) {
    const int v = t + 256 * i, r = v >> 5, cq = (v & 31) * 4;
    float4 x = make_float4(0, 0, 0, 0);
    if (r0 + r < B) x = *reinterpret_cast<const float4*>(q + (long)(r0 + r) * PD + cq);
    float* d = qs + r * ILD + cq;
    d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
  }
  __syncthreads();
  if (t < SLICE) {
    float s = 0.f;
    for (int c = 0; c < PD; ++c) { const float v = Qs[c * ILD + t]; s += v * v; }
    c2[t] = s;
  } else if (t < SLICE + RT) {
    const int r = t - SLICE;
    float s = 0.f;
    for (int c = 0; c < PD; ++c) { const float v = qs[r * ILD + c]; s += v * v; }
    q2[r] = s;
  }

  const int wm = wave >> 1, wn = wave & 1;
  {  // step 1: S[64 x 128] = qs @ Qs   (A[i][k=c] = qs, B[k=c][j] = Qs)
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const float* ap = qs + (32 * wm + (lane & 31)) * ILD + (lane >> 5);
    const float* bp = Qs + (lane >> 5) * ILD + 64 * wn + (lane & 31);
#pragma unroll 8
    for (int k = 0; k < PD; k += 2) {
      const float a = ap[k], b0 = bp[k * ILD], b1 = bp[k * ILD + 32];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = 64 * wn + (lane & 31);
      Ps[row * ILD + col] = a0[r];
      Ps[row * ILD + col + 32] = a1[r];
    }
  }
  __syncthreads();
  {  // step 2: per-row block softmax + metrics; 4 threads per row, columns interleaved by 4
    const int row = t >> 2, sub = t & 3;
    float m = -INFINITY, best = -INFINITY;
    int bi = 0;
    float sd = 0.f, sc = 0.f, so = 0.f;
    const float qq = q2[row];
    for (int i = 0; i < SLICE / 4; ++i) {
      const int col = 4 * i + sub;
      const float dot = Ps[row * ILD + col];
      if (dot > best) { best = dot; bi = col; }
      const float cc = c2[col];
      sd += sqrtf(fmaxf(qq + cc - 2.f * dot, 0.f));
      sc += dot / fmaxf(sqrtf(qq) * sqrtf(cc), 1e-6f);
      so += dot;
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      sd += __shfl_xor(sd, o, 64);
      sc += __shfl_xor(sc, o, 64);
      so += __shfl_xor(so, o, 64);
    }
    m = best * invT;
    float z = 0.f;
    for (int i = 0; i < SLICE / 4; ++i) {
      const int col = 4 * i + sub;
      const float p = __expf(Ps[row * ILD + col] * invT - m);
      Ps[row * ILD + col] = p;
      z += p;
    }
    z += __shfl_xor(z, 1, 64);
    z += __shfl_xor(z, 2, 64);
    if (sub == 0 && r0 + row < B) {
      float* o = part + ((long)slice * Bpad + r0 + row) * NPART;
      o[0] = m; o[1] = z; o[2] = best; o[3] = __int_as_float((int)(j0 + bi));
      o[4] = sd; o[5] = sc; o[6] = so; o[7] = 0.f;
    }
  }
  __syncthreads();
  {  // step 3: dq_part[64 x 128(c)] = Ps[64 x 128(j)] @ Qs^T  (A[i][k=j] = Ps, B[k=j][c] = Qs[c][j])
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const float* ap = Ps + (32 * wm + (lane & 31)) * ILD + (lane >> 5);
    const float* bp0 = Qs + (64 * wn + (lane & 31)) * ILD + (lane >> 5);
    const float* bp1 = bp0 + 32 * ILD;
#pragma unroll 8
    for (int k = 0; k < SLICE; k += 2) {
      const float a = ap[k];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp0[k], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp1[k], a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = 64 * wn + (lane & 31);
      if (row < B) {
        float* o = dq_part + ((long)slice * Bpad + row) * PD;
        o[col] = a0[r];
        o[col + 32] = a1[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 2: the same pass with 256 queue columns per workgroup as FOUR 64-column sub-slices folded on the fly (running
// max / sum / dq with the usual rescaling), so half as many per-workgroup partials reach the combine kernel, the query
// block lives in registers (A operand of the logit MFMA), and the next sub-slice's 32 KB are in flight (global -> registers
// -> the other LDS buffer) while the current one is computed.  One workgroup per CU at Kq = 65 536 (256 workgroups).
// ---------------------------------------------------------------------------------------------------------------------
#define SL2 64       // columns per sub-slice
#define NS2 4        // sub-slices per workgroup
#define ILD2 65


// MFMA loops of the folded forms.  hipcc serialises "ds_read -> s_waitcnt lgkmcnt(0) -> MFMA" when the LDS operand of each MFMA
// is read inside the loop (one LDS round trip per MFMA pair, the matrix cores idle half of the time), so the operands are read
// in batches into registers one batch AHEAD of the MFMAs that use them, with scheduling barriers pinning the two blocks.
// (Measured: this alone does not move the kernel - 44.6 -> 44.2 us - the MFMAs are not what bounds it; an eight-wave
// ping-pong form of the same pass, two groups one barrier apart, measured 46.6 - 50 us and was dropped.  What did cost was the
// query block: indexed by a rolled loop it lived in SCRATCH memory, every MFMA fetching its A operand from there: 51.3 -> 44.6 us.)
// logits: a0 += q[:, 2i..2i+1] x Qs[2i..2i+1, :] for i < PD/2 (areg[i] = this lane's q value of k-step i)
#define NCE_LOGITS_MFMA(a0, areg, bp)                                                                               \
  {                                                                                                                 \
    float bv[2][8];                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) bv[0][j] = (bp)[2 * j * ILD2];                                    \
    _Pragma("unroll") for (int ic = 0; ic < PD / 2; ic += 8) {                                                      \
      if (ic + 8 < PD / 2) {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) bv[((ic >> 3) + 1) & 1][j] = (bp)[2 * (ic + 8 + j) * ILD2];   \
      }                                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                 \
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32((areg)[ic + j], bv[(ic >> 3) & 1][j], a0, 0, 0, 0);               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                               \
  }
// dq: d0 / d1 += (fsr * P[:, k..k+1]) x Qs^T[k..k+1, 0..31 / 32..63] for k < SL2
#define NCE_DQ_MFMA(d0, d1, ap, bp0, bp1, fsr)                                                                      \
  {                                                                                                                 \
    float av[2][4], b0v[2][4], b1v[2][4];                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { av[0][j] = (ap)[2 * j]; b0v[0][j] = (bp0)[2 * j]; b1v[0][j] = (bp1)[2 * j]; } \
    _Pragma("unroll") for (int kc = 0; kc < SL2 / 2; kc += 4) {                                                     \
      const int cur = (kc >> 2) & 1, nxt = cur ^ 1;                                                                 \
      if (kc + 4 < SL2 / 2) {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
          av[nxt][j] = (ap)[2 * (kc + 4 + j)]; b0v[nxt][j] = (bp0)[2 * (kc + 4 + j)]; b1v[nxt][j] = (bp1)[2 * (kc + 4 + j)]; \
        }                                                                                                           \
      }                                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
        const float a = av[cur][j] * (fsr);                                                                         \
        d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0v[cur][j], d0, 0, 0, 0);                                     \
        d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1v[cur][j], d1, 0, 0, 0);                                     \
      }                                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                               \
  }

__global__ __launch_bounds__(256) void infonce_partial2_kernel(const float* __restrict__ q, const float* __restrict__ queue, long Kq,
                                                               int B, float invT, float* __restrict__ part, float* __restrict__ dq_part,
                                                               int Bpad) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  auto Qbuf = [&](int i) { return sm + i * (PD * ILD2); };   // two [PD][ILD2] buffers: Qs[c][j]
  float* Ps = sm + 2 * PD * ILD2;        // [RT][ILD2]
  float* c2p = Ps + RT * ILD2;           // [4][SL2] squared column norms: the partial of each wave's 32 rows (written by deposit)
  float* q2 = c2p + 4 * SL2;             // [RT]
  float* fo = q2 + RT;                   // [RT] rescale of the running sums for this sub-slice
  float* fs = fo + RT;                   // [RT] weight of this sub-slice

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = blockIdx.x, r0 = blockIdx.y * RT;
  const long j00 = (long)wg * (SL2 * NS2);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- query block -> LDS (Qb[1] as scratch, pitch 129) -> registers: areg[i] = q[32 wm + lane%32][2 i + lane/32]
  {
    float* qs = Qbuf(1);
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, r = v >> 5, cq = (v & 31) * 4;
      float4 x = make_float4(0, 0, 0, 0);
      if (r0 + r < B) x = *reinterpret_cast<const float4*>(q + (long)(r0 + r) * PD + cq);
      float* d = qs + r * 129 + cq;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
      float sq = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);   // squared row norm: the 32 lanes of a half-wave hold row r
#pragma unroll
      for (int o = 1; o <= 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
      if ((t & 31) == 0) q2[r] = sq;
    }
  }
  float4 pre[8];                                               // one sub-slice in flight: 32 floats per thread
  auto fetch = [&](int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, c = v >> 4, jq = (v & 15) * 4;
      pre[i] = *reinterpret_cast<const float4*>(queue + (long)c * Kq + j00 + (long)s * SL2 + jq);
    }
  };
  // registers -> LDS image; the squared column norms of the sub-slice (queue-distance metrics) are summed on the way: a thread
  // holds 8 rows x 4 columns, the 4 lanes with equal lane % 16 cover this wave's 32 rows (a 64-thread serial walk down the
  // 128 rows of the image cost 3.4 us per sub-slice with the other three waves waiting at the barrier)
  auto deposit = [&](float* Qs) {
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, c = v >> 4, jq = (v & 15) * 4;
      float* d = Qs + c * ILD2 + jq;
      d[0] = pre[i].x; d[1] = pre[i].y; d[2] = pre[i].z; d[3] = pre[i].w;
      cs0 += pre[i].x * pre[i].x; cs1 += pre[i].y * pre[i].y; cs2 += pre[i].z * pre[i].z; cs3 += pre[i].w * pre[i].w;
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      cs0 += __shfl_xor(cs0, o, 64); cs1 += __shfl_xor(cs1, o, 64); cs2 += __shfl_xor(cs2, o, 64); cs3 += __shfl_xor(cs3, o, 64);
    }
    if (lane < 16) {
      float* o = c2p + wave * SL2 + lane * 4;
      o[0] = cs0; o[1] = cs1; o[2] = cs2; o[3] = cs3;
    }
  };
  fetch(0);
  __syncthreads();
  float areg[PD / 2];
  {
    const float* ap = Qbuf(1) + (32 * wm + (lane & 31)) * 129 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < PD / 2; ++i) areg[i] = ap[2 * i];
  }
  deposit(Qbuf(0));
  __syncthreads();                                             // (Qb[1] is free from here: q lives in registers)

  f32x16 d0, d1;                                               // running dq: rows 32 wm .., columns 64 wn + {0..31, 32..63}
#pragma unroll
  for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
  // per-row running scalars (the 4 threads of a row keep identical copies)
  const int row = t >> 2, sub = t & 3;
  float m_run = -INFINITY, z_run = 0.f, best = -INFINITY, sd = 0.f, sc = 0.f, so = 0.f;
  int bi = 0;
  const float qq = q2[row];

  for (int s = 0; s < NS2; ++s) {
    float* Qs = Qbuf(s & 1);
    if (s + 1 < NS2) fetch(s + 1);
    {  // step 1: S[64 x 64] = q @ Qs: one 32x32 tile per wave
      f32x16 a0;
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = 0.f;
      const float* bp = Qs + (lane >> 5) * ILD2 + 32 * wn + (lane & 31);
      NCE_LOGITS_MFMA(a0, areg, bp);                               // (static areg indices: a rolled loop puts areg in scratch memory)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Ps[rr * ILD2 + 32 * wn + (lane & 31)] = a0[r];
      }
    }
    __syncthreads();
    {  // step 2: block softmax of this sub-slice + metrics, merged into the running values
      float sbest = -INFINITY;
      int sbi = 0;
      for (int i = 0; i < SL2 / 4; ++i) {
        const int col = 4 * i + sub;
        const float dot = Ps[row * ILD2 + col];
        if (dot > sbest) { sbest = dot; sbi = col; }
        const float cc = (c2p[col] + c2p[SL2 + col]) + (c2p[2 * SL2 + col] + c2p[3 * SL2 + col]);
        sd += sqrtf(fmaxf(qq + cc - 2.f * dot, 0.f));
        sc += dot / fmaxf(sqrtf(qq) * sqrtf(cc), 1e-6f);
        so += dot;
      }
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {
        const float ob = __shfl_xor(sbest, o, 64);
        const int oi = __shfl_xor(sbi, o, 64);
        if (ob > sbest || (ob == sbest && oi < sbi)) { sbest = ob; sbi = oi; }
      }
      const float ms = sbest * invT;
      float z = 0.f;
      for (int i = 0; i < SL2 / 4; ++i) {
        const int col = 4 * i + sub;
        const float p = __expf(Ps[row * ILD2 + col] * invT - ms);
        Ps[row * ILD2 + col] = p;
        z += p;
      }
      z += __shfl_xor(z, 1, 64);
      z += __shfl_xor(z, 2, 64);
      const float mn = fmaxf(m_run, ms), f_old = __expf(m_run - mn), f_s = __expf(ms - mn);   // (m_run = -inf first: f_old = 0)
      z_run = z_run * f_old + z * f_s;
      m_run = mn;
      if (sbest > best) { best = sbest; bi = s * SL2 + sbi; }                                   // (earlier sub-slice wins ties: smaller index)
      if (sub == 0) { fo[row] = f_old; fs[row] = f_s; }
    }
    __syncthreads();
    {  // step 3: dq = dq * f_old + (f_s * P) @ Qs^T
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float f = fo[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        d0[r] *= f;
        d1[r] *= f;
      }
      const float fsr = fs[32 * wm + (lane & 31)];
      const float* ap = Ps + (32 * wm + (lane & 31)) * ILD2 + (lane >> 5);
      const float* bp0 = Qs + (64 * wn + (lane & 31)) * ILD2 + (lane >> 5);
      const float* bp1 = bp0 + 32 * ILD2;
      NCE_DQ_MFMA(d0, d1, ap, bp0, bp1, fsr);
    }
    if (s + 1 < NS2) deposit(Qbuf((s + 1) & 1));                 // the other buffer: its last readers passed the barrier above one
    __syncthreads();                                           // sub-slice ago; also fences Ps / fo / fs for the next round
  }
  // metric sums across the 4 threads of a row
#pragma unroll
  for (int o = 1; o <= 2; o <<= 1) {
    sd += __shfl_xor(sd, o, 64);
    sc += __shfl_xor(sc, o, 64);
    so += __shfl_xor(so, o, 64);
  }
  if (sub == 0 && r0 + row < B) {
    float* o = part + ((long)wg * Bpad + r0 + row) * NPART;
    o[0] = m_run; o[1] = z_run; o[2] = best; o[3] = __int_as_float((int)(j00 + bi));
    o[4] = sd; o[5] = sc; o[6] = so; o[7] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = r0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = 64 * wn + (lane & 31);
    if (rr < B) {
      float* o = dq_part + ((long)wg * Bpad + rr) * PD;
      o[col] = d0[r];
      o[col + 32] = d1[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the folded pass on the bf16 matrix cores with SPLIT operands (x = hi + lo, hi = bf16(x), lo = bf16(x - hi)):
// a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi leaves a 2^-16 relative error per product - logits of +-40 move by ~1e-4, two orders
// below what plain bf16 logits would cost - at 3/16 of the exact-f32 MFMA time (v_mfma_f32_16x16x32_bf16: 16 cycles for 16 K FLOP
// against 64 cycles for 4 K FLOP of v_mfma_f32_32x32x2_f32).  The fp32 queue is read as before (no shadow copy, no change to the
// enqueue) and split on its way into LDS.  One image per sub-slice and half: [128 c][64 j] bf16, 128-byte rows, the transposed-read
// swizzle of attention.hip's V image - read by ds_read_b64_tr_b16 for the logits (contraction over c) and by rows for
// dq += P Qs^T (contraction over j).  Same partials as infonce_partial2_kernel: infonce_combine2_kernel merges them.
//   logits^T tile: D[queue column][query row] = sum_c A[column][c] B[c][row]: wave w owns column tile w, 4 row tiles, K = 128:
//       48 MFMAs per sub-slice; dq^T: D[c][row] = sum_j Qs[c][j] P[row][j]: wave w owns c tiles 2w, 2w+1, 4 row tiles: 48 MFMAs.
// METRICS = false (the PGD passes: only dq is consumed) skips the column norms and the queue-distance sums.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void nce_split8(const float (&x)[8], nbf16x8& hi, nbf16x8& lo) {
  union { nbf16x8 v; bf16_t e[8]; } h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h.e[i] = f2bf(x[i]);
    l.e[i] = f2bf(x[i] - bf2f(h.e[i]));
  }
  hi = h.v;
  lo = l.v;
}
// A operand of a 32-deep k step out of a [k rows][64 columns] image (128-byte rows, 32-byte group swizzle ^ ((row >> 1) & 3)):
// rows of the MFMA = the 16 image columns of tile `dt`, k order (g, j) -> base + 16 (j >> 2) + 4 g + (j & 3)
__device__ __forceinline__ nbf16x8 nce_frag_tr(const char* img, int base, int dt, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  union { nbf16x8 v; s16x4 h[2]; } u;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = base + 16 * half + 4 * g + q;
    const int t = dt ^ ((row >> 1) & 3);
    u.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + row * 128 + t * 32 + p * 8));
  }
  return u.v;
}
// row fragment (rows row0 + lane % 16, k = 32 s + 8 (lane / 16) .. + 7) of the same image
__device__ __forceinline__ nbf16x8 nce_frag_row(const char* img, int row0, int s, int lane) {
  const int row = row0 + (lane & 15);
  const int c = 4 * s + (lane >> 4);
  const int cp = (((c >> 1) ^ ((row >> 1) & 3)) << 1) | (c & 1);
  return *reinterpret_cast<const nbf16x8*>(img + row * 128 + cp * 16);
}

#define NCE3_IMG (PD * 128)                  // one half image: 128 rows x 128 B = 16 KiB

template <bool METRICS>
__global__ __launch_bounds__(256) void infonce_partial3_kernel(const float* __restrict__ q, const float* __restrict__ queue, long Kq,
                                                               int B, float invT, float* __restrict__ part, float* __restrict__ dq_part,
                                                               int Bpad) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  auto Qhi = [&](int i) { return smc + i * (2 * NCE3_IMG); };
  auto Qlo = [&](int i) { return smc + i * (2 * NCE3_IMG) + NCE3_IMG; };
  float* Ps = reinterpret_cast<float*>(smc + 4 * NCE3_IMG);     // [RT][PLD] logits, then probabilities; first the query block [RT][129]
  constexpr int PLD = 68;                                       // 16-byte aligned rows, 4-row period over the 64 banks
  float* c2p = Ps + RT * 132;                                   // [4][SL2]
  float* q2 = c2p + 4 * SL2;                                    // [RT]
  float* fo = q2 + RT;
  float* fs = fo + RT;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
  const int wg = blockIdx.x, r0 = blockIdx.y * RT;
  const long j00 = (long)wg * (SL2 * NS2);

  // ---- query block -> LDS [RT][129] (inside the Ps area) -> B-operand fragments (hi, lo) in the transposed reads' k order
  {
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, r = v >> 5, cq = (v & 31) * 4;
      float4 x = make_float4(0, 0, 0, 0);
      if (r0 + r < B) x = *reinterpret_cast<const float4*>(q + (long)(r0 + r) * PD + cq);
      float* d = Ps + r * 129 + cq;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
      float sq = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
#pragma unroll
      for (int o = 1; o <= 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
      if ((t & 31) == 0) q2[r] = sq;
    }
  }
  float4 pre[8];
  auto fetch = [&](int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, c = v >> 4, jq = (v & 15) * 4;
      pre[i] = *reinterpret_cast<const float4*>(queue + (long)c * Kq + j00 + (long)s * SL2 + jq);
    }
  };
  auto deposit = [&](int buf) {
    char* hi = Qhi(buf);
    char* lo = Qlo(buf);
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = t + 256 * i, c = v >> 4, jq = (v & 15) * 4;
      const float x[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
      bf16_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = f2bf(x[e]); l[e] = f2bf(x[e] - bf2f(h[e])); }
      const int off = c * 128 + (((jq >> 4) ^ ((c >> 1) & 3)) << 5) + ((jq & 15) << 1);
      *reinterpret_cast<uint2*>(hi + off) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      *reinterpret_cast<uint2*>(lo + off) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
      if (METRICS) { cs0 += x[0] * x[0]; cs1 += x[1] * x[1]; cs2 += x[2] * x[2]; cs3 += x[3] * x[3]; }
    }
    if (METRICS) {
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        cs0 += __shfl_xor(cs0, o, 64); cs1 += __shfl_xor(cs1, o, 64); cs2 += __shfl_xor(cs2, o, 64); cs3 += __shfl_xor(cs3, o, 64);
      }
      if (lane < 16) {
        float* o = c2p + wave * SL2 + lane * 4;
        o[0] = cs0; o[1] = cs1; o[2] = cs2; o[3] = cs3;
      }
    }
  };
  fetch(0);
  __syncthreads();
  nbf16x8 qh[4][4], ql[4][4];                                  // [row tile][k step]: q[16 rt + lane % 16][32 ks + 16 (j >> 2) + 4 g + (j & 3)]
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
      const float* src = Ps + (16 * rt + (lane & 15)) * 129 + 32 * ks + 4 * g;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = src[16 * (j >> 2) + (j & 3)];
      nce_split8(x, qh[rt][ks], ql[rt][ks]);
    }
  deposit(0);
  __syncthreads();                                             // (the query block's LDS copy is dead: Ps is free)

  f32x4 dq[2][4];                                              // running dq^T: c tile 2 wave + i, row tile rt: lane = query row, regs = 4 c
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) dq[i][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int row = t >> 2, sub = t & 3;
  float m_run = -INFINITY, z_run = 0.f, best = -INFINITY, sd = 0.f, sc = 0.f, so = 0.f;
  int bi = 0;
  const float qq = q2[row];

  for (int s = 0; s < NS2; ++s) {
    const char* hi = Qhi(s & 1);
    const char* lo = Qlo(s & 1);
    if (s + 1 < NS2) fetch(s + 1);
    {  // step 1: logits^T tile of this wave's 16 columns x 64 rows
      f32x4 acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const nbf16x8 ah = nce_frag_tr(hi, 32 * ks, wave, lane), al = nce_frag_tr(lo, 32 * ks, wave, lane);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, qh[rt][ks], acc[rt], 0, 0, 0);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ql[rt][ks], acc[rt], 0, 0, 0);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, qh[rt][ks], acc[rt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)                            // lane: query row 16 rt + lane % 16, columns 16 wave + 4 g .. + 3
        *reinterpret_cast<float4*>(Ps + (16 * rt + (lane & 15)) * PLD + 16 * wave + 4 * g) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    }
    __syncthreads();
    {  // step 2: block softmax of this sub-slice (+ metrics), merged into the running values: 4 threads per row, 16 columns each
      float sbest = -INFINITY;
      int sbi = 0;
      for (int i = 0; i < SL2 / 4; ++i) {
        const int col = 4 * i + sub;
        const float dot = Ps[row * PLD + col];
        if (dot > sbest) { sbest = dot; sbi = col; }
        if (METRICS) {
          const float cc = (c2p[col] + c2p[SL2 + col]) + (c2p[2 * SL2 + col] + c2p[3 * SL2 + col]);
          sd += sqrtf(fmaxf(qq + cc - 2.f * dot, 0.f));
          sc += dot / fmaxf(sqrtf(qq) * sqrtf(cc), 1e-6f);
          so += dot;
        }
      }
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {
        const float ob = __shfl_xor(sbest, o, 64);
        const int oi = __shfl_xor(sbi, o, 64);
        if (ob > sbest || (ob == sbest && oi < sbi)) { sbest = ob; sbi = oi; }
      }
      const float ms = sbest * invT;
      float z = 0.f;
      for (int i = 0; i < SL2 / 4; ++i) {
        const int col = 4 * i + sub;
        const float p = __expf(Ps[row * PLD + col] * invT - ms);
        Ps[row * PLD + col] = p;
        z += p;
      }
      z += __shfl_xor(z, 1, 64);
      z += __shfl_xor(z, 2, 64);
      const float mn = fmaxf(m_run, ms), f_old = __expf(m_run - mn), f_s = __expf(ms - mn);
      z_run = z_run * f_old + z * f_s;
      m_run = mn;
      if (sbest > best) { best = sbest; bi = s * SL2 + sbi; }
      if (sub == 0) { fo[row] = f_old; fs[row] = f_s; }
    }
    __syncthreads();
    {  // step 3: dq^T = dq^T * f_old + Qs (f_s P)^T
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float f = fo[16 * rt + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 2; ++i) dq[i][rt] *= f;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        nbf16x8 ph[4], pl[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int pr = 16 * rt + (lane & 15);
          const float fsr = fs[pr];
          const float4 a = *reinterpret_cast<const float4*>(Ps + pr * PLD + 32 * ks + 8 * g);
          const float4 b = *reinterpret_cast<const float4*>(Ps + pr * PLD + 32 * ks + 8 * g + 4);
          const float x[8] = {a.x * fsr, a.y * fsr, a.z * fsr, a.w * fsr, b.x * fsr, b.y * fsr, b.z * fsr, b.w * fsr};
          nce_split8(x, ph[rt], pl[rt]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c0 = 16 * (2 * wave + i);
          const nbf16x8 ah = nce_frag_row(hi, c0, ks, lane), al = nce_frag_row(lo, c0, ks, lane);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            dq[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph[rt], dq[i][rt], 0, 0, 0);
            dq[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl[rt], dq[i][rt], 0, 0, 0);
            dq[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph[rt], dq[i][rt], 0, 0, 0);
          }
        }
      }
    }
    if (s + 1 < NS2) deposit((s + 1) & 1);
    __syncthreads();
  }
  if (METRICS) {
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      sd += __shfl_xor(sd, o, 64);
      sc += __shfl_xor(sc, o, 64);
      so += __shfl_xor(so, o, 64);
    }
  }
  if (sub == 0 && r0 + row < B) {
    float* o = part + ((long)wg * Bpad + r0 + row) * NPART;
    o[0] = m_run; o[1] = z_run; o[2] = best; o[3] = __int_as_float((int)(j00 + bi));
    o[4] = sd; o[5] = sc; o[6] = so; o[7] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int rr = r0 + 16 * rt + (lane & 15);
      if (rr < B)
        *reinterpret_cast<float4*>(dq_part + ((long)wg * Bpad + rr) * PD + 16 * (2 * wave + i) + 4 * g) =
            make_float4(dq[i][rt][0], dq[i][rt][1], dq[i][rt][2], dq[i][rt][3]);
    }
}

// One workgroup per query row: merge slice partials, add the positive pair.  1024 threads = 8 groups of PD: group 0 does the
// scalar bookkeeping, all eight split the slices of the Z / dq merge (the serial 512-slice loop was latency-bound).
// rows_out[i, 0..9] = loss_i, pred_i (argmax of the logits incl. the positive at index 0),
//   l_pos, pos_dist, pos_cos, pos_dot, neg_dist_mean, neg_cos_mean, neg_dot_mean, logsumexp
#define CMB_G 8
__global__ __launch_bounds__(CMB_G * PD) void infonce_combine_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ part, const float* __restrict__ dq_part,
                                                             int nslice, int B, int Bpad, long Kq, float invT, float gscale,
                                                             float* __restrict__ dq, float* __restrict__ rows_out,
                                                             float* __restrict__ loss_sum) {
  __shared__ float red[4][2];
  __shared__ float sh[8];
  __shared__ float accs[CMB_G][PD];
  __shared__ float zs[CMB_G];
  const int i = blockIdx.x, c = threadIdx.x & (PD - 1), grp = threadIdx.x / PD, lane = c & 63, wave = c >> 6;
  const float qc = q[(long)i * PD + c], kc = k[(long)i * PD + c];
  float s_qk = wave_sum(qc * kc), s_qq = wave_sum(qc * qc), s_kk = wave_sum(kc * kc), s_d = wave_sum((qc - kc) * (qc - kc));
  if (grp == 0 && lane == 0) { red[0][wave] = s_qk; red[1][wave] = s_qq; red[2][wave] = s_kk; red[3][wave] = s_d; }
  __syncthreads();
  const float dotp = red[0][0] + red[0][1], qq = red[1][0] + red[1][1], kk = red[2][0] + red[2][1], dd = red[3][0] + red[3][1];
  const float lpos = dotp * invT;
  // pass 1 over the slice scalars (strided over threads), reduce max / best
  float M = lpos, best = -INFINITY;
  int bidx = 0;
  float sd = 0.f, sc = 0.f, so = 0.f;
  for (int s = grp == 0 ? c : nslice; s < nslice; s += PD) {
    const float* o = part + ((long)s * Bpad + i) * NPART;
    M = fmaxf(M, o[0]);
    const int oi = __float_as_int(o[3]);
    if (o[2] > best || (o[2] == best && oi < bidx)) { best = o[2]; bidx = oi; }
    sd += o[4]; sc += o[5]; so += o[6];
  }
  __syncthreads();
  // block reductions through LDS (2 waves)
  M = wave_max(M);
  sd = wave_sum(sd); sc = wave_sum(sc); so = wave_sum(so);
  float wb = best; int wi = bidx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(wb, o, 64); const int oi = __shfl_xor(wi, o, 64);
    if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
  }
  __shared__ float r2[2][8];
  if (grp == 0 && lane == 0) { r2[wave][0] = M; r2[wave][1] = sd; r2[wave][2] = sc; r2[wave][3] = so; r2[wave][4] = wb; r2[wave][5] = __int_as_float(wi); }
  __syncthreads();
  M = fmaxf(r2[0][0], r2[1][0]);
  sd = r2[0][1] + r2[1][1]; sc = r2[0][2] + r2[1][2]; so = r2[0][3] + r2[1][3];
  {
    const float b0 = r2[0][4], b1 = r2[1][4];
    const int i0 = __float_as_int(r2[0][5]), i1 = __float_as_int(r2[1][5]);
    if (b1 > b0 || (b1 == b0 && i1 < i0)) { best = b1; bidx = i1; } else { best = b0; bidx = i0; }
  }
  // pass 2: Z and dq (each thread owns column c of dq)
  float Z = 0.f, acc = 0.f;
#pragma unroll 8
  for (int s = grp; s < nslice; s += CMB_G) {
    const float* o = part + ((long)s * Bpad + i) * NPART;
    const float f = __expf(o[0] - M);
    Z += o[1] * f;
    acc += f * dq_part[((long)s * Bpad + i) * PD + c];
  }
  accs[grp][c] = acc;
  if (c == 0) zs[grp] = Z;
  __syncthreads();
  if (grp != 0) return;
  Z = 0.f; acc = 0.f;
#pragma unroll
  for (int gi = 0; gi < CMB_G; ++gi) { Z += zs[gi]; acc += accs[gi][c]; }
  const float epos = __expf(lpos - M);
  Z += epos;
  const float lse = logf(Z) + M;
  if (dq) dq[(long)i * PD + c] = gscale * invT * ((epos / Z - 1.0f) * kc + acc / Z);
  if (c == 0) {
    float* o = rows_out + (long)i * 10;
    const float loss = lse - lpos;
    o[0] = loss;
    o[1] = (dotp >= best) ? 0.f : (float)(bidx + 1);
    o[2] = lpos;
    o[3] = sqrtf(dd);
    o[4] = dotp / fmaxf(sqrtf(qq) * sqrtf(kk), 1e-6f);
    o[5] = dotp;
    o[6] = sd / (float)Kq; o[7] = sc / (float)Kq; o[8] = so / (float)Kq;
    o[9] = lse;
    if (loss_sum) atomicAdd(loss_sum, loss / (float)B);
  }
  (void)sh;
}

// The same merge with the dq columns of a row split over PD / CW workgroups (B x 4 instead of B workgroups: with 64 rows the
// one-workgroup-per-row form kept 64 of 256 CUs busy reading the 8.4 MB of dq partials).  Every workgroup repeats the scalar
// bookkeeping of its row (256 x 8 floats); threads 0..127 play the scalar role, all 256 threads = 8 slice groups x CW columns
// the dq role.  rows_out / loss_sum are written by column block 0.
#define CW 32
__global__ __launch_bounds__(CMB_G * CW) void infonce_combine2_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                      const float* __restrict__ part, const float* __restrict__ dq_part,
                                                                      int nslice, int B, int Bpad, long Kq, float invT, float gscale,
                                                                      float* __restrict__ dq, float* __restrict__ rows_out,
                                                                      float* __restrict__ loss_sum) {
  __shared__ float red[4][2];
  __shared__ float r2[2][8];
  __shared__ float accs[CMB_G][CW];
  __shared__ float zs[CMB_G];
  const int i = blockIdx.x, cb = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool scalar = t < PD;                                  // waves 0 and 1
  const int c = t & (PD - 1);
  const float qc = q[(long)i * PD + c], kc = k[(long)i * PD + c];
  {
    const float s_qk = wave_sum(qc * kc), s_qq = wave_sum(qc * qc), s_kk = wave_sum(kc * kc), s_d = wave_sum((qc - kc) * (qc - kc));
    if (scalar && lane == 0) { red[0][wave] = s_qk; red[1][wave] = s_qq; red[2][wave] = s_kk; red[3][wave] = s_d; }
  }
  __syncthreads();
  const float dotp = red[0][0] + red[0][1], qq = red[1][0] + red[1][1], kk = red[2][0] + red[2][1], dd = red[3][0] + red[3][1];
  const float lpos = dotp * invT;
  float M = lpos, best = -INFINITY;
  int bidx = 0;
  float sd = 0.f, sc = 0.f, so = 0.f;
  for (int s = scalar ? c : nslice; s < nslice; s += PD) {
    const float* o = part + ((long)s * Bpad + i) * NPART;
    M = fmaxf(M, o[0]);
    const int oi = __float_as_int(o[3]);
    if (o[2] > best || (o[2] == best && oi < bidx)) { best = o[2]; bidx = oi; }
    sd += o[4]; sc += o[5]; so += o[6];
  }
  M = wave_max(M);
  sd = wave_sum(sd); sc = wave_sum(sc); so = wave_sum(so);
  float wb = best; int wi = bidx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(wb, o, 64); const int oi = __shfl_xor(wi, o, 64);
    if (ob > wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
  }
  if (scalar && lane == 0) { r2[wave][0] = M; r2[wave][1] = sd; r2[wave][2] = sc; r2[wave][3] = so; r2[wave][4] = wb; r2[wave][5] = __int_as_float(wi); }
  __syncthreads();
  M = fmaxf(r2[0][0], r2[1][0]);
  sd = r2[0][1] + r2[1][1]; sc = r2[0][2] + r2[1][2]; so = r2[0][3] + r2[1][3];
  {
    const float b0 = r2[0][4], b1 = r2[1][4];
    const int i0 = __float_as_int(r2[0][5]), i1 = __float_as_int(r2[1][5]);
    if (b1 > b0 || (b1 == b0 && i1 < i0)) { best = b1; bidx = i1; } else { best = b0; bidx = i0; }
  }
  // Z and dq: slice group grp, column cb * CW + cc
  const int grp = t / CW, cc = t - grp * CW, col = cb * CW + cc;
  float Z = 0.f, acc = 0.f;
  if (nslice == 32 * CMB_G) {
    // the step's shape (65 536 queue columns = 256 partials per row): all 32 (max, sum) pairs and all 32 dq values of this thread are
    // fetched in ONE batch - with the loop unrolled by 8 the merge was four dependent rounds of L2 latency
    float2 ms[32];
    float dv[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const long sr = (long)(grp + k * CMB_G) * Bpad + i;
      ms[k] = *reinterpret_cast<const float2*>(part + sr * NPART);
      dv[k] = dq_part[sr * PD + col];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float f = __expf(ms[k].x - M);
      Z += ms[k].y * f;
      acc += f * dv[k];
    }
  } else {
#pragma unroll 8
    for (int s = grp; s < nslice; s += CMB_G) {
      const float* o = part + ((long)s * Bpad + i) * NPART;
      const float f = __expf(o[0] - M);
      Z += o[1] * f;
      acc += f * dq_part[((long)s * Bpad + i) * PD + col];
    }
  }
  accs[grp][cc] = acc;
  if (cc == 0) zs[grp] = Z;
  __syncthreads();
  if (grp != 0) return;
  Z = 0.f; acc = 0.f;
#pragma unroll
  for (int gi = 0; gi < CMB_G; ++gi) { Z += zs[gi]; acc += accs[gi][cc]; }
  const float epos = __expf(lpos - M);
  Z += epos;
  const float lse = logf(Z) + M;
  if (dq) dq[(long)i * PD + col] = gscale * invT * ((epos / Z - 1.0f) * k[(long)i * PD + col] + acc / Z);
  if (cb == 0 && cc == 0) {
    float* o = rows_out + (long)i * 10;
    const float loss = lse - lpos;
    o[0] = loss;
    o[1] = (dotp >= best) ? 0.f : (float)(bidx + 1);
    o[2] = lpos;
    o[3] = sqrtf(dd);
    o[4] = dotp / fmaxf(sqrtf(qq) * sqrtf(kk), 1e-6f);
    o[5] = dotp;
    o[6] = sd / (float)Kq; o[7] = sc / (float)Kq; o[8] = so / (float)Kq;
    o[9] = lse;
    if (loss_sum) atomicAdd(loss_sum, loss / (float)B);
  }
}

int g_infonce_fold = 1;              // rmcl_tune_set key 5: 0 = one slice per workgroup + one-workgroup-per-row combine (round-1 form)


long rmcl_infonce_workspace_bytes(int B, long Kq) {
  const long Bpad = (B + RT - 1) / RT * RT, ns = Kq / SLICE;
  return ns * Bpad * (NPART + PD) * (long)sizeof(float);
}

int rmcl_infonce(const float* q, const float* k, const float* queue, int B, int Pd, long Kq, float T, float gscale, float* dq,
                 float* rows_out, float* loss_sum, void* workspace, hipStream_t s, int form) {
  RMCL_REQUIRE(Pd == PD, "infonce: projection dim must be 128");
  RMCL_REQUIRE(Kq % SLICE == 0 && Kq > 0, "infonce: queue length must be a multiple of 128");
  RMCL_REQUIRE(B > 0, "infonce: empty batch");
  const int Bpad = (B + RT - 1) / RT * RT, ns = (int)(Kq / SLICE);
  float* part = reinterpret_cast<float*>(workspace);
  float* dq_part = part + (long)ns * Bpad * NPART;
  const size_t lds = (PD * ILD + 2 * RT * ILD + SLICE + RT) * sizeof(float);
  static RmclLdsOnce once1;
  RMCL_TRY(rmcl_set_max_lds(once1, reinterpret_cast<const void*>(infonce_partial_kernel), (int)lds));
  int nparts = ns;
  if (form != 0 && Kq % (SL2 * NS2) == 0 && g_infonce_fold >= 1) {
    // split-bf16 matrix cores (form 1: with the queue-distance metrics; 2: without - the PGD passes read dq only)
    nparts = (int)(Kq / (SL2 * NS2));
    const size_t lds3 = 4 * NCE3_IMG + (RT * 132 + 4 * SL2 + 3 * RT) * sizeof(float);
    static RmclLdsOnce once3a, once3b;
    RMCL_TRY(rmcl_set_max_lds(once3a, reinterpret_cast<const void*>(infonce_partial3_kernel<true>), (int)lds3));
    RMCL_TRY(rmcl_set_max_lds(once3b, reinterpret_cast<const void*>(infonce_partial3_kernel<false>), (int)lds3));
    if (form == 2) RMCL_LAUNCH(infonce_partial3_kernel<false>, dim3(nparts, Bpad / RT), dim3(256), lds3, s, q, queue, Kq, B, 1.0f / T, part, dq_part, Bpad);
    else RMCL_LAUNCH(infonce_partial3_kernel<true>, dim3(nparts, Bpad / RT), dim3(256), lds3, s, q, queue, Kq, B, 1.0f / T, part, dq_part, Bpad);
  } else if (Kq % (SL2 * NS2) == 0 && g_infonce_fold >= 1) {           // 256 columns per workgroup, four folded 64-column sub-slices
    nparts = (int)(Kq / (SL2 * NS2));
    const size_t lds2 = (2 * PD * ILD2 + RT * ILD2 + 4 * SL2 + 3 * RT) * sizeof(float);
    static RmclLdsOnce once2;
    RMCL_TRY(rmcl_set_max_lds(once2, reinterpret_cast<const void*>(infonce_partial2_kernel), (int)lds2));
    RMCL_LAUNCH(infonce_partial2_kernel, dim3(nparts, Bpad / RT), dim3(256), lds2, s, q, queue, Kq, B, 1.0f / T, part, dq_part, Bpad);
  } else {
    RMCL_LAUNCH(infonce_partial_kernel, dim3(ns, Bpad / RT), dim3(256), lds, s, q, queue, Kq, B, 1.0f / T, part, dq_part, Bpad);
  }
  RMCL_CHECK_LAUNCH();
  if (g_infonce_fold >= 1)
    RMCL_LAUNCH(infonce_combine2_kernel, dim3(B, PD / CW), dim3(CMB_G * CW), 0, s, q, k, part, dq_part, nparts, B, Bpad, Kq, 1.0f / T, gscale, dq,
                rows_out, loss_sum);
  else
    RMCL_LAUNCH(infonce_combine_kernel, dim3(B), dim3(CMB_G * PD), 0, s, q, k, part, dq_part, nparts, B, Bpad, Kq, 1.0f / T, gscale, dq,
                rows_out, loss_sum);
  RMCL_CHECK_LAUNCH();
  return 0;
}
