// LayerNorm forward/backward, masked row softmax forward/backward, column sums.
// All are HBM-bound row kernels: one 64-lane wave per row, 16-byte vector accesses, f32 math.
#include "rmcl_common.h"
#include "kernels.h"

#define LN_MAXV 4  // float4 per lane -> D <= 1024

// wave sum by DPP row operations (all 64 lanes must be active); every lane receives lane 63's total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ln_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += ln_dpp<0xB1, 0xf>(v);     // quad_perm [1, 0, 3, 2]
  v += ln_dpp<0x4E, 0xf>(v);     // quad_perm [2, 3, 0, 1]
  v += ln_dpp<0x141, 0xf>(v);    // row_half_mirror
  v += ln_dpp<0x140, 0xf>(v);    // row_mirror: every lane of a 16-lane row holds the row's sum
  v += ln_dpp<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3
  v += ln_dpp<0x143, 0xc>(v);    // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * w + b  (optionally ReLU), x f32 [M,D], y TO [M,D]
// ---------------------------------------------------------------------------------------------
// NV = float4 per lane; FULL: D == 256 * NV (the encoder's 768: no column guards).  The weight / bias vectors are loaded with the row (they
// sat behind the two reductions: a second round trip per row) and the row sums are DPP row operations instead of twelve ds_bpermute.
template <typename TO, int NV, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                     const float* __restrict__ b, float eps, TO* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     int relu) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  float4 v[NV], ww[NV], bb[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = make_float4(0, 0, 0, 0);
    ww[i] = make_float4(0, 0, 0, 0);
    bb[i] = make_float4(0, 0, 0, 0);
    if (FULL || c < D) {
      v[i] = *reinterpret_cast<const float4*>(xr + c);
      ww[i] = *reinterpret_cast<const float4*>(w + c);
      bb[i] = *reinterpret_cast<const float4*>(b + c);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  const float mu = wave_sum_dpp(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (FULL || c < D) {
      const float a = v[i].x - mu, b2 = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
      q += a * a + b2 * b2 + cc * cc + d * d;
    }
  }
  const float rs = rsqrtf(wave_sum_dpp(q) / D + eps);
  if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
  TO* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (FULL || c < D) {
      float o[4] = {(v[i].x - mu) * rs * ww[i].x + bb[i].x, (v[i].y - mu) * rs * ww[i].y + bb[i].y,
                    (v[i].z - mu) * rs * ww[i].z + bb[i].z, (v[i].w - mu) * rs * ww[i].w + bb[i].w};
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
      }
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 pk;
        pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
        pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
        *reinterpret_cast<uint2*>(yr + c) = pk;
      }
    }
  }
}

int rmcl_ln_fwd(const float* x, long ldx, const float* w, const float* b, float eps, void* y, long ldy, int dt_out,
                float* mean, float* rstd, int M, int D, int relu, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0 && D <= 256 * LN_MAXV && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: D must be a multiple of 4 and <= 1024");
  if (M <= 0) return 0;
  dim3 grid(cdiv(M, 4));
#define LN_FWD_LAUNCH(TO)                                                                                                                      \
  do {                                                                                                                                         \
    if (D == 768) RMCL_LAUNCH((ln_fwd_kernel<TO, 3, true>), grid, dim3(256), 0, s, x, ldx, w, b, eps, (TO*)y, ldy, mean, rstd, M, D, relu);         \
    else RMCL_LAUNCH((ln_fwd_kernel<TO, LN_MAXV, false>), grid, dim3(256), 0, s, x, ldx, w, b, eps, (TO*)y, ldy, mean, rstd, M, D, relu);          \
  } while (0)
  if (dt_out == RMCL_F32) LN_FWD_LAUNCH(float);
  else LN_FWD_LAUNCH(bf16_t);
#undef LN_FWD_LAUNCH
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  dy TG [M,D] (grad of the LN output, before the optional ReLU mask),
//   dx_out = (add ? dx_out : 0) + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * w
//   dgamma += sum_rows dy*xhat, dbeta += sum_rows dy   (optional, fp32 atomics, one add per block)
// relu: dy is first masked with (xhat*w + b > 0).
// Rows handled per block: LNB_ROWS waves x LNB_ITERS rows, so dgamma/dbeta partials are reduced
// in registers over LNB_ITERS rows, then across the 4 waves through LDS, then one atomic per column.
// ---------------------------------------------------------------------------------------------
#define LN_REP 32
#define LN_REP_LD 1024
#define LNB_ITERS(WG) ((WG) ? 4 : 2)   // rows per wave: fewer -> more workgroups in flight (the data-gradient form has no block reduction to amortise)
// WG: accumulate dgamma/dbeta (weight-gradient backward only); RELU: mask dy with the ReLU after the LN (MoCo head).
// The data-gradient-only instantiation (PGD backward, 3/4 of all calls) carries 32 fewer accumulator registers.
// NV = float4 per lane (row width <= 256 * NV): 3 for the encoder width 768, LN_MAXV otherwise.
// Round 4: every load of a row - x, dy, the old dx when accumulating - is issued before the first one is consumed, and the two row sums
// are taken with DPP row operations.  Before, each of the three column chunks was its own guarded block (load x, load dy, wait, compute)
// and the accumulate form loaded the old value per chunk behind the reductions: six serial memory round trips per row, plus twelve
// ds_bpermute round trips for the two sums - the kernel ran at 3.5-4.4 TB/s with 6 % of its cycles issuing.  FULL: D == 256 * NV, no
// column guards (the encoder's 768).
template <typename TG, bool WG, bool RELU, int NV, bool FULL, int ADD = -1>   // ADD: 0 / 1 known at compile time (no branch around the old-value loads), -1 run time
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TG* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ dx, long lddx, int add, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int D, int relu,
                                                     void* __restrict__ dx_copy, int copy_f32, uint32_t dseed, uint32_t dthresh,
                                                     float dinv, float* __restrict__ rep) {
  __shared__ float red[WG ? 2 : 1][WG ? 4 : 1][WG ? 64 * 4 * NV : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gw[NV], gb[NV], ww[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gw[i] = make_float4(0, 0, 0, 0);
    gb[i] = make_float4(0, 0, 0, 0);
    ww[i] = make_float4(0, 0, 0, 0);
    bb[i] = make_float4(0, 0, 0, 0);
    const int c = (lane + 64 * i) * 4;
    if (FULL || c < D) {
      ww[i] = *reinterpret_cast<const float4*>(w + c);
      if (RELU) bb[i] = *reinterpret_cast<const float4*>(b + c);
    }
  }
  const bool addv = ADD < 0 ? add != 0 : ADD != 0;
  for (int it = 0; it < LNB_ITERS(WG); ++it) {
    const int row = (blockIdx.x * LNB_ITERS(WG) + it) * 4 + wave;
    if (row >= M) break;
    // ---- every load of the row, the old values (accumulate form) first.  As plain loads the compiler sank the old values below the two
    // reductions, to their use - a second, serial memory round trip per row.  They stay plain loads (a first version issued them by inline
    // assembly and handed the registers back behind the reductions: the register allocator is free to copy such a register before the data
    // has landed - one instantiation faulted); what keeps them up here is the empty assembly statement below, which "uses" every register
    // the loads of the row write: all of them are issued, in this order, before it, and nothing is sunk past it.
    typedef unsigned int ln_u32x2 __attribute__((ext_vector_type(2)));
    f32x4 oldr[NV], xr[NV], dyr[NV];
    ln_u32x2 dyb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      oldr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (addv && (FULL || c < D)) oldr[i] = *reinterpret_cast<const f32x4*>(dx + (long)row * lddx + c);
    }
    float mu = mean[row], rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      dyr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      dyb[i] = ln_u32x2{0u, 0u};
      if (FULL || c < D) {
        xr[i] = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + c);
        if constexpr (sizeof(TG) == 4) dyr[i] = *reinterpret_cast<const f32x4*>(dy + (long)row * lddy + c);
        else dyb[i] = *reinterpret_cast<const ln_u32x2*>(dy + (long)row * lddy + c);
      }
    }
    if constexpr (NV == 3) {
      if constexpr (sizeof(TG) == 4)
        asm volatile("" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(dyr[0]), "+v"(dyr[1]), "+v"(dyr[2]), "+v"(mu), "+v"(rs));
      else
        asm volatile("" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(dyb[0]), "+v"(dyb[1]), "+v"(dyb[2]), "+v"(mu), "+v"(rs));
    } else {
      static_assert(NV == 4, "ln_bwd: NV is 3 or 4");
      if constexpr (sizeof(TG) == 4)
        asm volatile("" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(oldr[3]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(dyr[0]), "+v"(dyr[1]),
                     "+v"(dyr[2]), "+v"(dyr[3]), "+v"(mu), "+v"(rs));
      else
        asm volatile("" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(oldr[3]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(dyb[0]), "+v"(dyb[1]),
                     "+v"(dyb[2]), "+v"(dyb[3]), "+v"(mu), "+v"(rs));
    }
    float4 xv[NV], dyf[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] = make_float4(xr[i].x, xr[i].y, xr[i].z, xr[i].w);
      dyf[i] = make_float4(dyr[i].x, dyr[i].y, dyr[i].z, dyr[i].w);
    }
    // ---- dy * w, the two row sums
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      xh[i] = make_float4(0, 0, 0, 0);
      g[i] = make_float4(0, 0, 0, 0);
      if (FULL || c < D) {
        float d[4];
        if constexpr (sizeof(TG) == 4) {
          d[0] = dyf[i].x; d[1] = dyf[i].y; d[2] = dyf[i].z; d[3] = dyf[i].w;
        } else {
          d[0] = __uint_as_float(dyb[i].x << 16); d[1] = __uint_as_float(dyb[i].x & 0xffff0000u);
          d[2] = __uint_as_float(dyb[i].y << 16); d[3] = __uint_as_float(dyb[i].y & 0xffff0000u);
        }
        xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
        if (RELU) {
          if (xh[i].x * ww[i].x + bb[i].x <= 0.f) d[0] = 0.f;
          if (xh[i].y * ww[i].y + bb[i].y <= 0.f) d[1] = 0.f;
          if (xh[i].z * ww[i].z + bb[i].z <= 0.f) d[2] = 0.f;
          if (xh[i].w * ww[i].w + bb[i].w <= 0.f) d[3] = 0.f;
        }
        if (WG) {
          gb[i].x += d[0]; gb[i].y += d[1]; gb[i].z += d[2]; gb[i].w += d[3];
          gw[i].x += d[0] * xh[i].x; gw[i].y += d[1] * xh[i].y; gw[i].z += d[2] * xh[i].z; gw[i].w += d[3] * xh[i].w;
        }
        g[i] = make_float4(d[0] * ww[i].x, d[1] * ww[i].y, d[2] * ww[i].z, d[3] * ww[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
    }
    s1 = wave_sum_dpp(s1) / D;
    s2 = wave_sum_dpp(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (FULL || c < D) {
        float4 o = make_float4(rs * (g[i].x - s1 - xh[i].x * s2), rs * (g[i].y - s1 - xh[i].y * s2),
                               rs * (g[i].z - s1 - xh[i].z * s2), rs * (g[i].w - s1 - xh[i].w * s2));
        float* p = dx + (long)row * lddx + c;
        if (addv) { o.x += oldr[i].x; o.y += oldr[i].y; o.z += oldr[i].z; o.w += oldr[i].w; }
        *reinterpret_cast<float4*>(p) = o;
        if (dx_copy) {   // copy of the updated residual-stream gradient = A operand of the next dX / dW GEMMs,
                         // already multiplied by the dropout mask of the branch output it flows into
          if (dthresh) {
            const uint32_t di = (uint32_t)((long)row * lddx + c);
            drop_scale4(dseed, di, dthresh, dinv, o.x, o.y, o.z, o.w);
          }
          if (copy_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dx_copy) + (long)row * lddx + c) = o;
          } else {
            uint2 pk;
            pk.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
            pk.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dx_copy) + (long)row * lddx + c) = pk;
          }
        }
      }
    }
  }
  if constexpr (WG) if (dgamma) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (FULL || c < D) {
        *reinterpret_cast<float4*>(&red[0][wave][c]) = gw[i];
        *reinterpret_cast<float4*>(&red[1][wave][c]) = gb[i];
      }
    }
    __syncthreads();
    // hundreds of blocks adding into the same 2 D addresses serialise in the memory-side atomic units: spread them
    // over LN_REP replicas (summed into dgamma/dbeta by ln_rep_finish_kernel) when the caller provides the buffer
    float* tg = rep ? rep + (blockIdx.x % LN_REP) * (2 * LN_REP_LD) : dgamma;
    float* tb = rep ? tg + LN_REP_LD : dbeta;
    for (int c = threadIdx.x; c < D; c += 256) {
      atomicAdd(tg + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
      atomicAdd(tb + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
  }
}

// dgamma/dbeta += sum of the replicas; leaves the replicas zeroed for the next call
__global__ __launch_bounds__(256) void ln_rep_finish_kernel(float* __restrict__ rep, float* __restrict__ dgamma, float* __restrict__ dbeta, int D) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float sg = 0.f, sb = 0.f;
#pragma unroll 8
  for (int r = 0; r < LN_REP; ++r) {
    float* p = rep + r * (2 * LN_REP_LD);
    sg += p[c]; sb += p[LN_REP_LD + c];
    p[c] = 0.f; p[LN_REP_LD + c] = 0.f;
  }
  dgamma[c] += sg;
  dbeta[c] += sb;
}

// replica buffer (one per process and device; the LayerNorm backward of the encoder runs on one stream)
static float* ln_rep_buffer() {
  static float* buf[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!buf[dev]) {
    float* p = nullptr;
    const size_t bytes = (size_t)LN_REP * 2 * LN_REP_LD * sizeof(float);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
    buf[dev] = p;
  }
  return buf[dev];
}

int rmcl_ln_bwd(const void* dy, long lddy, int dt_dy, const float* x, long ldx, const float* mean, const float* rstd,
                const float* w, const float* b, float* dx, long lddx, int add, float* dgamma, float* dbeta, int M, int D,
                int relu, hipStream_t s) {
  return rmcl_ln_bwd_lp(dy, lddy, dt_dy, x, ldx, mean, rstd, w, b, dx, lddx, add, dgamma, dbeta, M, D, relu, nullptr, RMCL_BF16, 0, 0,
                        1.0f, nullptr, s);
}

int rmcl_ln_bwd_lp(const void* dy, long lddy, int dt_dy, const float* x, long ldx, const float* mean, const float* rstd,
                   const float* w, const float* b, float* dx, long lddx, int add, float* dgamma, float* dbeta, int M, int D,
                   int relu, void* dx_copy, int copy_dt, uint32_t dseed, uint32_t dthresh, float dinv, float* rep_slot, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0 && D <= 256 * LN_MAXV && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "layernorm bwd: bad D/ld");
  if (M <= 0) return 0;
  const bool wg = dgamma != nullptr;
  dim3 grid(cdiv(M, 4 * LNB_ITERS(wg)));
  // rep_slot: a caller-owned, pre-zeroed replica region (RMCL_LN_REP_FLOATS floats) - dgamma/dbeta then stay in the replicas
  // and the CALLER sums them into the gradient arena later (the per-layer weight-gradient kernel does, off the critical path)
  RMCL_REQUIRE(!rep_slot || (wg && D <= LN_REP_LD), "layernorm bwd: replica slot needs dgamma and D <= 1024");
  float* rep = rep_slot ? rep_slot : ((wg && D <= LN_REP_LD && grid.x >= 4 * LN_REP) ? ln_rep_buffer() : nullptr);
#define LN_BWD_LAUNCH(TG, WGv, RLv) \
  do { if (D == 768 && add) RMCL_LAUNCH((ln_bwd_kernel<TG, WGv, RLv, 3, true, 1>), grid, dim3(256), 0, s, (const TG*)dy, lddy, x, ldx, mean, rstd, w, b, dx, lddx, add, dgamma, \
              dbeta, M, D, relu, dx_copy, copy_dt == RMCL_F32, dseed, dthresh, dinv, rep); \
       else if (D == 768) RMCL_LAUNCH((ln_bwd_kernel<TG, WGv, RLv, 3, true, 0>), grid, dim3(256), 0, s, (const TG*)dy, lddy, x, ldx, mean, rstd, w, b, dx, lddx, add, dgamma, \
              dbeta, M, D, relu, dx_copy, copy_dt == RMCL_F32, dseed, dthresh, dinv, rep); \
       else if (D < 768) RMCL_LAUNCH((ln_bwd_kernel<TG, WGv, RLv, 3, false>), grid, dim3(256), 0, s, (const TG*)dy, lddy, x, ldx, mean, rstd, w, b, dx, lddx, add, dgamma, \
              dbeta, M, D, relu, dx_copy, copy_dt == RMCL_F32, dseed, dthresh, dinv, rep); \
       else RMCL_LAUNCH((ln_bwd_kernel<TG, WGv, RLv, LN_MAXV, false>), grid, dim3(256), 0, s, (const TG*)dy, lddy, x, ldx, mean, rstd, w, b, dx, lddx, add, dgamma, \
              dbeta, M, D, relu, dx_copy, copy_dt == RMCL_F32, dseed, dthresh, dinv, rep); } while (0)
  if (dt_dy == RMCL_F32) {
    if (relu) { if (wg) LN_BWD_LAUNCH(float, true, true); else LN_BWD_LAUNCH(float, false, true); }
    else { if (wg) LN_BWD_LAUNCH(float, true, false); else LN_BWD_LAUNCH(float, false, false); }
  } else {
    if (relu) { if (wg) LN_BWD_LAUNCH(bf16_t, true, true); else LN_BWD_LAUNCH(bf16_t, false, true); }
    else { if (wg) LN_BWD_LAUNCH(bf16_t, true, false); else LN_BWD_LAUNCH(bf16_t, false, false); }
  }
#undef LN_BWD_LAUNCH
  RMCL_CHECK_LAUNCH();
  if (rep && !rep_slot) {
    RMCL_LAUNCH(ln_rep_finish_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, rep, dgamma, dbeta, D);
    RMCL_CHECK_LAUNCH();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Masked row softmax (attention scores).  S f32 [Z, N, lds] -> P T [Z, N, ldp]; key j of batch
// b = z / H is dropped (exact 0 probability, like masked_fill(-inf)) when mask[b*N + j] == 0.
// Pad columns N..ldp-1 are written as zeros so that P can be a GEMM operand with K = N.
// ---------------------------------------------------------------------------------------------
#define SM_MAXE 8  // elements per lane -> N <= 512
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, long lds, const int* __restrict__ mask,
                                                          T* __restrict__ P, long ldp, int rows, int N, int H) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int b = (int)(row / ((long)N * H));
  const float* sr = S + row * lds;
  const int* mr = mask + (long)b * N;
  float v[SM_MAXE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int j = lane + 64 * i;
    v[i] = (j < N && mr[j] != 0) ? sr[j] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    v[i] = (v[i] == -INFINITY) ? 0.f : __expf(v[i] - mx);
    sum += v[i];
  }
  const float inv = 1.0f / wave_sum(sum);
  T* pr = P + row * ldp;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int j = lane + 64 * i;
    if (j < ldp) pr[j] = from_f32<T>(j < N ? v[i] * inv : 0.f);
  }
}

int rmcl_softmax_fwd(const float* S, long lds, const int* mask, void* P, long ldp, int dt, int Z, int N, int H, hipStream_t s) {
  RMCL_REQUIRE(N <= 64 * SM_MAXE && ldp <= 64 * SM_MAXE, "softmax: N too large");
  const long rows = (long)Z * N;
  dim3 grid(cdiv(rows, 4));
  if (dt == RMCL_F32) RMCL_LAUNCH(softmax_fwd_kernel<float>, grid, dim3(256), 0, s, S, lds, mask, (float*)P, ldp, (int)rows, N, H);
  else RMCL_LAUNCH(softmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, S, lds, mask, (bf16_t*)P, ldp, (int)rows, N, H);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// dS = scale * P * (dP - sum_j dP_j P_j)   (P T, dP f32, dS T; pads written as zero)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, long ldp, const float* __restrict__ dP, long lddp,
                                                          T* __restrict__ dS, long ldds, int rows, int N, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float p[SM_MAXE], d[SM_MAXE];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int j = lane + 64 * i;
    p[i] = j < N ? to_f32<T>(P[row * ldp + j]) : 0.f;
    d[i] = j < N ? dP[row * lddp + j] : 0.f;
    dot += p[i] * d[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int j = lane + 64 * i;
    if (j < ldds) dS[row * ldds + j] = from_f32<T>(j < N ? scale * p[i] * (d[i] - dot) : 0.f);
  }
}

int rmcl_softmax_bwd(const void* P, long ldp, const float* dP, long lddp, void* dS, long ldds, int dt, int Z, int N,
                     float scale, hipStream_t s) {
  RMCL_REQUIRE(N <= 64 * SM_MAXE && ldds <= 64 * SM_MAXE, "softmax bwd: N too large");
  const long rows = (long)Z * N;
  dim3 grid(cdiv(rows, 4));
  if (dt == RMCL_F32) RMCL_LAUNCH(softmax_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)P, ldp, dP, lddp, (float*)dS, ldds, (int)rows, N, scale);
  else RMCL_LAUNCH(softmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)P, ldp, dP, lddp, (bf16_t*)dS, ldds, (int)rows, N, scale);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Column sums (bias gradients): out[n] += sum_m X[m*ld + n].  Block = 64 columns x 4 row-lanes.
// ---------------------------------------------------------------------------------------------
#define CS_ROWS 128
// column sums (bias gradients): a lane owns 4 adjacent columns (8- or 16-byte loads, 512 B - 1 KiB per wave-row), a wave
// walks every 4th row of a 128-row slab; the four waves combine through LDS and issue one f32 atomic per column and slab.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ld, float* __restrict__ out, int M, int N) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      if constexpr (sizeof(T) == 2) {
        const uint2 u = *reinterpret_cast<const uint2*>(X + (long)r * ld + c);
        acc.x += __uint_as_float(u.x << 16); acc.y += __uint_as_float(u.x & 0xffff0000u);
        acc.z += __uint_as_float(u.y << 16); acc.w += __uint_as_float(u.y & 0xffff0000u);
      } else {
        const float4 u = *reinterpret_cast<const float4*>(X + (long)r * ld + c);
        acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
      }
    }
  }
  red[rl][lane] = acc;
  __syncthreads();
  if (rl == 0 && c < N) {
    const float4 a = red[0][lane], b = red[1][lane], d = red[2][lane], e = red[3][lane];
    atomicAdd(out + c, a.x + b.x + d.x + e.x);
    atomicAdd(out + c + 1, a.y + b.y + d.y + e.y);
    atomicAdd(out + c + 2, a.z + b.z + d.z + e.z);
    atomicAdd(out + c + 3, a.w + b.w + d.w + e.w);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_any_kernel(const T* __restrict__ X, long ld, float* __restrict__ out, int M, int N) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float acc = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) acc += to_f32<T>(X[(long)r * ld + c]);
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int rmcl_colsum(const void* X, long ld, int dt, float* out, int M, int N, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  const int esz = dt == RMCL_F32 ? 4 : 2;
  if (N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X % (4 * esz)) == 0) {
    dim3 grid(cdiv(N, 256), cdiv(M, CS_ROWS));
    if (dt == RMCL_F32) RMCL_LAUNCH(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, ld, out, M, N);
    else RMCL_LAUNCH(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, ld, out, M, N);
  } else {
    dim3 grid(cdiv(N, 64), cdiv(M, CS_ROWS));
    if (dt == RMCL_F32) RMCL_LAUNCH(colsum_any_kernel<float>, grid, dim3(256), 0, s, (const float*)X, ld, out, M, N);
    else RMCL_LAUNCH(colsum_any_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, ld, out, M, N);
  }
  RMCL_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// LayerNorm fold (gemm.h EPI_LNFOLD): y = LN(x) W^T + b = rstd * (x W'^T - mean * s) + c with
//   W'[n][k] = bf16(W[n][k] * gamma[k]),  s[n] = sum_k W'[n][k] (of the ROUNDED values: what the MFMA multiplies by),
//   c[n] = sum_k W[n][k] * beta[k] + b[n].
// One workgroup per output row n of one (layer, matrix) pair; wf: [layers][3D + mlp][D] bf16 (qkv rows, then fc1 rows),
// sc: [layers][2][3D + mlp] f32 (s, then c).  Runs after every change of the fp32 masters (optimizer step, momentum update).
// ---------------------------------------------------------------------------------------------
// one wave per output row (16 rows per workgroup), float4 loads, 8-byte bf16 stores
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ p32, long layer0, long stride, long ln1_w, long ln1_b, long qkv_w,
                                                      long qkv_b, long ln2_w, long ln2_b, long fc1_w, long fc1_b, int D, int mlp, int total_rows,
                                                      bf16_t* __restrict__ wf, float* __restrict__ sc) {
  const int rows = 3 * D + mlp, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int gr = blockIdx.x * 16 + wave * 4 + i;
    if (gr >= total_rows) return;
    const int l = gr / rows, r = gr - l * rows;
    const float* base = p32 + layer0 + (long)l * stride;
    const bool is_qkv = r < 3 * D;
    const int n = is_qkv ? r : r - 3 * D;
    const float* W = base + (is_qkv ? qkv_w : fc1_w) + (long)n * D;
    const float* gam = base + (is_qkv ? ln1_w : ln2_w);
    const float* bet = base + (is_qkv ? ln1_b : ln2_b);
    bf16_t* o = wf + ((long)l * rows + r) * D;
    float ssum = 0.f, csum = 0.f;
    for (int k = lane * 4; k < D; k += 256) {
      const float4 w = *reinterpret_cast<const float4*>(W + k), g4 = *reinterpret_cast<const float4*>(gam + k),
                   b4 = *reinterpret_cast<const float4*>(bet + k);
      const bf16_t q0 = f2bf(w.x * g4.x), q1 = f2bf(w.y * g4.y), q2 = f2bf(w.z * g4.z), q3 = f2bf(w.w * g4.w);
      uint2 pk;
      pk.x = (uint32_t)q0 | ((uint32_t)q1 << 16);
      pk.y = (uint32_t)q2 | ((uint32_t)q3 << 16);
      *reinterpret_cast<uint2*>(o + k) = pk;
      ssum += (bf2f(q0) + bf2f(q1)) + (bf2f(q2) + bf2f(q3));
      csum += (w.x * b4.x + w.y * b4.y) + (w.z * b4.z + w.w * b4.w);
    }
    ssum = wave_sum(ssum);
    csum = wave_sum(csum);
    if (lane == 0) {
      sc[(long)l * 2 * rows + r] = ssum;
      sc[(long)l * 2 * rows + rows + r] = csum + base[(is_qkv ? qkv_b : fc1_b) + n];
    }
  }
}
int rmcl_ln_fold_launch(const float* p32, long layer0, long stride, int layers, long ln1_w, long ln1_b, long qkv_w, long qkv_b, long ln2_w,
                        long ln2_b, long fc1_w, long fc1_b, int D, int mlp, unsigned short* wf, float* sc, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0, "ln_fold: D%4");
  const int total = layers * (3 * D + mlp);
  RMCL_LAUNCH(ln_fold_kernel, dim3(cdiv(total, 16)), dim3(256), 0, s, p32, layer0, stride, ln1_w, ln1_b, qkv_w, qkv_b, ln2_w, ln2_b, fc1_w,
              fc1_b, D, mlp, total, wf, sc);
  RMCL_CHECK_LAUNCH();
  return 0;
}
