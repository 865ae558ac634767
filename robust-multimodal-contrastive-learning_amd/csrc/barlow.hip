// Barlow-Twins variant of the contrastive step: BarlowTwinsHead (vilt/modules/heads.py:88-107:
// Linear(D,H1,no bias) - BatchNorm1d - ReLU - Linear(H1,H2) - BatchNorm1d - ReLU - Linear(H2,H3), then BatchNorm1d(affine=False))
// and the cross-correlation loss of compute_barlowtwins_contrastive (vilt/modules/objectives.py:449-602).
//
// Shapes: the batch is the SHORT dimension (B = 64 rows) and the features are wide (8192), so
//   * the three linears and their data gradients are weight-streaming skinny GEMMs (exact fp32, gemm_exact.hip),
//   * the weight gradients and the cross-correlation c = zq^T zk / bs are outer-product-like GEMMs with K = B,
//   * BatchNorm works over the batch rows of one column: 64 columns x 4 row groups per workgroup (coalesced across threads).
// Everything is fp32 like the other heads (pooler / MoCo head): 3 x 64 x 8192 activations are small, the 8192 x 8192 weights
// and the correlation matrix are read once per pass from HBM.
#include <algorithm>
#include "rmcl_common.h"
#include "kernels.h"
#include "../../include/rmcl.h"

namespace {
inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// BatchNorm over the batch rows of a column.  Workgroup = 64 columns x 4 row groups (thread t: column t & 63, rows t >> 6, +4, ...):
// a wave reads 64 consecutive floats of one row (256-byte segments), the four partial sums of a column meet in LDS.  One thread
// per column with all B rows serial (the first version) left the chip at 32 workgroups and 46 us per call for 2 MB.
//
// y[b, n] = relu?( gamma[n] * (x[b,n] - mean[n]) * rstd[n] + beta[n] ); training: batch statistics over the B rows (biased variance
// for the normalisation, unbiased for the running estimate - torch.nn.BatchNorm1d), eval: the running statistics.
__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, int B, int N, int relu,
                                                     int training) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const bool live = n < N;
  float mean = 0.f, var = 1.f;
  if (training) {
    float s = 0.f;
    if (live)
      for (int b = rg; b < B; b += 4) s += x[(long)b * N + n];
    red[rg][c] = s;
    __syncthreads();
    mean = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / (float)B;
    __syncthreads();
    float q = 0.f;
    if (live)
      for (int b = rg; b < B; b += 4) {
        const float d = x[(long)b * N + n] - mean;
        q = fmaf(d, d, q);
      }
    red[rg][c] = q;
    __syncthreads();
    const float qs = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    var = qs / (float)B;
    if (run_mean && live && rg == 0) {
      run_mean[n] = (1.f - momentum) * run_mean[n] + momentum * mean;
      run_var[n] = (1.f - momentum) * run_var[n] + momentum * (B > 1 ? qs / (float)(B - 1) : var);
    }
  } else if (live) {
    mean = run_mean[n];
    var = run_var[n];
  }
  if (!live) return;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (rg == 0) {
    mean_out[n] = mean;
    rstd_out[n] = rstd;
  }
  const float g = gamma ? gamma[n] : 1.f, bt = beta ? beta[n] : 0.f;
  for (int b = rg; b < B; b += 4) {
    float v = (x[(long)b * N + n] - mean) * rstd * g + bt;
    if (relu) v = fmaxf(v, 0.f);
    y[(long)b * N + n] = v;
  }
}

// dx through (ReLU after) BatchNorm.  training: dx = g rstd / B (B dy - sum dy - xhat sum(dy xhat)); eval: dx = g rstd dy.
// y: the forward output (ReLU mask: y > 0), NULL without ReLU.  dgamma / dbeta accumulate (+=), NULL = not wanted.
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int N,
                                                     int training) {
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const bool live = n < N;
  const float mu = live ? mean[n] : 0.f, rs = live ? rstd[n] : 0.f, g = (gamma && live) ? gamma[n] : 1.f;
  float s1 = 0.f, s2 = 0.f;
  if (live)
    for (int b = rg; b < B; b += 4) {
      const long i = (long)b * N + n;
      float d = dy[i];
      if (y && !(y[i] > 0.f)) d = 0.f;
      s1 += d;
      s2 = fmaf(d, (x[i] - mu) * rs, s2);
    }
  red[0][rg][c] = s1;
  red[1][rg][c] = s2;
  __syncthreads();
  s1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  s2 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  if (!live) return;
  if (rg == 0) {
    if (dgamma) dgamma[n] += s2;
    if (dbeta) dbeta[n] += s1;
  }
  const float invB = 1.f / (float)B;
  for (int b = rg; b < B; b += 4) {
    const long i = (long)b * N + n;
    float d = dy[i];
    if (y && !(y[i] > 0.f)) d = 0.f;
    const float xh = (x[i] - mu) * rs;
    dx[i] = training ? g * rs * (d - invB * s1 - xh * invB * s2) : g * rs * d;
  }
}

// In place over the N x N cross-correlation: c <- scale * 2 w (c - I), w = 1 on the diagonal, lambda off it (the gradient of
// on_diag + lambda off_diag, objectives.py:481-484); per-workgroup partial sums of (c_ii - 1)^2 and of c_ij^2 (i != j) in
// `part` [grid][2], summed in a fixed order by barlow_loss_finish (no float atomics: the loss is bit-reproducible).
__global__ __launch_bounds__(256) void barlow_loss_kernel(float* __restrict__ c, int N, float lambda, float scale, float* __restrict__ part) {
  const long total4 = (long)N * N / 4;
  float on = 0.f, off = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    float4 v = reinterpret_cast<float4*>(c)[i];
    const long e = i * 4;
    const int row = (int)(e / N), col = (int)(e - (long)row * N);
    float* p = &v.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (col + j == row) {
        const float d = p[j] - 1.f;
        on = fmaf(d, d, on);
        p[j] = scale * 2.f * d;
      } else {
        off = fmaf(p[j], p[j], off);
        p[j] = scale * 2.f * lambda * p[j];
      }
    }
    reinterpret_cast<float4*>(c)[i] = v;
  }
  __shared__ float red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    on += __shfl_down(on, o);
    off += __shfl_down(off, o);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = on; red[1][threadIdx.x >> 6] = off; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
__global__ __launch_bounds__(256) void barlow_loss_finish(const float* __restrict__ part, int nparts, float* __restrict__ out2) {
  __shared__ double red[2][256];
  double on = 0.0, off = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { on += part[2 * i]; off += part[2 * i + 1]; }
  red[0][threadIdx.x] = on; red[1][threadIdx.x] = off;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = (float)red[0][0]; out2[1] = (float)red[1][0]; }
}

// rows[b] = (||q_b - k_b||_2, cosine(q_b, k_b) with eps 1e-6 as nn.CosineSimilarity, q_b . k_b): objectives.py:496-498
__global__ __launch_bounds__(256) void pair_metrics_kernel(const float* __restrict__ q, const float* __restrict__ k, int N, float* __restrict__ rows) {
  const int b = blockIdx.x;
  float dd = 0.f, qq = 0.f, kk = 0.f, qk = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float a = q[(long)b * N + n], c = k[(long)b * N + n];
    dd = fmaf(a - c, a - c, dd); qq = fmaf(a, a, qq); kk = fmaf(c, c, kk); qk = fmaf(a, c, qk);
  }
  __shared__ float red[4][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dd += __shfl_down(dd, o); qq += __shfl_down(qq, o); kk += __shfl_down(kk, o); qk += __shfl_down(qk, o);
  }
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = dd; red[1][w] = qq; red[2][w] = kk; red[3][w] = qk; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s[4];
    for (int i = 0; i < 4; ++i) s[i] = (red[i][0] + red[i][1]) + (red[i][2] + red[i][3]);
    rows[3 * b] = sqrtf(s[0]);
    rows[3 * b + 1] = s[3] / (fmaxf(sqrtf(s[1]), 1e-6f) * fmaxf(sqrtf(s[2]), 1e-6f));
    rows[3 * b + 2] = s[3];
  }
}

GemmArgs mk(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, int ldc) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.f; g.splitk = 1; g.nb1 = 1; g.nb2 = 1;
  return g;
}

struct BtStash {                     // per pass, all fp32 (rmcl_bt_stash_floats)
  float *x0, *h1, *a1, *h2, *a2, *h3, *stat, *t0, *t1;     // stat: mean1 rstd1 mean2 rstd2 mean3 rstd3
};
long carve(const rmcl_bt_head& h, int B, float* base, BtStash* s) {
  long o = 0;
  auto take = [&](long n) { float* p = base ? base + o : nullptr; o += (n + 63) / 64 * 64; return p; };
  const int Hm = std::max(std::max(h.H1, h.H2), std::max(h.H3, h.D));
  s->x0 = take((long)B * h.D);
  s->h1 = take((long)B * h.H1); s->a1 = take((long)B * h.H1);
  s->h2 = take((long)B * h.H2); s->a2 = take((long)B * h.H2);
  s->h3 = take((long)B * h.H3);
  s->stat = take(2L * (h.H1 + h.H2 + h.H3));
  s->t0 = take((long)B * Hm); s->t1 = take((long)B * Hm);
  return o;
}
int bn_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, float* rm, float* rv, float mom, int B, int N,
           int relu, int training, hipStream_t s) {
  RMCL_LAUNCH(bn_fwd_kernel, dim3((N + 63) / 64), dim3(256), 0, s, x, g, b, 1e-5f, y, mean, rstd, rm, rv, mom, B, N, relu, training);
  RMCL_CHECK_LAUNCH();
  return 0;
}
int bn_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, const float* g, float* dx, float* dg, float* db,
           int B, int N, int training, hipStream_t s) {
  RMCL_LAUNCH(bn_bwd_kernel, dim3((N + 63) / 64), dim3(256), 0, s, dy, x, y, mean, rstd, g, dx, dg, db, B, N, training);
  RMCL_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" {

int64_t rmcl_bt_stash_floats(const rmcl_bt_head* h, int B) {
  BtStash s;
  return carve(*h, B, nullptr, &s);
}

int rmcl_bt_head_forward(const rmcl_bt_head* h, const float* params, const float* cls_feats, int B, int training, float* running, float momentum,
                         float* stash, float* z, void* stream) {
  RMCL_REQUIRE(h && params && cls_feats && stash && z && B >= 1, "bt_head_forward: NULL argument");
  RMCL_REQUIRE(training || running, "bt_head_forward: eval mode needs the running statistics");
  RMCL_REQUIRE(h->D % 16 == 0 && h->H1 % 16 == 0 && h->H2 % 16 == 0 && h->H3 % 16 == 0, "bt_head_forward: widths must be multiples of 16");
  hipStream_t s = (hipStream_t)stream;
  BtStash st;
  carve(*h, B, stash, &st);
  float* rm[3] = {nullptr, nullptr, nullptr};
  float* rv[3] = {nullptr, nullptr, nullptr};
  if (running) {
    rm[0] = running; rv[0] = rm[0] + h->H1;
    rm[1] = rv[0] + h->H1; rv[1] = rm[1] + h->H2;
    rm[2] = rv[1] + h->H2; rv[2] = rm[2] + h->H3;
  }
  float* mean1 = st.stat, *rstd1 = mean1 + h->H1, *mean2 = rstd1 + h->H1, *rstd2 = mean2 + h->H2, *mean3 = rstd2 + h->H2, *rstd3 = mean3 + h->H3;
  hipError_t e = hipMemcpyAsync(st.x0, cls_feats, (size_t)B * h->D * 4, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.x0, params + h->w1, st.h1, B, h->H1, h->D, h->D, h->D, h->H1), RMCL_F32, RMCL_F32, 1, 1, s));
  RMCL_TRY(bn_fwd(st.h1, params + h->g1, params + h->b1, st.a1, mean1, rstd1, rm[0], rv[0], momentum, B, h->H1, 1, training, s));
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.a1, params + h->w2, st.h2, B, h->H2, h->H1, h->H1, h->H1, h->H2), RMCL_F32, RMCL_F32, 1, 1, s));
  RMCL_TRY(bn_fwd(st.h2, params + h->g2, params + h->b2, st.a2, mean2, rstd2, rm[1], rv[1], momentum, B, h->H2, 1, training, s));
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.a2, params + h->w3, st.h3, B, h->H3, h->H2, h->H2, h->H2, h->H3), RMCL_F32, RMCL_F32, 1, 1, s));
  RMCL_TRY(bn_fwd(st.h3, nullptr, nullptr, z, mean3, rstd3, rm[2], rv[2], momentum, B, h->H3, 0, training, s));
  return 0;
}

int rmcl_bt_head_backward(const rmcl_bt_head* h, const float* params, float* stash, const float* dz, int B, int training, float* G, float* dcls,
                          void* stream) {
  RMCL_REQUIRE(h && params && stash && dz && dcls, "bt_head_backward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  BtStash st;
  carve(*h, B, stash, &st);
  float* mean1 = st.stat, *rstd1 = mean1 + h->H1, *mean2 = rstd1 + h->H1, *rstd2 = mean2 + h->H2, *mean3 = rstd2 + h->H2, *rstd3 = mean3 + h->H3;
  // norm (affine=False): dh3
  RMCL_TRY(bn_bwd(dz, st.h3, nullptr, mean3, rstd3, nullptr, st.t0, nullptr, nullptr, B, h->H3, training, s));
  if (G) {                                                                   // dW3 += dh3^T a2
    GemmArgs g = mk(st.t0, st.a2, G + h->w3, h->H3, h->H2, B, h->H3, h->H2, h->H2);
    g.epi = EPI_ACCUM;
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
  }
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.t0, params + h->w3, st.t1, B, h->H2, h->H3, h->H3, h->H2, h->H2), RMCL_F32, RMCL_F32, 1, 0, s));   // da2
  RMCL_TRY(bn_bwd(st.t1, st.h2, st.a2, mean2, rstd2, params + h->g2, st.t0, G ? G + h->g2 : nullptr, G ? G + h->b2 : nullptr, B, h->H2, training, s));
  if (G) {
    GemmArgs g = mk(st.t0, st.a1, G + h->w2, h->H2, h->H1, B, h->H2, h->H1, h->H1);
    g.epi = EPI_ACCUM;
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
  }
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.t0, params + h->w2, st.t1, B, h->H1, h->H2, h->H2, h->H1, h->H1), RMCL_F32, RMCL_F32, 1, 0, s));   // da1
  RMCL_TRY(bn_bwd(st.t1, st.h1, st.a1, mean1, rstd1, params + h->g1, st.t0, G ? G + h->g1 : nullptr, G ? G + h->b1 : nullptr, B, h->H1, training, s));
  if (G) {
    GemmArgs g = mk(st.t0, st.x0, G + h->w1, h->H1, h->D, B, h->H1, h->D, h->D);
    g.epi = EPI_ACCUM;
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
  }
  RMCL_TRY(rmcl_launch_gemm_exact(mk(st.t0, params + h->w1, dcls, B, h->D, h->H1, h->H1, h->D, h->D), RMCL_F32, RMCL_F32, 1, 0, s));       // dcls
  return 0;
}

int rmcl_bt_corr(const float* zq, const float* zk, int B, int N, float inv_bs, float* c, void* stream) {
  RMCL_REQUIRE(zq && zk && c && N % 4 == 0, "bt_corr: NULL argument / N % 4");
  GemmArgs g = mk(zq, zk, c, N, N, B, N, N, N);                               // c = zq^T zk (objectives.py:478)
  g.alpha = inv_bs;
  return rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, (hipStream_t)stream);
}

int64_t rmcl_bt_loss_ws_floats(int N) { return 2 * std::min<long>(cdivl((long)N * N / 4, 256), 4096); }

int rmcl_bt_loss(float* c, int N, float lambda, float grad_scale, float* ws, float* loss2, void* stream) {
  RMCL_REQUIRE(c && ws && loss2 && N % 4 == 0, "bt_loss: NULL argument / N % 4");
  hipStream_t s = (hipStream_t)stream;
  const int grid = (int)std::min<long>(cdivl((long)N * N / 4, 256), 4096);
  RMCL_LAUNCH(barlow_loss_kernel, dim3(grid), dim3(256), 0, s, c, N, lambda, grad_scale, ws);
  RMCL_CHECK_LAUNCH();
  RMCL_LAUNCH(barlow_loss_finish, dim3(1), dim3(256), 0, s, ws, grid, loss2);
  RMCL_CHECK_LAUNCH();
  return 0;
}

int rmcl_bt_dz(const float* zk, const float* G, int B, int N, float inv_bs, float* dzq, void* stream) {
  RMCL_REQUIRE(zk && G && dzq, "bt_dz: NULL argument");
  GemmArgs g = mk(zk, G, dzq, B, N, N, N, N, N);                              // dzq[b,i] = sum_j G[i,j] zk[b,j] / bs
  g.alpha = inv_bs;
  return rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, (hipStream_t)stream);
}

int rmcl_bt_pair_metrics(const float* q, const float* k, int B, int N, float* rows, void* stream) {
  RMCL_REQUIRE(q && k && rows, "bt_pair_metrics: NULL argument");
  RMCL_LAUNCH(pair_metrics_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, k, N, rows);
  RMCL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
