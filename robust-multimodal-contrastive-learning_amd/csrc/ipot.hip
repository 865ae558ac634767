// Word-patch-alignment helpers for the ITM objective (BASELINE configs 1-2):
//   * IPOT (objectives.py:46-76): 50 proximal-point iterations, ONE workgroup per sample with all
//     state (A, T, sigma, delta) resident in LDS; replaces ~300 tiny torch launches per step.
//   * masked cosine cost post-processing and the transport-plan -> dcost scatter.
#include "rmcl_common.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void ipot_kernel(const float* __restrict__ cost, const int* __restrict__ txt_valid,
                                                   const int* __restrict__ img_valid, float* __restrict__ Tout, int Lt, int Li, int ldc,
                                                   float beta, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = Lt + 1;
  float* A = sm;                 // [Li][ld]
  float* T = A + Li * ld;        // [Li][ld]
  float* sigma = T + Li * ld;    // [Lt]
  float* delta = sigma + Lt;     // [Li]
  float* xmask = delta + Li;     // [Lt]
  float* ymask = xmask + Lt;     // [Li]
  __shared__ float lens[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* C = cost + (long)b * Lt * ldc;
  const int* tv = txt_valid + (long)b * Lt;
  const int* iv = img_valid + (long)b * Li;
  if (t == 0) {
    int xl = 0, yl = 0;
    for (int m = 0; m < Lt; ++m) xl += tv[m] != 0;
    for (int n = 0; n < Li; ++n) yl += iv[n] != 0;
    lens[0] = (float)xl; lens[1] = (float)yl;
  }
  __syncthreads();
  const float x_len = lens[0], y_len = lens[1];
  for (int m = t; m < Lt; m += 256) { sigma[m] = tv[m] ? 1.0f / x_len : 0.f; xmask[m] = tv[m] ? 0.f : 1e4f; }
  for (int n = t; n < Li; n += 256) ymask[n] = iv[n] ? 0.f : 1e4f;
  for (int i = t; i < Li * Lt; i += 256) {
    const int n = i / Lt, m = i % Lt;
    const bool pad = !(tv[m] && iv[n]);
    A[n * ld + m] = pad ? 0.f : expf(-C[(long)m * ldc + n] / beta);
    T[n * ld + m] = pad ? 0.f : 1.f;
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int n = t; n < Li; n += 256) {          // delta = 1 / (y_len * (Q sigma) + y_mask)
      float s = 0.f;
      for (int m = 0; m < Lt; ++m) s += A[n * ld + m] * T[n * ld + m] * sigma[m];
      delta[n] = 1.0f / (y_len * s + ymask[n]);
    }
    __syncthreads();
    for (int m = t; m < Lt; m += 256) {          // sigma = 1 / (x_len * (delta Q) + x_mask)
      float s = 0.f;
      for (int n = 0; n < Li; ++n) s += delta[n] * A[n * ld + m] * T[n * ld + m];
      sigma[m] = 1.0f / (x_len * s + xmask[m]);
    }
    __syncthreads();
    for (int i = t; i < Li * Lt; i += 256) {     // T = delta * Q * sigma
      const int n = i / Lt, m = i % Lt;
      T[n * ld + m] = delta[n] * A[n * ld + m] * T[n * ld + m] * sigma[m];
    }
    __syncthreads();
  }
  for (int i = t; i < Li * Lt; i += 256) {
    const int n = i / Lt, m = i % Lt;
    Tout[(long)b * Li * Lt + i] = (tv[m] && iv[n]) ? T[n * ld + m] : 0.f;
  }
}

int rmcl_ipot(const float* cost, const int* txt_valid, const int* img_valid, float* T, int B, int Lt, int Li, int ld, float beta,
              int iters, hipStream_t s) {
  const size_t lds = ((size_t)2 * Li * (Lt + 1) + 2 * Lt + 2 * Li) * sizeof(float);
  RMCL_REQUIRE(lds <= 150 * 1024, "ipot: problem too large for LDS");
  static RmclLdsOnce once;                                     // (raised again when a larger problem arrives)
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(ipot_kernel), (int)lds));
  RMCL_LAUNCH(ipot_kernel, dim3(B), dim3(256), lds, s, cost, txt_valid, img_valid, T, Lt, Li, ld, beta, iters);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// cost[b,m,n] = masked ? 0 : 1 - dots[b,m,n]   (dots = cosine similarities, in place)
__global__ __launch_bounds__(256) void cost_finish_kernel(float* __restrict__ cost, const int* __restrict__ txt_valid,
                                                          const int* __restrict__ img_valid, int B, int Lt, int Li, int ld) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * Lt * ld) return;
  const int n = (int)(i % ld), m = (int)((i / ld) % Lt), b = (int)(i / ((long)ld * Lt));
  const bool ok = n < Li && txt_valid[(long)b * Lt + m] && img_valid[(long)b * Li + n];
  cost[i] = ok ? 1.0f - cost[i] : 0.f;
}
int rmcl_cost_finish(float* cost, const int* txt_valid, const int* img_valid, int B, int Lt, int Li, int ld, hipStream_t s) {
  RMCL_LAUNCH(cost_finish_kernel, dim3(cdiv((long)B * Lt * ld, 256)), dim3(256), 0, s, cost, txt_valid, img_valid, B, Lt, Li, ld);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// dist[b] = sum_{m,n} cost[b,m,n] * T[b,n,m];  dsim[b,m,n] = -w[b] * T[b,n,m]  (d loss / d cosine-sim)
__global__ __launch_bounds__(256) void wpa_dist_kernel(const float* __restrict__ cost, const float* __restrict__ T,
                                                       const float* __restrict__ w, float* __restrict__ dist, float* __restrict__ dsim,
                                                       int Lt, int Li, int ld) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  for (int i = t; i < Lt * ld; i += 256) {
    const int m = i / ld, n = i % ld;
    const float tv = n < Li ? T[(long)b * Li * Lt + (long)n * Lt + m] : 0.f;
    acc += cost[(long)b * Lt * ld + i] * tv;
    if (dsim) dsim[(long)b * Lt * ld + i] = -w[b] * tv;
  }
  acc = wave_sum(acc);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t == 0) dist[b] = red[0] + red[1] + red[2] + red[3];
}
int rmcl_wpa_dist(const float* cost, const float* T, const float* w, float* dist, float* dsim, int B, int Lt, int Li, int ld,
                  hipStream_t s) {
  RMCL_LAUNCH(wpa_dist_kernel, dim3(B), dim3(256), 0, s, cost, T, w, dist, dsim, Lt, Li, ld);
  RMCL_CHECK_LAUNCH();
  return 0;
}
