// ITM head (vilt/modules/heads.py:173-180) + 2-class cross-entropy (objectives.py:764-765): tiny kernels.
#include "rmcl_common.h"
#include "kernels.h"

// one wave per sample: logits = cls W^T + b; loss_sum += CE/B; dlogits = gscale * (softmax - onehot)
__global__ __launch_bounds__(64) void itm_head_fwd_kernel(const float* __restrict__ cls, const float* __restrict__ W,
                                                          const float* __restrict__ bias, const int* __restrict__ labels,
                                                          float* __restrict__ logits, float* __restrict__ dlogits,
                                                          float* __restrict__ loss_sum, int B, int D, float gscale) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float x = cls[(long)b * D + c];
    s0 += x * W[c];
    s1 += x * W[D + c];
  }
  s0 = wave_sum(s0) + bias[0];
  s1 = wave_sum(s1) + bias[1];
  if (lane == 0) {
    logits[2 * b] = s0; logits[2 * b + 1] = s1;
    const float m = fmaxf(s0, s1), e0 = __expf(s0 - m), e1 = __expf(s1 - m), z = e0 + e1;
    const int y = labels[b];
    if (loss_sum) atomicAdd(loss_sum, (m + logf(z) - (y ? s1 : s0)) / (float)B);
    if (dlogits) {
      dlogits[2 * b] = gscale * (e0 / z - (y == 0 ? 1.f : 0.f));
      dlogits[2 * b + 1] = gscale * (e1 / z - (y == 1 ? 1.f : 0.f));
    }
  }
}
int rmcl_itm_head_fwd(const float* cls, const float* W, const float* bias, const int* labels, float* logits, float* dlogits,
                      float* loss_sum, int B, int D, float gscale, hipStream_t s) {
  RMCL_LAUNCH(itm_head_fwd_kernel, dim3(B), dim3(64), 0, s, cls, W, bias, labels, logits, dlogits, loss_sum, B, D, gscale);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// dcls[b][c] = sum_j dl[b][j] W[j][c];  dW[j][c] += sum_b dl[b][j] cls[b][c];  db[j] += sum_b dl[b][j]
__global__ __launch_bounds__(256) void itm_head_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ cls,
                                                           const float* __restrict__ W, float* __restrict__ dcls,
                                                           float* __restrict__ dW, float* __restrict__ db, int B, int D, float scale) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const float w0 = W[c], w1 = W[D + c];
  float g0 = 0.f, g1 = 0.f, b0 = 0.f, b1 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d0 = scale * dl[2 * b], d1 = scale * dl[2 * b + 1], x = cls[(long)b * D + c];
    dcls[(long)b * D + c] = d0 * w0 + d1 * w1;
    g0 += d0 * x; g1 += d1 * x; b0 += d0; b1 += d1;
  }
  if (dW) { dW[c] += g0; dW[D + c] += g1; }
  if (db && c == 0) { db[0] += b0; db[1] += b1; }
}
int rmcl_itm_head_bwd(const float* dl, const float* cls, const float* W, float* dcls, float* dW, float* db, int B, int D, float scale,
                      hipStream_t s) {
  RMCL_LAUNCH(itm_head_bwd_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, dl, cls, W, dcls, dW, db, B, D, scale);
  RMCL_CHECK_LAUNCH();
  return 0;
}
