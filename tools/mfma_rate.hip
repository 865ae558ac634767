// Issue rate of v_mfma_f32_32x32x2_f32 (the exact-fp32 MFMA of infonce.hip / gemm_exact.hip) on one MI355X: one wave per SIMD,
// (a) ONE dependent accumulator chain, (b) two, (c) four independent chains.  Prints cycles per MFMA per SIMD and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH>
void run(float* out, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd;
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * 8 * CH;
  const double flops = mfma_per_wave * 4096.0 * 4 * grid;
  printf("chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", CH, waves_per_simd, ms,
         flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  run<1>(out, 1); run<2>(out, 1); run<4>(out, 1); run<1>(out, 2); run<2>(out, 2);
  return 0;
}
