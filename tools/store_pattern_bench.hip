// Microbenchmark: HBM write rate of the GEMM epilogue's store pattern (8 B per lane, 16 rows x 32-byte segments per wave
// instruction) against fully coalesced 16 B-per-lane row stores, for a [M, N] bf16 output written once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512) void k_frag(uint2* out, int M, int N) {   // tile 192x192 per block, wave 96x48, frags 16x16
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
  const int tiles_n = N / 192, tr = blockIdx.x / tiles_n, tc = blockIdx.x % tiles_n;
  for (int i = 0; i < 6; ++i) {
    const long m = (long)tr * 192 + wm * 96 + i * 16 + (lane & 15);
    if (m >= M) continue;
    for (int j = 0; j < 3; ++j) {
      const int n = tc * 192 + wn * 48 + j * 16 + 4 * (lane >> 4);
      out[(m * N + n) / 4] = make_uint2(lane, i * 3 + j);
    }
  }
}
__global__ __launch_bounds__(512) void k_rows(uint4* out, int M, int N) {   // same tile, rows written contiguously: 24 lanes x 16 B per row
  const int t = threadIdx.x;
  const int tiles_n = N / 192, tr = blockIdx.x / tiles_n, tc = blockIdx.x % tiles_n;
  for (int q = t; q < 192 * 24; q += 512) {
    const int r = q / 24, c = q % 24;
    const long m = (long)tr * 192 + r;
    if (m >= M) continue;
    out[(m * N + tc * 192) / 8 + c] = make_uint4(t, q, r, c);
  }
}
int main() {
  const int M = 11840, N = 3072;
  void* buf; hipMalloc(&buf, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = ((M + 191) / 192) * (N / 192);
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 3; ++rep) { if (v == 0) k_frag<<<blocks, 512>>>((uint2*)buf, M, N); else k_rows<<<blocks, 512>>>((uint4*)buf, M, N); }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) { if (v == 0) k_frag<<<blocks, 512>>>((uint2*)buf, M, N); else k_rows<<<blocks, 512>>>((uint4*)buf, M, N); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us per pass, %.2f TB/s\n", v == 0 ? "fragment pattern (8 B/lane, 32-B segments)" : "row pattern (16 B/lane, 384-B rows)", ms / 20 * 1e3,
           (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
  }
  return 0;
}
