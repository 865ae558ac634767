// Microbenchmark: bytes per second ONE CU can take in from its XCD's L2 in the GEMM k-loop's access pattern (48 KiB k-tiles: 384 rows x
// 128 bytes out of rows of 1536 bytes, each panel shared by 8 workgroups of one XCD), by destination:
//   dma  : global_load_lds 16 B per lane into a 3-stage LDS ring (what gemm_st / gemm_sw / gemm_dp do), two k-tiles in flight
//   reg  : global_load_dwordx4 into registers (consumed by an xor), two k-tiles in flight
//   mix  : half of every k-tile each way
// 256 workgroups of 512 threads, one per CU (144 KiB of LDS requested in every mode so that residency is the same).
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_path_bench.hip -o tools/cu_path_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define ROWB 1536            // bytes per row (K = 768 bf16)
#define PANEL_ROWS 384
#define NKT 12               // k-tiles per panel pass
#define STAGE 49152

template <int MODE>          // 0 dma, 1 reg, 2 mix
__global__ __launch_bounds__(512) void k_stream(const char* __restrict__ src, int panels_per_xcd, int passes, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const char* panel = src + ((long)xcd * panels_per_xcd + (j % panels_per_xcd)) * (long)PANEL_ROWS * ROWB;
  uint4 acc = make_uint4(0, 0, 0, 0);
  constexpr int NDMA = MODE == 0 ? 6 : MODE == 1 ? 0 : 3, NREG = 6 - NDMA;
  uint4 r[3][NREG ? NREG : 1];
  const int total = passes * NKT;
  auto issue = [&](int i, const int st) {
    const int kt = i % NKT;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
      const int inst = wave * 6 + q, row = inst * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
      __builtin_amdgcn_global_load_lds((glb_void*)(panel + (long)row * ROWB + kt * 128 + ch * 16), (lds_void*)(lds + st * STAGE + inst * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NREG; ++q) {
      const int inst = wave * 6 + NDMA + q, row = inst * 8 + (lane >> 3), ch = lane & 7;
      r[st][q] = *reinterpret_cast<const uint4*>(panel + (long)row * ROWB + kt * 128 + ch * 16);
    }
  };
  issue(0, 0);
  issue(1, 1);
  for (int i = 0; i < total; i += 3) {                          // (total is a multiple of 3: stages are compile-time indices)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (i + u + 2 < total) issue(i + u + 2, (u + 2) % 3);
      // wait for k-tile i+u: at most the two younger k-tiles' loads stay in flight
      if (i + u + 2 < total) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < NREG; ++q) { acc.x ^= r[u][q].x; acc.y ^= r[u][q].y; acc.z ^= r[u][q].z; acc.w ^= r[u][q].w; }
      __builtin_amdgcn_s_barrier();
    }
  }
  if (MODE != 1) acc.x ^= *reinterpret_cast<const uint32_t*>(lds + t * 4);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

// GEMM-like sharing: groups of 4 workgroups of one XCD stream the SAME panel in (loose) lockstep and move to a fresh panel of a
// pool far larger than the L2 every pass, so the first reader of every line misses the L2 (served by MALL / HBM), the other three hit.
// TOUCH = D > 0: k-tile i + D's 384 lines are touched into the L2 (one dword per line, LDS-DMA into a sink) while k-tile i is awaited:
// by every workgroup (WHO = 0) or by the group's first workgroup only (WHO = 1).
__device__ __forceinline__ void touch_line(const void* p, uint32_t sink_lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(sink_lds_addr));
}
template <int D, int WHO, int DEPTH = 2>
__global__ __launch_bounds__(512) void k_shared(const char* __restrict__ src, int pool, int passes, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, grp = j >> 2;
  const char* xbase = src + (long)xcd * pool * (long)PANEL_ROWS * ROWB;
  const int total = passes * NKT;
  auto addr = [&](int i, int inst) {
    const int pass = i / NKT, kt = i - pass * NKT;
    const char* panel = xbase + (long)((grp + 8 * pass) % pool) * (long)PANEL_ROWS * ROWB;
    const int row = inst * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
    return panel + (long)row * ROWB + kt * 128 + ch * 16;
  };
  auto issue = [&](int i, const int st) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
      __builtin_amdgcn_global_load_lds((glb_void*)addr(i, wave * 6 + q), (lds_void*)(lds + st * STAGE + (wave * 6 + q) * 1024), 16, 0, 0);
  };
  const uint32_t sinkaddr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds + 3 * STAGE + wave * 256));
  auto touch = [&](int i) {                                      // 384 lines of k-tile i: wave w touches rows 48 w .. 48 w + 47 (lanes 0..47)
    if (WHO == 1 && (j & 3) != 0) return;
    const int pass = i / NKT, kt = i - pass * NKT;
    const char* panel = xbase + (long)((grp + 8 * pass) % pool) * (long)PANEL_ROWS * ROWB;
    const int row = wave * 48 + (lane < 48 ? lane : 47);
    touch_line(panel + (long)row * ROWB + kt * 128, sinkaddr);
  };
  if (D > 0) for (int i = 2; i < 2 + D && i < total; ++i) touch(i);
  issue(0, 0);
  issue(1, 1);
  for (int i = 0; i < total; i += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (D > 0 && i + u + 2 + D < total) touch(i + u + 2 + D);
      if (i + u + 2 < total) issue(i + u + 2, (u + 2) % 3);
      if (DEPTH == 1) {                                          // wait for k-tile i+u+1 right behind the issue of i+u+2: every DMA has ONE period to land
        if (i + u + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (i + u + 2 < total) { if (D > 0 && (WHO == 0 || (j & 3) == 0)) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (*reinterpret_cast<const uint32_t*>(lds + t * 4) == 0x12345678u) sink[blockIdx.x] = 1;
}
template <int D, int WHO, int DEPTH = 2> static float run_shared(const char* src, int pool, int passes, uint32_t* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_shared<D, WHO, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE + 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_shared<D, WHO, DEPTH><<<256, 512, 3 * STAGE + 2048>>>(src, pool, 3, sink);
  hipEventRecord(e0);
  k_shared<D, WHO, DEPTH><<<256, 512, 3 * STAGE + 2048>>>(src, pool, passes, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

// gemm_st's k-loop skeleton on the same shared-panel stream: 8 waves, two groups one barrier apart, a k-tile = two phases of
// {optional 9 ds_read_b128, 3 LDS-DMA (A half of tile t+2 in phase 0, B half + vmcnt(6) in phase 1), lgkmcnt(0), barrier,
// optional 18 v_mfma_f32_16x16x32_bf16, barrier}.  MF / RD switch the MFMAs and the LDS reads on.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// DM (round 4): where the three LDS-DMA pieces of a phase are issued - 0: in the read phase, ahead of the barrier (gemm_st today); 1: between the
// MFMAs of the wave's own MFMA phase (one piece behind every sixth MFMA), so that the read phase - what the OTHER group's 18 MFMAs have to cover -
// carries the nine ds_read_b128 only
template <bool MF, bool RD, int PH, int DM = 0>
__global__ __launch_bounds__(512) void k_skel(const char* __restrict__ src, int pool, int passes, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 2, wn = wave & 3;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, grp = j >> 2;
  const char* xbase = src + (long)xcd * pool * (long)PANEL_ROWS * ROWB;
  const int total = passes * NKT;
  auto issue_piece = [&](int i, const int st, const int half, const int q) {
    const int pass = i / NKT, kt = i - pass * NKT;
    const char* panel = xbase + (long)((grp + 8 * pass) % pool) * (long)PANEL_ROWS * ROWB;
    const int inst = half * 24 + wave * 3 + q, row = inst * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
    __builtin_amdgcn_global_load_lds((glb_void*)(panel + (long)row * ROWB + kt * 128 + ch * 16), (lds_void*)(lds + st * STAGE + inst * 1024), 16, 0, 0);
  };
  auto issue_half = [&](int i, const int st, const int half) {      // half 0: rows 0..191 ("A"), 1: rows 192..383 ("B")
#pragma unroll
    for (int q = 0; q < 3; ++q) issue_piece(i, st, half, q);
  };
  f32x4 acc[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][6], fb[2][3];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int a = 0; a < 6; ++a) fa[s2][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 3; ++b) fb[s2][b] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  int aoff[2], boff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int sw = ((4 * s2 + (lane >> 4)) ^ (lane & 7)) * 16;
    aoff[s2] = (wm * 96 + (lane & 15)) * 128 + sw;
    boff[s2] = 24576 + (wn * 48 + (lane & 15)) * 128 + sw;
  }
  issue_half(0, 0, 0); issue_half(0, 0, 1);
  issue_half(1, 1, 0); issue_half(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int i = 0; i < total; i += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const char* cur = lds + u * STAGE;
      const bool more = i + u + 2 < total;
      if (PH == 2) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (RD) {
#pragma unroll
            for (int b = 0; b < 3; ++b) fb[0][b] = *reinterpret_cast<const bf16x8*>(cur + b * 2048 + boff[s2]);
#pragma unroll
            for (int a = 0; a < 6; ++a) fa[0][a] = *reinterpret_cast<const bf16x8*>(cur + a * 2048 + aoff[s2]);
          }
          if (DM == 0) {
            if (more) { issue_half(i + u + 2, (u + 2) % 3, s2); if (s2 == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else if (s2 == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            // the pieces of this phase go out between the MFMAs below; what must have landed before the NEXT k-tile is read was issued
            // one k-tile ago: 6 pieces of tile t+2 (this k-tile's) may stay in flight... they are issued AFTER this wait, so vmcnt(0) here
            // would wait for tile t+1 only if nothing younger is outstanding: k-step 0's three pieces are - hence vmcnt(3)
            if (s2 == 1) { if (more) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (MF) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
              for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][b], fa[0][a], acc[a][b], 0, 0, 0);
              if (DM == 1 && more && (a & 1)) {                  // one piece behind MFMAs 6, 12, 18
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(i + u + 2, (u + 2) % 3, s2, a >> 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_s_setprio(0);
          } else if (DM == 1 && more) {
            issue_half(i + u + 2, (u + 2) % 3, s2);
          }
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if (RD) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int b = 0; b < 3; ++b) fb[s2][b] = *reinterpret_cast<const bf16x8*>(cur + b * 2048 + boff[s2]);
#pragma unroll
            for (int a = 0; a < 6; ++a) fa[s2][a] = *reinterpret_cast<const bf16x8*>(cur + a * 2048 + aoff[s2]);
          }
        }
        if (more) { issue_half(i + u + 2, (u + 2) % 3, 0); issue_half(i + u + 2, (u + 2) % 3, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (MF) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s2][b], fa[s2][a], acc[a][b], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) r += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (r == 12345.f) sink[blockIdx.x] = r;
}
// Software-pipelined alternative: every wave runs MFMA(t, s) with the fragment reads of the NEXT k-step issued between its MFMAs
// (second fragment register set), ONE barrier per k-tile (after k-step 0: tile t+1 has landed for everyone, stage t % 3 has been read
// by everyone), the DMA of tile t+3 issued right behind that barrier into the stage just freed: every DMA has two periods to land.
template <int ILV>           // ILV = MFMAs per interleaved ds_read (2: one read per 2 MFMAs; 0: reads first, then MFMAs - hipcc's default order)
__global__ __launch_bounds__(512) void k_pipe(const char* __restrict__ src, int pool, int passes, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 2, wn = wave & 3;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, grp = j >> 2;
  const char* xbase = src + (long)xcd * pool * (long)PANEL_ROWS * ROWB;
  const int total = passes * NKT;
  auto issue = [&](int i, const int st) {
    const int pass = i / NKT, kt = i - pass * NKT;
    const char* panel = xbase + (long)((grp + 8 * pass) % pool) * (long)PANEL_ROWS * ROWB;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int inst = wave * 6 + q, row = inst * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
      __builtin_amdgcn_global_load_lds((glb_void*)(panel + (long)row * ROWB + kt * 128 + ch * 16), (lds_void*)(lds + st * STAGE + inst * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int aoff[2], boff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int sw = ((4 * s2 + (lane >> 4)) ^ (lane & 7)) * 16;
    aoff[s2] = (wm * 96 + (lane & 15)) * 128 + sw;
    boff[s2] = 24576 + (wn * 48 + (lane & 15)) * 128 + sw;
  }
  bf16x8 fa[2][6], fb[2][3];
  auto rd = [&](bf16x8 (&A)[6], bf16x8 (&B)[3], const char* st, const int s2) {
#pragma unroll
    for (int b = 0; b < 3; ++b) B[b] = *reinterpret_cast<const bf16x8*>(st + b * 2048 + boff[s2]);
#pragma unroll
    for (int a = 0; a < 6; ++a) A[a] = *reinterpret_cast<const bf16x8*>(st + a * 2048 + aoff[s2]);
  };
  auto mm = [&](const bf16x8 (&A)[6], const bf16x8 (&B)[3]) {
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[b], A[a], acc[a][b], 0, 0, 0);
  };
  auto ilv = [&]() {                                             // 9 reads spread over 18 MFMAs
    if (ILV > 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);       // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
      }
    }
  };
  issue(0, 0); issue(1, 1); issue(2, 2);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rd(fa[0], fb[0], lds, 0);
  for (int i = 0; i < total; i += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const char* cur = lds + u * STAGE;
      const char* nxt = lds + ((u + 1) % 3) * STAGE;
      // k-step 0 of tile i+u, reading k-step 1's fragments meanwhile
      rd(fa[1], fb[1], cur, 1);
      mm(fa[0], fb[0]);
      ilv();
      __builtin_amdgcn_sched_barrier(0);
      // tile i+u+1 landed (own part), everyone's reads of `cur` retired -> barrier -> stage `cur` is free for tile i+u+3
      if (i + u + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (i + u + 3 < total) issue(i + u + 3, u);
      rd(fa[0], fb[0], nxt, 0);
      mm(fa[1], fb[1]);
      ilv();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) r += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (r == 12345.f) sink[blockIdx.x] = r;
}
template <int ILV> static float run_pipe(const char* src, int pool, int passes, uint32_t* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe<ILV>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_pipe<ILV><<<256, 512, 3 * STAGE>>>(src, pool, 3, (float*)sink);
  hipEventRecord(e0);
  k_pipe<ILV><<<256, 512, 3 * STAGE>>>(src, pool, passes, (float*)sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <bool MF, bool RD, int PH, int DM = 0> static float run_skel(const char* src, int pool, int passes, uint32_t* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel<MF, RD, PH, DM>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_skel<MF, RD, PH, DM><<<256, 512, 3 * STAGE>>>(src, pool, 3, (float*)sink);
  hipEventRecord(e0);
  k_skel<MF, RD, PH, DM><<<256, 512, 3 * STAGE>>>(src, pool, passes, (float*)sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int MODE> static float run(const char* src, int ppx, int passes, uint32_t* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_stream<MODE><<<256, 512, 3 * STAGE>>>(src, ppx, 3, sink);    // warm the L2
  hipEventRecord(e0);
  k_stream<MODE><<<256, 512, 3 * STAGE>>>(src, ppx, passes, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  uint32_t* sink; hipMalloc(&sink, 4096);
  for (int ppx : {4, 8, 32}) {                                   // 4 panels per XCD = 2.4 MB (L2), 8 = 4.7 MB, 32 = 18.9 MB per XCD (MALL)
    const size_t bytes = (size_t)8 * ppx * PANEL_ROWS * ROWB;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    const int passes = 60;
    const double per_cu = (double)passes * NKT * STAGE;
    float a = run<0>(src, ppx, passes, sink), b = run<1>(src, ppx, passes, sink), c = run<2>(src, ppx, passes, sink);
    printf("panels per XCD %2d (%.1f MB per XCD): dma %.1f GB/s per CU   reg %.1f   mix %.1f   (chip: %.1f / %.1f / %.1f TB/s)\n", ppx, bytes / 8 / 1e6,
           per_cu / a / 1e6, per_cu / b / 1e6, per_cu / c / 1e6, per_cu * 256 / a / 1e9, per_cu * 256 / b / 1e9, per_cu * 256 / c / 1e9);
    hipFree(src);
  }
  for (int pool : {2, 32, 128}) {                                // (2 panels per XCD x 8 groups... pool 2: L2-resident)                                   // panels per XCD: 19 MB (151 MB in all: MALL) / 75 MB (604 MB: HBM)
    const size_t bytes = (size_t)8 * pool * PANEL_ROWS * ROWB;
    char* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    const int passes = 64;
    const double per_cu = (double)passes * NKT * STAGE;
    auto gb = [&](float ms) { return per_cu / ms / 1e6; };
    printf("4 workgroups per panel, pool %3d panels per XCD: no touch %.1f GB/s per CU | touch by all, 2 / 4 / 8 ahead: %.1f %.1f %.1f | by the first of 4: %.1f %.1f %.1f\n", pool,
           gb(run_shared<0, 0>(src, pool, passes, sink)), gb(run_shared<2, 0>(src, pool, passes, sink)), gb(run_shared<4, 0>(src, pool, passes, sink)),
           gb(run_shared<8, 0>(src, pool, passes, sink)), gb(run_shared<2, 1>(src, pool, passes, sink)), gb(run_shared<4, 1>(src, pool, passes, sink)),
           gb(run_shared<8, 1>(src, pool, passes, sink)));
    printf("   same, every k-tile awaited one period after its issue (gemm_st's schedule): %.1f GB/s per CU; two periods (above): %.1f\n",
           gb(run_shared<0, 0, 1>(src, pool, passes, sink)), gb(run_shared<0, 0, 2>(src, pool, passes, sink)));
    auto us = [&](float ms) { return ms * 1e3 / (passes * NKT); };
    printf("   gemm_st loop skeleton, us per k-tile (MFMA alone = 0.48): two phases per k-tile: loads only %.3f | + MFMA %.3f | + LDS reads %.3f | + both %.3f\n",
           us(run_skel<false, false, 2>(src, pool, passes, sink)), us(run_skel<true, false, 2>(src, pool, passes, sink)),
           us(run_skel<false, true, 2>(src, pool, passes, sink)), us(run_skel<true, true, 2>(src, pool, passes, sink)));
    printf("   two phases, LDS-DMA pieces issued BETWEEN the MFMAs of the wave's own MFMA phase (round 4): + MFMA %.3f | + both %.3f\n",
           us(run_skel<true, false, 2, 1>(src, pool, passes, sink)), us(run_skel<true, true, 2, 1>(src, pool, passes, sink)));
    printf("   one phase per k-tile: loads only %.3f | + MFMA %.3f | + LDS reads %.3f | + both %.3f\n",
           us(run_skel<false, false, 1>(src, pool, passes, sink)), us(run_skel<true, false, 1>(src, pool, passes, sink)),
           us(run_skel<false, true, 1>(src, pool, passes, sink)), us(run_skel<true, true, 1>(src, pool, passes, sink)));
    printf("   software-pipelined loop (one barrier per k-tile, reads between the MFMAs): %.3f us per k-tile; reads placed by hipcc: %.3f\n",
           us(run_pipe<2>(src, pool, passes, sink)), us(run_pipe<0>(src, pool, passes, sink)));
    hipFree(src);
  }
  return 0;
}
