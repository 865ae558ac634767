"""Round-4 experiment: does an L2 prefetch agent beside the GEMM (csrc/embed_misc.hip l2_prefetch_kernel, C ABI rmcl_l2_prefetch_experiment)
bring the k-loop from its cold rate (~1.0 us per k-tile: first reader of every activation line misses to the Infinity Cache / HBM) towards
the hot one (0.71 in the skeleton, tools/cu_path_bench.hip)?

One plain bf16 GEMM of the step's data-gradient class, out[M, 768] = A[M, K] W[768, K]^T at M = 11 840 (62 row tiles x 4 column tiles = 248
workgroups, XCD x works on tile ids 31 x .. 31 x + 30), K = 768 / 2304 / 3072, operands evicted between launches (512 MB fill).  The agent is
launched on a second stream just ahead of the GEMM: `wgs` single-wave workgroups, the first `per_xcd` per XCD work; `tick` = pacing per k-tile,
`lead` = k-tiles ahead of the clock.  Prints GEMM time alone (cold / hot) and with the agent for a sweep of (tick, lead, per_xcd)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, DEV

M, N = 11840, 768
g = torch.Generator().manual_seed(0)
big = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def sp(st):
    return ctypes.c_void_p(st.cuda_stream)


def run_case(K):
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    counter = torch.zeros(8, dtype=torch.int32, device=DEV)
    stamps = torch.zeros(16, dtype=torch.int64, device=DEV)
    nk, rpt = K // 64, -(-M // 62)

    def gemm():
        check(lib.rmcl_gemm(P(A), P(W), P(out), None, None, None, M, N, K, I64(K), I64(K), N, 0, F(1.0), 0, 1, L.BF16, L.BF16, 1, 1, 0, sp(main)), "gemm")

    def timed(agent, cold=True, reps=6):
        ts = []
        for _ in range(reps):
            if cold:
                big.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if agent is not None:
                tick, lead, per, wgs = agent
                counter.zero_()
                torch.cuda.synchronize()
                check(lib.rmcl_l2_prefetch_experiment(P(A), I64(K * 2), M, rpt, 31, 4, P(W), I64(K * 2), N, nk, tick, lead, per, wgs, P(counter), P(stamps),
                                                      sp(side)), "agent")
            e0.record(main)
            gemm()
            e1.record(main)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    gemm()
    torch.cuda.synchronize()
    ref = out.clone()
    base_c, base_h = timed(None, True), timed(None, False)
    print(f"K = {K}: GEMM alone cold {base_c:.1f} us, hot {base_h:.1f} us  ({nk} k-tiles)")
    for per, wgs in ((1, 64), (2, 128), (4, 256)):
        for tick in (60, 75, 90):
            for lead in (3, 6, 12):
                t = timed((tick, lead, per, wgs))
                st = stamps.cpu().view(8, 2)
                dur = (st[:, 1] - st[:, 0]).float().mean().item() / 100
                print(f"   agent per_xcd {per} ({wgs} wgs), tick {tick / 100:.2f} us, lead {lead:2d}: GEMM {t:.1f} us   (agent ran {dur:.1f} us; counters {counter.cpu().tolist()})")
    assert torch.equal(out, ref)


for K in (3072, 768, 2304):
    run_case(K)
