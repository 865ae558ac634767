"""Experiment: do the synchronized epilogue store bursts of the persistent GEMM cost time?  Two independent qkv-forward GEMMs
(M=11840, N=2304, K=768, bias, bf16 out) back to back on one stream with 248 workgroups each, vs concurrently on two streams
with 124 workgroups each (disjoint CUs, free to drift out of phase)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, DEV
import ctypes as C

M, N, K = 11840, 2304, 768
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
A = [mk(M, K), mk(M, K)]
W = [mk(N, K), mk(N, K)]
Cc = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
bias = torch.zeros(N, device=DEV)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def gemm(i, st):
    check(lib.rmcl_gemm(P(A[i]), P(W[i]), P(Cc[i]), None, P(bias), None, M, N, K, I64(K), I64(K), N, 0, F(1.0), 1, 1, L.BF16, L.BF16, 1, 1, 0,
                        C.c_void_p(st.cuda_stream)))


def run(concurrent, reps=20):
    main = torch.cuda.current_stream()
    lib.rmcl_tune_set(1, 132 if concurrent else 8)
    for it in range(reps + 3):
        if it == 3:
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if concurrent:
            for s in streams:
                s.wait_stream(main)
            gemm(0, streams[0]); gemm(1, streams[1])
            for s in streams:
                main.wait_stream(s)
        else:
            gemm(0, main); gemm(1, main)
    e1.record(); torch.cuda.synchronize()
    lib.rmcl_tune_set(1, 8)
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(2):
    print(f"two GEMMs sequential (248 WGs each): {run(False):.1f} us   concurrent on two streams (124 WGs each): {run(True):.1f} us")
