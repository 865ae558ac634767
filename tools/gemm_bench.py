"""Micro-benchmark of the GEMM shapes of one RMCL step (B=64): TFLOP/s per shape and layout,
random bf16 data (guide rule 25), median of interleaved rounds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import DEV, L, lib, check, P, I64, F, stream

def bench(M, N, K, a_kc, b_kc, dto=L.BF16, epi=0, iters=20, exact=0):
    g = torch.Generator(device="cpu").manual_seed(0)
    A = (torch.randn(M, K, generator=g) if a_kc else torch.randn(K, M, generator=g)).to(DEV).to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g) * 0.05 if b_kc else torch.randn(K, N, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    Cm = torch.zeros(M, N, dtype=torch.float32 if dto == L.F32 else torch.bfloat16, device=DEV)
    bias = torch.zeros(N, device=DEV)
    lda, ldb = A.shape[1], B.shape[1]
    def run():
        check(lib.rmcl_gemm(P(A), P(B), P(Cm), None, P(bias), None, M, N, K, I64(lda), I64(ldb), N, 0, F(1.0), epi, 1,
                            L.BF16, dto, a_kc, b_kc, exact, stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    M = 11840
    shapes = [("qkv fwd NT", M, 2304, 768, 1, 1, L.BF16, 1), ("proj fwd NT", M, 768, 768, 1, 1, L.F32, 1),
              ("fc1 fwd NT", M, 3072, 768, 1, 1, L.BF16, 1 | 2), ("fc2 fwd NT", M, 768, 3072, 1, 1, L.F32, 1),
              ("patch fwd NT", 9216, 768, 3072, 1, 1, L.F32, 1),
              ("fc2 dX NN", M, 3072, 768, 1, 0, L.BF16, 0), ("fc1 dX NN", M, 768, 3072, 1, 0, L.F32, 0),
              ("qkv dX NN", M, 768, 2304, 1, 0, L.F32, 0), ("proj dX NN", M, 768, 768, 1, 0, L.BF16, 0),
              ("fc1 dW TN", 3072, 768, M, 0, 0, L.F32, 64), ("fc2 dW TN", 768, 3072, M, 0, 0, L.F32, 64),
              ("qkv dW TN", 2304, 768, M, 0, 0, L.F32, 64), ("proj dW TN", 768, 768, M, 0, 0, L.F32, 64)]
    cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
    for name, m, n, k, akc, bkc, dto, epi in shapes:
        line = f"{name:14s} M={m:6d} N={n:5d} K={k:6d} "
        for c in cfgs:
            lib.rmcl_tune_set(0, c)
            ms, tf = bench(m, n, k, akc, bkc, dto, epi)
            line += f" | cfg{c}: {ms*1e3:7.1f} us {tf:7.1f} TF"
        print(line, flush=True)
