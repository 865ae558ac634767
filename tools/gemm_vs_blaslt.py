"""The encoder's four linear shapes at the step's token count: this library's GEMM (bias epilogue, bf16 out) beside the
vendor library reached through torch.nn.functional.linear (hipBLASLt on this image).  A yardstick, not a product path: it
says how far the hand-written kernels are from what the vendor's tuned assembly reaches on the same shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, stream, DEV

M = 11840
g = torch.Generator().manual_seed(0)


def t(fn, n=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b = torch.zeros(N, device=DEV)
    bb = b.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    mine = lambda: check(lib.rmcl_gemm(P(A), P(W), P(out), None, P(b), None, M, N, K, I64(K), I64(K), N, 0, F(1.0), 1, 1, L.BF16, L.BF16,
                                       1, 1, 0, stream()))
    vend = lambda: torch.nn.functional.linear(A, W, bb)
    vend_nb = lambda: torch.matmul(A, W.t())
    fl = 2.0 * M * N * K
    tm, tv, tn = t(mine), t(vend), t(vend_nb)
    print(f"{name} (N={N}, K={K}): this library {tm:.1f} us = {fl / tm * 1e-6:.0f} TFLOP/s   F.linear {tv:.1f} us = {fl / tv * 1e-6:.0f} TFLOP/s"
          f"   matmul (no bias) {tn:.1f} us = {fl / tn * 1e-6:.0f} TFLOP/s", flush=True)

# what the vendor library launches for these shapes
from torch.profiler import profile, ProfilerActivity
A = torch.randn(M, 768, generator=g).to(DEV).to(torch.bfloat16)
W = (torch.randn(3072, 768, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        torch.matmul(A, W.t())
    torch.cuda.synchronize()
for e in prof.key_averages():
    print(e.key[:200], e.count, f"{e.device_time_total / max(e.count, 1):.1f} us")
