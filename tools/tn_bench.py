"""Short-K fp32 TN GEMM (C += A^T B, K = batch rows): first form (LDS-staged 128x128x16 kernel) vs gemm_tn_shortk_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, stream, DEV


def t(fn, n=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, M, N in ((64, 768, 3072), (64, 3072, 768), (64, 768, 768), (64, 8192, 8192)):
    A = torch.randn(K, M, device=DEV)
    B = torch.randn(K, N, device=DEV)
    C_ = torch.zeros(M, N, device=DEV)
    fn = lambda: check(lib.rmcl_gemm(P(A), P(B), P(C_), None, None, None, M, N, K, I64(M), I64(N), N, 0, F(1.0), 64, 1, L.F32, L.F32, 0, 0, 1, stream()))
    out = []
    for form in (0, 1):
        check(lib.rmcl_tune_set(6, form))
        out.append(t(fn))
    check(lib.rmcl_tune_set(6, -1))
    print(f"K={K} M={M} N={N}: LDS-staged {out[0]:.1f} us   short-K {out[1]:.1f} us", flush=True)
