"""qkv / fc1 forward GEMM at the step's shape: plain (A = LayerNorm output, bias epilogue) vs LayerNorm-folded consumer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, stream, DEV

M, D = 11840, 768
g = torch.Generator().manual_seed(0)
xb = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
part = torch.rand(M, 16, 2, generator=g).to(DEV) + 1.0
mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)


def t(fn, n=300):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, gelu in (("qkv", 2304, 0), ("fc1", 3072, 1)):
    W = (torch.randn(N, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b, s, c = torch.zeros(N, device=DEV), torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if gelu else None
    epi = 1 | (2 | 4 if gelu else 0)
    plain = lambda: check(lib.rmcl_gemm(P(xb), P(W), P(out), P(pre), P(b), None, M, N, D, I64(D), I64(D), N, 0, F(1.0), epi, 1, L.BF16, L.BF16,
                                        1, 1, 0, stream()))
    fold = lambda: check(lib.rmcl_linear_lnfold(P(xb), P(W), P(s), P(c), P(part), 16, P(out), P(pre), M, N, D, gelu, F(1e-6), P(mean), P(rstd),
                                                stream()))
    line = f"{name}: plain {t(plain):.1f} us   LayerNorm-folded {t(fold):.1f} us"
    print(line)

# producer side: fp32 output + residual, plain vs with the bf16 copy + per-row partial sums
for name, K in (("proj", 768), ("fc2", 3072)):
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(D, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b, res = torch.zeros(D, device=DEV), torch.randn(M, D, generator=g).to(DEV)
    out, outb = torch.empty(M, D, device=DEV), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    prt = torch.empty(M, 16, 2, device=DEV)
    plain = lambda: check(lib.rmcl_gemm(P(A), P(W), P(out), None, P(b), P(res), M, D, K, I64(K), I64(K), D, D, F(1.0), 1 | 8, 1, L.BF16, L.F32,
                                        1, 1, 0, stream()))
    prod = lambda: check(lib.rmcl_linear_rowstat(P(A), P(W), P(b), P(res), P(out), P(outb), P(prt), M, D, K, stream()))
    print(f"{name}: plain {t(plain):.1f} us   with bf16 copy + row partials {t(prod):.1f} us")
