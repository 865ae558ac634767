"""Correctness + speed of one tune cfg of the bf16 GEMM against torch (fp32 reference of the same bf16 operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import DEV, L, lib, check, P, I64, F, stream
from tools.gemm_bench import bench

def run_check(cfg, M, N, K, dto):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    Cm = torch.full((M, N), 7.0, dtype=torch.float32 if dto == L.F32 else torch.bfloat16, device=DEV)
    lib.rmcl_tune_set(0, cfg)
    check(lib.rmcl_gemm(P(A), P(B), P(Cm), None, P(bias), None, M, N, K, I64(K), I64(K), N, 0, F(1.0), 1, 1, L.BF16, dto, 1, 1, 0, stream()))
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t() + bias
    err = (Cm.float() - ref).abs().max().item()
    return err, ref.abs().max().item()

if __name__ == "__main__":
    cfgs = [int(c) for c in sys.argv[1].split(",")]
    for (M, N, K, dto) in [(192, 768, 128, L.F32), (300, 768, 192, L.F32), (11840, 768, 768, L.F32), (11840, 2304, 768, L.BF16), (1000, 768, 3072, L.BF16)]:
        for c in cfgs:
            for rep in range(3):
                err, mx = run_check(c, M, N, K, dto)
            print(f"check cfg{c} M={M} N={N} K={K} dto={dto}: max err {err:.4g} (ref max {mx:.3g})", flush=True)
    Mb = 11840
    for name, m, n, k, dto in [("3840^3", 3840, 3840, 3840, L.BF16), ("7680^3", 7680, 7680, 7680, L.BF16), ("qkv", Mb, 2304, 768, L.BF16), ("proj", Mb, 768, 768, L.F32),
                               ("fc1", Mb, 3072, 768, L.BF16), ("fc2", Mb, 768, 3072, L.F32), ("M12288 fc2", 12288, 768, 3072, L.F32), ("M12288 fc1", 12288, 3072, 768, L.BF16)]:
        line = f"{name:12s}"
        for c in cfgs:
            lib.rmcl_tune_set(0, c)
            ms, tf = bench(m, n, k, 1, 1, dto, 1)
            line += f" | cfg{c}: {ms*1e3:8.1f}us {tf:7.1f}TF"
        print(line, flush=True)
