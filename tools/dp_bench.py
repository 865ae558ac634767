"""A/B of the GEMM kernel families on the step's activation GEMMs WITH their real epilogues (B = 64: M = 11840):
tune cfg -1 (gemm_st / gemm_sw: one 8-wave workgroup per CU) against cfg 80 (gemm_dp: two 4-wave workgroups per CU).
Interleaved rounds in one process (guide rule 24), random operands (rule 25); "hot" = launches back to back, "cold" = a
512 MiB fill between launches (the state the operands are in inside a step).  usage: python tools/dp_bench.py [cfgs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import DEV, L, lib, check, P, I64, F, stream

M, D, MLP = 11840, 768, 3072


def rnd(*shape, seed=0, scale=1.0, dt=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


def make_cases():
    bf = torch.bfloat16
    cases = {}
    xb = rnd(M, D, seed=1, dt=bf)
    part = torch.rand(M, 16, 2, device=DEV) + 1.0
    part[:, :, 1] += 60.0
    for name, N in (("qkv lnfold", 3 * D), ("fc1 lnfold+gelu+stash", MLP)):
        wf = rnd(N, D, seed=2, scale=0.05, dt=bf)
        s, c = rnd(N, seed=3), rnd(N, seed=4)
        out = torch.empty(M, N, dtype=bf, device=DEV)
        pre = torch.empty(M, N, dtype=bf, device=DEV) if N == MLP else None
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        gelu = 1 if N == MLP else 0
        cases[name] = (2.0 * M * N * D, lambda wf=wf, s=s, c=c, out=out, pre=pre, mean=mean, rstd=rstd, N=N, gelu=gelu: check(
            lib.rmcl_linear_lnfold(P(xb), P(wf), P(s), P(c), P(part), 16, P(out), P(pre), M, N, D, gelu, F(1e-6), P(mean), P(rstd), stream())))
    for name, K in (("proj rowstat", D), ("fc2 rowstat", MLP)):
        A = rnd(M, K, seed=5, dt=bf)
        W = rnd(D, K, seed=6, scale=0.05, dt=bf)
        b, res = rnd(D, seed=7), rnd(M, D, seed=8)
        out = torch.empty(M, D, device=DEV)
        outb = torch.empty(M, D, dtype=bf, device=DEV)
        pt = torch.empty(M, 16, 2, device=DEV)
        cases[name] = (2.0 * M * D * K, lambda A=A, W=W, b=b, res=res, out=out, outb=outb, pt=pt, K=K: check(
            lib.rmcl_linear_rowstat(P(A), P(W), P(b), P(res), P(out), P(outb), P(pt), M, D, K, stream())))

    def plain(name, N, K, epi, dto, aux=None, bias=True):
        A = rnd(M, K, seed=9, dt=bf)
        W = rnd(N, K, seed=10, scale=0.05, dt=bf)
        b = rnd(N, seed=11) if bias else None
        Cm = torch.empty(M, N, dtype=torch.float32 if dto == L.F32 else bf, device=DEV)
        U = rnd(M, N, seed=12, dt=bf) if aux else None
        cases[name] = (2.0 * M * N * K, lambda: check(lib.rmcl_gemm(P(A), P(W), P(Cm), None, P(b), P(U), M, N, K, I64(K), I64(K), N, N if aux else 0,
                                                                     F(1.0), epi, 1, L.BF16, dto, 1, 1, 0, stream())))
    def blocked(name, N, K, epi, dto, aux=None):
        """the same GEMM on gemm_dp with BOTH operands k-blocked [K/32][rows][32] (only cfg 80 differs from the row-major case)"""
        A = rnd(M, K, seed=9, dt=bf).view(M, K // 32, 32).permute(1, 0, 2).contiguous()
        W = rnd(N, K, seed=10, scale=0.05, dt=bf).view(N, K // 32, 32).permute(1, 0, 2).contiguous()
        b = rnd(N, seed=11)
        Cm = torch.empty(M, N, dtype=torch.float32 if dto == L.F32 else bf, device=DEV)
        C2 = torch.empty(M, N, dtype=bf, device=DEV) if epi & 4 else None
        U = rnd(M, N, seed=12, dt=bf) if aux else None
        cases[name] = (2.0 * M * N * K, lambda: check(lib.rmcl_gemm_kblk(P(A), P(W), P(Cm), P(C2), P(b), P(U), M, N, K, N, N if aux else 0, epi, dto, 3,
                                                                          stream())))
    blocked("KBLK qkv bias", 3 * D, D, 1, L.BF16)
    blocked("KBLK fc1 bias+gelu+stash", MLP, D, 1 | 2 | 4, L.BF16)
    blocked("KBLK fc2-dX gelu'", MLP, D, 16, L.BF16, aux=True)
    blocked("KBLK fc2 bias f32", D, MLP, 1, L.F32)
    blocked("KBLK proj bias f32", D, D, 1, L.F32)
    plain("qkv bias (FULL fwd)", 3 * D, D, 1, L.BF16)
    plain("fc2-dX gelu'", MLP, D, 16, L.BF16, aux=True, bias=False)
    plain("fc1-dX", D, MLP, 0, L.BF16, bias=False)
    plain("qkv-dX", D, 3 * D, 0, L.BF16, bias=False)
    plain("proj-dX", D, D, 0, L.BF16, bias=False)
    return cases


def main():
    # variants: "cfg" or "cfg:stagger" (stagger = tune key 7: start delay of every CU's second gemm_dp workgroup, 10 ns ticks)
    cfgs = [c for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else ["-1", "80"]

    def select(c):
        cfg, _, stg = c.partition(":")
        lib.rmcl_tune_set(0, int(cfg))
        lib.rmcl_tune_set(7, int(stg or 0))
    cases = make_cases()
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    rounds, hot_n = 7, 10
    for name, (flops, run) in cases.items():
        res = {c: {"hot": [], "cold": []} for c in cfgs}
        for c in cfgs:
            select(c)
            for _ in range(2):
                run()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for c in cfgs:
                select(c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(hot_n):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res[c]["hot"].append(e0.elapsed_time(e1) / hot_n * 1e3)
                flush.fill_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                res[c]["cold"].append(e0.elapsed_time(e1) * 1e3)
        line = f"{name:24s}"
        for c in cfgs:
            h, k = sorted(res[c]["hot"]), sorted(res[c]["cold"])
            hm, km = h[len(h) // 2], k[len(k) // 2]
            line += f" | {c:>7s}: hot {hm:6.1f} us ({flops / hm / 1e6:6.0f} TF) cold {km:6.1f} us ({flops / km / 1e6:6.0f} TF)"
        print(line, flush=True)
    lib.rmcl_tune_set(0, -1)
    lib.rmcl_tune_set(7, 0)


if __name__ == "__main__":
    main()
