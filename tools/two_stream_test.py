"""Does the encoder forward run faster as TWO half-batch chains on two HIP streams than as one full-batch chain?
One chain = 12 x [qkv GEMM, fused attention forward, proj producer (+residual, row statistics), fc1 GEMM (+GELU), fc2 producer] through the C ABI at
B = 64 (M = 11 840: every launch fills the chip, all CUs reach their epilogues together) against 2 x B = 32 (M = 5 920: 124-128 workgroups per launch,
the two chains side by side on the CUs; `lag` microseconds of start offset so that one chain's epilogues fall into the other's loops)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, DEV

D, H, N, LAYERS = 768, 12, 185, 12
g = torch.Generator().manual_seed(0)
Wqkv = (torch.randn(3 * D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
Wo = (torch.randn(D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
W1 = (torch.randn(4 * D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
W2 = (torch.randn(D, 4 * D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
bq, bo, b1, b2 = (torch.zeros(n, device=DEV) for n in (3 * D, D, 4 * D, D))
lib.rmcl_attention_scratch_elems.restype = ctypes.c_int64


class Chain:
    def __init__(self, B):
        self.B, self.M = B, B * N
        M = self.M
        self.xb = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
        self.x32 = self.xb.float()
        self.qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
        self.att = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        self.h = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=DEV)
        self.y32 = torch.empty(M, D, device=DEV)
        self.yb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        self.prt = torch.empty(M, 16, 2, device=DEV)
        self.mask = torch.ones(B, N, dtype=torch.int32, device=DEV)
        ne = lib.rmcl_attention_scratch_elems(B, H, N)
        self.probs = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
        self.scores = torch.empty(max(ne, 1), dtype=torch.float32, device=DEV)

    def layer(self, s):
        M, st = self.M, ctypes.c_void_p(s.cuda_stream)
        check(lib.rmcl_gemm(P(self.xb), P(Wqkv), P(self.qkv), None, P(bq), None, M, 3 * D, D, I64(D), I64(D), 3 * D, 0, F(1.0), L.EPI_BIAS, 1, L.BF16, L.BF16, 1, 1, 0, st))
        check(lib.rmcl_attention_fwd(P(self.qkv), P(self.mask), P(self.att), P(self.probs), P(self.scores), self.B, N, H, L.BF16, 0, st))
        check(lib.rmcl_linear_rowstat(P(self.att), P(Wo), P(bo), P(self.x32), P(self.y32), P(self.yb), P(self.prt), M, D, D, st))
        check(lib.rmcl_gemm(P(self.yb), P(W1), P(self.h), None, P(b1), None, M, 4 * D, D, I64(D), I64(D), 4 * D, 0, F(1.0), L.EPI_BIAS | L.EPI_GELU, 1, L.BF16, L.BF16, 1, 1, 0, st))
        check(lib.rmcl_linear_rowstat(P(self.h), P(W2), P(b2), P(self.y32), P(self.x32), P(self.xb), P(self.prt), M, D, 4 * D, st))


check(lib.rmcl_tune_set(0, 60))               # 192x192 tiles for every GEMM here (the default routing wants >= 0.7 of the CU rounds filled by ONE launch)
full, h0, h1 = Chain(64), Chain(32), Chain(32)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
spin_per_us = None


def timed(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[len(ts[1:]) // 2]


def one_chain():
    cur = torch.cuda.current_stream()
    for _ in range(LAYERS):
        full.layer(cur)


def halves_sequential():
    cur = torch.cuda.current_stream()
    for _ in range(LAYERS):
        h0.layer(cur)
        h1.layer(cur)


def two_chains(lag_us):
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur)
    s1.wait_stream(cur)
    if lag_us:
        with torch.cuda.stream(s1):
            torch.cuda._sleep(int(lag_us * spin_per_us))
    for _ in range(LAYERS):                        # enqueue alternately so that neither queue runs dry
        h0.layer(s0)
        h1.layer(s1)
    cur.wait_stream(s0)
    cur.wait_stream(s1)


# calibrate torch.cuda._sleep (cycles of the device's spin clock per microsecond)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
spin_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
print(f"_sleep calibration: {spin_per_us:.1f} cycles per us")
print(f"one chain, B = 64:                      {timed(one_chain):.3f} ms per 12-layer forward")
print(f"two B = 32 chains, one stream (serial): {timed(halves_sequential):.3f} ms")
for lag in (0, 10, 25, 50, 100, 150):
    print(f"two B = 32 chains, two streams, second chain {lag:3d} us late: {timed(lambda: two_chains(lag)):.3f} ms")
