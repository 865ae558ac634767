"""Times the fused attention kernels at the step's shape (B=64, N=185, H=12): forward, two-kernel backward, one-kernel backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, stream, DEV

B, N, H, D = 64, 185, 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * N, 3 * D, generator=g).to(DEV).to(torch.bfloat16)
dout = torch.randn(B * N, D, generator=g).to(DEV).to(torch.bfloat16)
mask = torch.ones(B, N, dtype=torch.int32, device=DEV)
ne = lib.rmcl_attention_scratch_elems(B, H, N)
out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
probs = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
scores = torch.empty(ne, dtype=torch.float32, device=DEV)
dS = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device=DEV)


def t(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(out), P(probs), P(scores), B, N, H, L.BF16, 0, stream()))
b2 = lambda: check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(dout), None, P(dqkv), P(scores), P(dS), B, N, H, L.BF16, 0, stream()))
b1 = lambda: check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(dout), P(out), P(dqkv), P(scores), P(dS), B, N, H, L.BF16, 0, stream()))
print(f"fwd {t(fwd):.1f} us   bwd two kernels {t(b2):.1f} us   bwd one kernel {t(b1):.1f} us")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def cold(fn, n=7):
    ts = []
    for _ in range(n):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


for wv in (4, 6, 8, 12):
    lib.rmcl_tune_set(4, wv)
    print(f"fwd with {wv} waves per workgroup: {t(fwd):.1f} us")
lib.rmcl_tune_set(4, 8)
for wg in (0, 128, 192, 248, 256, 384):
    lib.rmcl_tune_set(8, wg)
    print(f"bwd one kernel, {wg or B * H} workgroups: hot {t(b1):.1f} us   cold {cold(b1):.1f} us   (fwd cold {cold(fwd):.1f} us)")
lib.rmcl_tune_set(8, 256)
for tpw in (1, 2, 3):
    lib.rmcl_tune_set(9, tpw)
    for wg in (0, 256):
        lib.rmcl_tune_set(8, wg)
        print(f"bwd one kernel, {tpw} key tiles per wave, {wg or B * H} workgroups: hot {t(b1):.1f} us   cold {cold(b1):.1f} us")
lib.rmcl_tune_set(9, 1)
lib.rmcl_tune_set(8, 256)
