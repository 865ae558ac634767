"""Times the fused InfoNCE (forward + dq + metrics) at the step's shape (B=64, 128-d, queue 65536), both forms (rmcl_tune_set key 5)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, stream, DEV

B, Kq = 64, 65536
g = torch.Generator().manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, 128, generator=g), dim=1).to(DEV)
k = torch.nn.functional.normalize(torch.randn(B, 128, generator=g), dim=1).to(DEV)
queue = torch.randn(128, Kq, generator=g).to(DEV)
dq, rows, loss = torch.empty(B, 128, device=DEV), torch.empty(B, 10, device=DEV), torch.zeros(1, device=DEV)
lib.rmcl_infonce_ws_bytes.restype = __import__("ctypes").c_int64
ws = torch.empty(lib.rmcl_infonce_ws_bytes(B, I64(Kq)), dtype=torch.uint8, device=DEV)
run = lambda: check(lib.rmcl_infonce_f32(P(q), P(k), P(queue), B, 128, I64(Kq), F(0.07), F(1.0 / B), P(dq), P(rows), P(loss), P(ws), stream()))


def t(n=50):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
for fold in (0, 1):
    lib.rmcl_tune_set(5, fold)
    loss.zero_()
    run()
    res[fold] = (dq.clone(), rows.clone(), float(loss))
    print(f"fold={fold}: {t():.1f} us per call (partial + combine); queue 33.5 MB -> {33.5e6 / (t() * 1e-6) / 1e12:.2f} TB/s effective")
run3 = lambda m: (lambda: check(lib.rmcl_infonce_split_bf16(P(q), P(k), P(queue), B, 128, I64(Kq), F(0.07), F(1.0 / B), P(dq), P(rows), P(loss), P(ws), m, stream())))
for m in (1, 0):
    run = run3(m)
    loss.zero_()
    run()
    l3 = float(loss)
    us = t()
    print(f"split-bf16 form, metrics={m}: {us:.1f} us per call; queue 33.5 MB -> {33.5e6 / (us * 1e-6) / 1e12:.2f} TB/s effective; "
          f"loss {l3:.6f} vs exact-f32 {res[1][2]:.6f} (drift {abs(l3 - res[1][2]):.2e}); max |dq diff| {float((res[1][0] - dq).abs().max()):.2e}")
for f in (1,):
    print(f"form {f} vs 0: max |dq diff|", float((res[0][0] - res[f][0]).abs().max()), " rows diff", float((res[0][1] - res[f][1]).abs().max()), res[0][2], res[f][2])
