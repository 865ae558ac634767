"""Phase timeline of ONE workgroup of gemm_st_kernel (developer build: csrc/build.sh with RMCL_EXTRA_FLAGS=-DST_TRACE).
Usage: python tools/st_trace.py {proj|fc2|qkv|projdx|fc1|fc2dx|attnbwd|attnfwd|dw}   - launches the step's form of that GEMM 5x and prints the last stamps."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, stream, DEV

which = sys.argv[1] if len(sys.argv) > 1 else "proj"
if which == "dw":                          # the per-layer weight-gradient launch INSIDE a training step (beside the data-gradient chain)
    import rmcl_pkg  # noqa: F401
    from rmcl_amd.vilt.config import task_moco
    from rmcl_amd.vilt.modules import ViLTransformerSS
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = task_moco(per_gpu_batchsize=64, num_gpus=1, num_nodes=1, adv_steps_img=3, drop_rate=0.0, image_view=True, max_steps=1000, dense_images=True)
    model = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
    model.train()
    (opt,), _ = model.configure_optimizers()
    batch = bench.synthetic_batch(cfg, 64, 1, DEV)
    for i in range(3):
        loss = model.training_step(batch, i); loss.backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert lib.rmcl_debug_dw_trace(buf) == 0
    names = {0: "start", 1: "first two k-tiles landed", 2: "k-loop done (185 k-tiles)", 3: "epilogue (+= into the gradient arena)", 4: "stores drained"}
    for w in range(2):
        t = [buf[w * 32 + i] for i in range(32)]
        print(f"--- wave {4 * w} of workgroup 100 of the LAST weight-gradient launch (layer 0) of a step, us")
        prev = t[0]
        for i in sorted(names):
            print(f"  {names[i]:38s} {(t[i] - t[0]) / 100:7.2f}  (+{(t[i] - prev) / 100:.2f})"); prev = t[i]
    sys.exit(0)
M, D = 11840, 768
g = torch.Generator().manual_seed(0)
if which in ("proj", "fc2"):
    K = 768 if which == "proj" else 3072
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(D, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b, res = torch.zeros(D, device=DEV), torch.randn(M, D, generator=g).to(DEV)
    out, outb = torch.empty(M, D, device=DEV), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    prt = torch.empty(M, 16, 2, device=DEV)
    run = lambda: check(lib.rmcl_linear_rowstat(P(A), P(W), P(b), P(res), P(out), P(outb), P(prt), M, D, K, stream()))
elif which == "qkv":
    N = 2304
    xb = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
    part = torch.rand(M, 16, 2, generator=g).to(DEV) + 1.0
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    W = (torch.randn(N, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    s_, c_ = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    run = lambda: check(lib.rmcl_linear_lnfold(P(xb), P(W), P(s_), P(c_), P(part), 16, P(out), None, M, N, D, 0, F(1e-6), P(mean), P(rstd), stream()))
elif which in ("fc1", "fc2dx"):             # the 192x384-tile kernel (gemm_sw.hip): fc1 forward with stash, fc2-dX with GELU'
    H = 3072
    if which == "fc1":
        xb = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
        part = torch.rand(M, 16, 2, generator=g).to(DEV) + 1.0
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        W = (torch.randn(H, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
        s_, c_ = torch.randn(H, generator=g).to(DEV), torch.randn(H, generator=g).to(DEV)
        out, pre = torch.empty(M, H, dtype=torch.bfloat16, device=DEV), torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
        run = lambda: check(lib.rmcl_linear_lnfold(P(xb), P(W), P(s_), P(c_), P(part), 16, P(out), P(pre), M, H, D, 1, F(1e-6), P(mean), P(rstd), stream()))
    else:
        dy = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
        W = (torch.randn(H, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)          # transposed shadow: [N = 3072][K = 768]
        u = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
        out = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
        run = lambda: check(lib.rmcl_gemm(P(dy), P(W), P(out), None, None, P(u), M, H, D, I64(D), I64(D), H, H, F(1.0), 16, 1, L.BF16, L.BF16, 1, 1, 0, stream()))
elif which in ("attnbwd", "attnfwd"):         # fused attention, B = 64, N = 185, 12 heads
    B, N, Hh = 64, 185, 12
    qkv = torch.randn(B * N, 3 * D, generator=g).to(DEV).to(torch.bfloat16)
    mask = torch.ones(B, N, dtype=torch.int32, device=DEV)
    o = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    lib.rmcl_attention_scratch_elems.restype = ctypes.c_int64
    ne = lib.rmcl_attention_scratch_elems(B, Hh, N)
    probs = torch.empty(ne, dtype=torch.bfloat16, device=DEV)   # (the fused path keeps the per-row log-sum-exp here)
    scores, dS = torch.empty(ne, device=DEV), torch.empty(ne, dtype=torch.bfloat16, device=DEV)
    check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(o), P(probs), P(scores), B, N, Hh, L.BF16, 0, stream()))
    do = torch.randn(B * N, D, generator=g).to(DEV).to(torch.bfloat16)
    dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device=DEV)
    run = lambda: check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(do), P(o), P(dqkv), P(scores), P(dS), B, N, Hh, L.BF16, 0, stream()))
    if which == "attnfwd":
        run = lambda: check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(o), P(probs), P(scores), B, N, Hh, L.BF16, 0, stream()))
else:                                     # plain bf16-out NT GEMM, K = 768 (proj-dX)
    A = torch.randn(M, D, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(D, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    out = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    run = lambda: check(lib.rmcl_gemm(P(A), P(W), P(out), None, None, None, M, D, D, I64(D), I64(D), D, 0, F(1.0), 0, 1, L.BF16, L.BF16, 1, 1, 0, stream()))
big = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
for _ in range(5):
    if not os.environ.get("ST_TRACE_HOT"):
        big.zero_()                        # evict the operands from L2 / MALL (ST_TRACE_HOT=1: leave them where the previous launch left them)
    run()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
if which == "attnbwd":
    assert lib.rmcl_debug_at_trace(buf) == 0
    names = {0: "start", 1: "Q / dO / K staging issued", 2: "delta done (dO, O rows from global)", 3: "K / V fragments in registers", 4: "barrier (images landed)",
             5: "phase 1 done (12 query tiles)", 6: "dK / dV stores issued", 7: "barrier", 8: "phase 2 done (dQ)", 9: "dQ stores issued", 10: "stores drained"}
elif which == "attnfwd":
    assert lib.rmcl_debug_at_trace(buf) == 0
    for w in range(2):                                           # (the forward's stamps sit at 16.. of the same array)
        for i_ in range(16):
            buf[w * 32 + i_] = buf[w * 32 + 16 + i_] if i_ < 10 else 0
    names = {0: "start", 1: "K / V staging issued, mask row filled", 2: "Q fragments issued", 3: "barrier (images landed)", 4: "first query tile done",
             6: "second query tile done", 8: "output stores issued", 9: "stores drained"}
elif which in ("fc1", "fc2dx"):
    assert lib.rmcl_debug_sw_trace(buf) == 0
    names = {0: "start", 1: "prologue landed", 2: "k-loop done", 10: "epilogue done", 11: "stores drained"}
    for ch in range(2):
        names.update({4 + 3 * ch: f"chunk{ch}: math + LDS images", 5 + 3 * ch: f"chunk{ch}: barrier + read-back + stores issued", 6 + 3 * ch: f"chunk{ch}: barrier"})
else:
    assert lib.rmcl_debug_st_trace(buf) == 0
    names = {0: "start", 1: "prologue landed", 2: "k-loop done", 3: "epi: vectors/rowstat", 4: "epi: barrier", 20: "epilogue done", 21: "stores drained"}
for ch in range(3 if which not in ("fc1", "fc2dx", "attnbwd", "attnfwd") else 0):
    names.update({5 + 4 * ch: f"chunk{ch}: math+LDS write", 6 + 4 * ch: f"chunk{ch}: barrier", 7 + 4 * ch: f"chunk{ch}: read-back+stores issued",
                  8 + 4 * ch: f"chunk{ch}: barrier"})
for w in range(2):
    t = [buf[w * 32 + i] for i in range(32)]
    print(f"--- traced wave {w} (GEMMs: wave {4 * w} of workgroup 100; attention: wave {5 * w} of workgroup 300), us from that wave's start")
    prev = t[0]
    for i in sorted(names):
        if t[i] >= t[0] and t[i] > 0:
            print(f"  {names[i]:38s} {(t[i] - t[0]) / 100:7.2f}  (+{(t[i] - prev) / 100:.2f})")
            prev = t[i]
