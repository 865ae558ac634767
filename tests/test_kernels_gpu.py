"""-m gpu: kernel-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp32/fp64
references of the same op.  Tolerances: exact-f32 kernels 1e-5 relative (fp32 summation order only);
bf16-storage variants 2^-8 relative to the row scale."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import DEV, L, lib, check, P, I64, F, stream, gemm, tdt  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("shape", [(370, 768, 768), (128, 128, 16), (200, 2304, 768), (37, 40, 24), (1, 8, 8), (513, 136, 264)])
def test_gemm_nt_nn_tn(dt, shape):
    M, N, K = shape
    X = rnd(M, K, seed=1).to(tdt(dt))
    W = rnd(N, K, seed=2, scale=0.05).to(tdt(dt))
    ref = X.double() @ W.double().t()
    tol = 2e-5 if dt == L.F32 else 1e-2
    out = gemm(X, W, M, N, K, 1, 1, dt, L.F32)                                   # NT: Y = X W^T
    assert rel_err(out, ref) < tol
    Wn = W.t().contiguous()                                                      # [K, N]
    out = gemm(X, Wn, M, N, K, 1, 0, dt, L.F32)                                  # NN: Y = X Wn
    assert rel_err(out, ref) < tol
    Xt = X.t().contiguous()                                                      # [K, M]
    if M % 8 == 0 or dt == L.F32 and M % 4 == 0:
        out = gemm(Xt, Wn, M, N, K, 0, 0, dt, L.F32)                             # TN: Y = Xt^T Wn
        assert rel_err(out, ref) < tol


def test_gemm_tn_splitk_atomic_accumulates():
    T, N, K = 1110, 768, 256                                                     # dW[N,K] += dY[T,N]^T X[T,K]
    dY, X = rnd(T, N, seed=3), rnd(T, K, seed=4)
    base = rnd(N, K, seed=5)
    ref = base.double() + dY.double().t() @ X.double()
    out = gemm(dY, X, N, K, T, 0, 0, L.F32, L.F32, epi=32, splitk=5, C_init=base)
    assert rel_err(out, ref) < 2e-5


def test_gemm_epilogues():
    M, N, K = 300, 256, 128
    X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    R = rnd(M, N, seed=4)
    pre = X.double() @ W.double().t() + b.double()
    out, u = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, epi=1 | 2 | 4, want_c2=True)      # bias + GELU, save pre-act
    assert rel_err(u, pre) < 2e-5
    assert rel_err(out, torch.nn.functional.gelu(pre)) < 2e-5
    out = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, aux=R, ld_aux=N, epi=1 | 8)          # bias + residual
    assert rel_err(out, pre + R.double()) < 2e-5
    out = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, epi=1 | 128)                         # bias + tanh
    assert rel_err(out, torch.tanh(pre)) < 2e-5
    U = rnd(M, N, seed=6)
    ud = U.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    out = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, aux=U, ld_aux=N, epi=16)                     # * gelu'(u)
    assert rel_err(out, (X.double() @ W.double().t()) * ud.grad) < 2e-5


def test_gemm_rejects_bad_leading_dim():
    X, W = rnd(8, 6), rnd(8, 6)
    with pytest.raises(L.RmclError):
        gemm(X, W, 8, 8, 6, 1, 1, L.F32, L.F32)                                                # lda=6 not a multiple of 4


# ------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("relu", [0, 1])
def test_layernorm_fwd_bwd(relu):
    M, D = 371, 768
    x, w, b, dy = rnd(M, D, seed=1), 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3), rnd(M, D, seed=4)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-6)
    if relu:
        y_ref = torch.relu(y_ref)
    y_ref.backward(dy.double())
    y = torch.empty(M, D, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    check(lib.rmcl_layernorm_fwd(P(x), P(w), P(b), F(1e-6), P(y), L.F32, P(mean), P(rstd), M, D, relu, stream()))
    assert rel_err(y, y_ref) < 1e-5
    dx0 = rnd(M, D, seed=9)
    dx = dx0.clone()
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    check(lib.rmcl_layernorm_bwd(P(dy), L.F32, P(x), P(mean), P(rstd), P(w), P(b), P(dx), 1, P(dg), P(db), M, D, relu, stream()))
    assert rel_err(dx - dx0, xd.grad) < 2e-5
    assert rel_err(dg, wd.grad) < 2e-5 and rel_err(db, bd.grad) < 2e-5


@pytest.mark.parametrize("D", [768, 512, 1024])
@pytest.mark.parametrize("add", [0, 1])
@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("wgrad", [0, 1])
def test_layernorm_bwd_variants(D, add, dt, wgrad):
    """Every instantiation family of the row kernel (round 4: loads of a row issued up front - the old values by inline assembly -, DPP row
    sums, guard-free D = 768 form): D = 768 / guarded widths, accumulate or overwrite, fp32 / bf16 dy, with / without dgamma-dbeta; M is not
    a multiple of the rows per workgroup.  Reference: torch autograd of F.layer_norm in fp64 on the values the kernel reads."""
    M = 203
    x, w, b = rnd(M, D, seed=11) * 2 + 0.5, 1 + 0.1 * rnd(D, seed=12), 0.1 * rnd(D, seed=13)
    dy = rnd(M, D, seed=14).to(tdt(dt))
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-6).backward(dy.double())
    y = torch.empty(M, D, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    check(lib.rmcl_layernorm_fwd(P(x), P(w), P(b), F(1e-6), P(y), L.F32, P(mean), P(rstd), M, D, 0, stream()))
    dx0 = rnd(M, D, seed=15)
    dx = dx0.clone()
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    check(lib.rmcl_layernorm_bwd(P(dy), dt, P(x), P(mean), P(rstd), P(w), P(b), P(dx), add, P(dg) if wgrad else None, P(db) if wgrad else None,
                                 M, D, 0, stream()))
    got = dx - dx0 if add else dx
    assert rel_err(got, xd.grad) < 2e-5
    if wgrad:
        assert rel_err(dg, wd.grad) < 2e-5 and rel_err(db, bd.grad) < 2e-5


# ------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("dt,exact", [(L.F32, 1), (L.BF16, 1), (L.BF16, 0)])
@pytest.mark.parametrize("BN", [(3, 185), (2, 64), (1, 241), (2, 130)])
def test_attention_fwd_bwd(dt, exact, BN):
    """exact=1: unfused exact-f32 path (scores materialised); (bf16, exact=0): fused flash-style kernels."""
    B, N = BN
    H, D = 12, 768
    qkv = rnd(B * N, 3 * D, seed=1).to(tdt(dt))
    mask = torch.ones(B, N, dtype=torch.int32)
    mask[0, N - 17:N - 3] = 0                                  # ragged: masked keys in the middle/end
    if B > 1:
        mask[1, 5:9] = 0
    mask = mask.to(DEV)
    dout = rnd(B * N, D, seed=2).to(tdt(dt))
    # reference (vision_transformer.py:309-332) in fp64
    x = qkv.double().requires_grad_(True)
    t = x.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (t[0] @ t[1].transpose(-2, -1)) * 0.125
    s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    o_ref = (s.softmax(-1) @ t[2]).transpose(1, 2).reshape(B * N, D)
    o_ref.backward(dout.double())
    ne = lib.rmcl_attention_scratch_elems(B, H, N)
    out = torch.empty(B * N, D, dtype=tdt(dt), device=DEV)
    probs = torch.empty(ne, dtype=tdt(dt), device=DEV)
    scores = torch.empty(ne, dtype=torch.float32, device=DEV)
    dS = torch.empty(ne, dtype=tdt(dt), device=DEV)
    dqkv = torch.empty(B * N, 3 * D, dtype=tdt(dt), device=DEV)
    check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(out), P(probs), P(scores), B, N, H, dt, exact, stream()))
    tol = 3e-5 if dt == L.F32 else 2e-2
    assert rel_err(out, o_ref) < tol
    # out=None: two-kernel backward (delta from P and dP); out given: ONE kernel (delta = rowsum(dO * O); N <= 192)
    for with_out in (False, True):
        dqkv.fill_(float("nan"))
        check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(dout), P(out) if with_out else None, P(dqkv), P(scores), P(dS),
                                     B, N, H, dt, exact, stream()))
        assert rel_err(dqkv, x.grad) < tol, with_out


ATTN_BWD_TPW_DEFAULT = 1


def test_attention_bwd_one_kernel_full_batch():
    """B = 64 (768 workgroups of 12 waves, 145.5 KiB LDS each): the one-kernel backward against the two-kernel form."""
    B, N, H, D = 64, 185, 12, 768
    dt = L.BF16
    qkv = rnd(B * N, 3 * D, seed=3).to(torch.bfloat16)
    mask = torch.ones(B, N, dtype=torch.int32)
    g = torch.Generator().manual_seed(4)
    for b in range(B):
        mask[b, int(torch.randint(8, 40, (1,), generator=g)):40] = 0     # ragged text, all image tokens valid
    mask = mask.to(DEV)
    dout = rnd(B * N, D, seed=5).to(torch.bfloat16)
    ne = lib.rmcl_attention_scratch_elems(B, H, N)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    probs = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
    scores = torch.empty(ne, dtype=torch.float32, device=DEV)
    dS = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
    check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(out), P(probs), P(scores), B, N, H, dt, 0, stream()))
    res = []
    for with_out in (False, True):
        dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(dout), P(out) if with_out else None, P(dqkv), P(scores), P(dS),
                                     B, N, H, dt, 0, stream()))
        res.append(dqkv.float())
    assert torch.isfinite(res[1]).all()
    assert rel_err(res[1], res[0]) < 1e-2
    # the default walks the 768 problems with 256 workgroups (next problem's lines touched into L2 during phase 1): one workgroup
    # per problem, and a grid that does not divide the problem count, give the same bits
    try:
        for wg, tpw in ((0, 1), (100, 1), (256, 2), (0, 3), (256, 3)):   # (key 9: key tiles per wave; same sums in the same order)
            check(lib.rmcl_tune_set(8, wg))
            check(lib.rmcl_tune_set(9, tpw))
            dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
            check(lib.rmcl_attention_bwd(P(qkv), P(mask), P(probs), P(dout), P(out), P(dqkv), P(scores), P(dS), B, N, H, dt, 0, stream()))
            assert torch.equal(dqkv.float(), res[1]), (wg, tpw)
    finally:
        check(lib.rmcl_tune_set(8, 256))
        check(lib.rmcl_tune_set(9, ATTN_BWD_TPW_DEFAULT))


# --------------------------------------------------------------------------------------- InfoNCE
@pytest.mark.parametrize("form", [1, 0])                       # rmcl_tune_set key 5: folded sub-slices + column-split combine (default) / one slice
@pytest.mark.parametrize("B,Kq", [(4, 1024), (64, 65536), (70, 4096)])
def test_infonce_matches_oracle(B, Kq, form):
    from oracle import rmcl_oracle as O
    T = 0.07
    check(lib.rmcl_tune_set(5, form))
    q = torch.nn.functional.normalize(rnd(B, 128, seed=1), dim=1)
    k = torch.nn.functional.normalize(rnd(B, 128, seed=2) + 2 * q, dim=1)
    queue = rnd(128, Kq, seed=3)
    queue[:, 5] = 30 * q[0]                                        # spike: forces a late max jump / argmax != 0
    qd = q.double().cpu().requires_grad_(True)
    logits = O.infonce_logits(qd, k.double().cpu(), queue.double().cpu(), T)
    loss_ref = O.infonce_loss(logits)
    (loss_ref / 3.0).backward()
    ws = torch.empty(lib.rmcl_infonce_ws_bytes(B, I64(Kq)), dtype=torch.uint8, device=DEV)
    dq, rows, lsum = torch.empty(B, 128, device=DEV), torch.empty(B, 10, device=DEV), torch.zeros(1, device=DEV)
    check(lib.rmcl_infonce_f32(P(q), P(k), P(queue), B, 128, I64(Kq), F(T), F(1.0 / (3.0 * B)), P(dq), P(rows), P(lsum), P(ws),
                               stream()))
    check(lib.rmcl_tune_set(5, 1))
    assert abs(float(lsum) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    assert rel_err(dq.cpu(), qd.grad) < 1e-4
    assert torch.equal(rows[:, 1].cpu().long(), logits.argmax(-1))
    m = O.queue_metrics(q.double().cpu(), k.double().cpu(), queue.double().cpu())
    for j, name in ((3, "pos_dist"), (4, "pos_cosine"), (5, "pos_dot"), (6, "neg_dist"), (7, "neg_cosine"), (8, "neg_dot")):
        assert abs(float(rows[:, j].mean()) - float(m[name])) < 2e-5 * max(1.0, abs(float(m[name]))), name


@pytest.mark.parametrize("metrics", [1, 0])
@pytest.mark.parametrize("B,Kq", [(4, 1024), (64, 65536), (70, 4096)])
def test_infonce_split_bf16_matches_oracle(B, Kq, metrics):
    """The bf16 engine's InfoNCE pass: logits and dq on the bf16 matrix cores with split operands (hi + lo, three products per pair)
    against the fp64 oracle - loss to 2e-4 relative (the exact-f32 form: 1e-4), dq to 5e-4, the same argmax, the same metrics; the
    un-normalised randn queue of the reference (vilt_module.py:92-94) with a spiked column that forces a late max jump."""
    from oracle import rmcl_oracle as O
    T = 0.07
    q = torch.nn.functional.normalize(rnd(B, 128, seed=1), dim=1)
    k = torch.nn.functional.normalize(rnd(B, 128, seed=2) + 2 * q, dim=1)
    queue = rnd(128, Kq, seed=3)
    queue[:, 5] = 30 * q[0]
    qd = q.double().cpu().requires_grad_(True)
    logits = O.infonce_logits(qd, k.double().cpu(), queue.double().cpu(), T)
    loss_ref = O.infonce_loss(logits)
    (loss_ref / 3.0).backward()
    ws = torch.empty(lib.rmcl_infonce_ws_bytes(B, I64(Kq)), dtype=torch.uint8, device=DEV)
    dq, rows, lsum = torch.empty(B, 128, device=DEV), torch.empty(B, 10, device=DEV), torch.zeros(1, device=DEV)
    check(lib.rmcl_infonce_split_bf16(P(q), P(k), P(queue), B, 128, I64(Kq), F(T), F(1.0 / (3.0 * B)), P(dq), P(rows), P(lsum), P(ws), metrics,
                                      stream()))
    assert abs(float(lsum) - float(loss_ref)) < 2e-4 * max(1.0, abs(float(loss_ref))), (float(lsum), float(loss_ref))
    assert rel_err(dq.cpu(), qd.grad) < 5e-4
    top2 = logits.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-2                     # (a 2^-16 product error may flip a near-tie; none in this data)
    assert torch.equal(rows[:, 1].cpu().long()[clear], logits.argmax(-1)[clear]) and int(clear.sum()) >= B - 1
    m = O.queue_metrics(q.double().cpu(), k.double().cpu(), queue.double().cpu())
    for j, name in ((3, "pos_dist"), (4, "pos_cosine"), (5, "pos_dot")) + (((6, "neg_dist"), (7, "neg_cosine"), (8, "neg_dot")) if metrics else ()):
        assert abs(float(rows[:, j].mean()) - float(m[name])) < 1e-4 * max(1.0, abs(float(m[name]))), name
    if not metrics:
        assert float(rows[:, 6:9].abs().max()) == 0.0
    # against the exact-f32 form on the same inputs: the drift the bf16 engine takes on (recorded for DESIGN.md)
    dq2, rows2, lsum2 = torch.empty_like(dq), torch.empty_like(rows), torch.zeros(1, device=DEV)
    check(lib.rmcl_infonce_f32(P(q), P(k), P(queue), B, 128, I64(Kq), F(T), F(1.0 / (3.0 * B)), P(dq2), P(rows2), P(lsum2), P(ws), stream()))
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "infonce_split_drift.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[f"B{B}_Kq{Kq}"] = {"loss_split": float(lsum), "loss_exact": float(lsum2), "loss_fp64": float(loss_ref),
                             "dq_rel_diff_vs_exact": rel_err(dq, dq2), "dq_rel_err_vs_fp64": rel_err(dq.cpu(), qd.grad)}
    json.dump(data, open(path, "w"), indent=1)


# ------------------------------------------------------------------------- PGD / EMA / queue / opt
@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_pgd_step(dt):
    B, per = 3, 144 * 3072
    g = rnd(B, per, seed=1).to(tdt(dt))
    g[2] = 0                                                       # all-zero gradient: 1e-8 floor, delta unchanged
    delta0 = (rnd(B, per, seed=2) * 0.002).clamp(-0.005, 0.005)
    delta = delta0.clone()
    amax = torch.empty(64 * B, dtype=torch.int32, device=DEV)
    check(lib.rmcl_pgd_step(P(g), dt, P(delta), P(amax), B, I64(per), F(0.05), F(0.005), stream()))
    gf = g.float()
    den = gf.abs().amax(dim=1, keepdim=True).clamp_min(1e-8)
    ref = (delta0 + 0.05 * gf / den).clamp(-0.005, 0.005)
    assert float((delta - ref).abs().max()) < 1e-8
    assert float(delta.abs().max()) <= 0.005


@pytest.mark.parametrize("dt,odt", [(L.BF16, L.BF16), (L.F32, L.F32), (L.F32, L.BF16)])
def test_pgd_step_fused(dt, odt):
    """rmcl_pgd_step_fused = rmcl_pgd_step + the operand the loop forms next (pgd_attack_vilt.py:144 / objectives.py:176), bit for bit."""
    B, per = 3, 144 * 3072
    g = rnd(B, per, seed=1).to(tdt(dt))
    g[2] = 0
    base = rnd(B, per, seed=3)
    delta0 = (rnd(B, per, seed=2) * 0.002).clamp(-0.005, 0.005)
    amax = torch.empty(64 * B, dtype=torch.int32, device=DEV)
    ref = delta0.clone()
    check(lib.rmcl_pgd_step(P(g), dt, P(ref), P(amax), B, I64(per), F(0.05), F(0.005), stream()))
    for flags in (0, L.PGD_SUM_PREV, L.PGD_DELTA_ZERO, L.PGD_DELTA_ZERO | L.PGD_SUM_PREV):
        zero = bool(flags & L.PGD_DELTA_ZERO)
        delta = torch.full_like(delta0, float("nan")) if zero else delta0.clone()      # (DELTA_ZERO must not read the buffer)
        out = torch.empty(B, per, dtype=tdt(odt), device=DEV)
        check(lib.rmcl_pgd_step_fused(P(g), dt, P(delta), P(amax), B, I64(per), F(0.05), F(0.005), P(base), P(out), odt, flags, stream()))
        if zero:
            want = torch.zeros_like(delta0)
            check(lib.rmcl_pgd_step(P(g), dt, P(want), P(amax), B, I64(per), F(0.05), F(0.005), stream()))
            old = torch.zeros_like(delta0)
        else:
            want, old = ref, delta0
        assert torch.equal(delta, want)
        op = (base + old) + want if (flags & L.PGD_SUM_PREV) and not zero else base + want
        assert torch.equal(out, op.to(tdt(odt)))
    # no operand: the update alone
    delta = delta0.clone()
    check(lib.rmcl_pgd_step_fused(P(g), dt, P(delta), P(amax), B, I64(per), F(0.05), F(0.005), None, None, odt, 0, stream()))
    assert torch.equal(delta, ref)


def test_ema_and_cast():
    n = 1 << 20
    k, q = rnd(n, seed=1), rnd(n, seed=2)
    lp = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ref = k * 0.999 + q * (1 - 0.999)
    check(lib.rmcl_ema_f32(P(k), P(q), P(lp), F(0.999), I64(n), stream()))
    assert float((k - ref).abs().max()) < 1e-6
    assert torch.equal(lp, k.to(torch.bfloat16))


def test_enqueue_and_bounds():
    Kq, n = 1024, 8
    queue, keys = rnd(128, Kq, seed=1), rnd(n, 128, seed=2)
    ref = queue.clone()
    ref[:, 16:16 + n] = keys.t()
    check(lib.rmcl_enqueue_f32(P(queue), P(keys), n, 128, I64(Kq), I64(16), stream()))
    assert torch.equal(queue, ref)
    with pytest.raises(L.RmclError):
        check(lib.rmcl_enqueue_f32(P(queue), P(keys), n, 128, I64(Kq), I64(Kq - 4), stream()))


def test_adamw_matches_oracle():
    from oracle import rmcl_oracle as O
    n = 4096
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr, mr, vr = p.cpu().clone(), torch.zeros(n), torch.zeros(n)
    seg_end = torch.tensor([1024, 4096], dtype=torch.int64, device=DEV)
    mult = torch.tensor([1.0, 5.0], device=DEV)
    wd = torch.tensor([0.01, 0.0], device=DEV)
    for step in (1, 2, 3):
        check(lib.rmcl_adamw_f32(P(p), P(g), P(m), P(v), None, P(seg_end), P(mult), P(wd), 2, F(1e-3), F(0.9), F(0.98), F(1e-8),
                                 step, F(1.0), I64(n), stream()))
        O.adamw_step(pr[:1024], g.cpu()[:1024], mr[:1024], vr[:1024], step, 1e-3, 0.01)
        O.adamw_step(pr[1024:], g.cpu()[1024:], mr[1024:], vr[1024:], step, 5e-3, 0.0)
    assert float((p.cpu() - pr).abs().max()) < 1e-6


def test_im2patch_roundtrip():
    from oracle import rmcl_oracle as O
    img = rnd(2, 3, 384, 384, seed=1)
    pat = torch.empty(2 * 144, 3072, device=DEV)
    check(lib.rmcl_im2patch_f32(P(img), P(pat), 2, 3, 384, 384, 32, 0, stream()))
    assert torch.equal(pat.cpu(), O.patchify(img.cpu(), 32).reshape(288, 3072))
    back = torch.empty_like(img)
    check(lib.rmcl_im2patch_f32(P(back), P(pat), 2, 3, 384, 384, 32, 1, stream()))
    assert torch.equal(back, img)


# ------------------------------------------------------------------ bf16 MFMA GEMM (fast path)
@pytest.mark.parametrize("cfg", [-1, 1, 30, 50, 60, 70])
@pytest.mark.parametrize("shape", [(11840, 768, 768), (300, 128, 64), (1000, 3072, 768), (256, 768, 3072), (555, 192, 128), (9216, 2304, 192), (400, 768, 256), (11840, 3072, 768)])
def test_gemm_fast_bf16_layouts(shape, cfg):
    """exact=0 routes to the glds/tr-read MFMA kernel; reference = fp64 matmul of the same bf16 inputs,
    tolerance = fp32 accumulation-order noise only (products of bf16 are exact in fp32)."""
    M, N, K = shape
    lib.rmcl_tune_set(0, cfg)                                                      # persistent grid: default 2 workgroups per CU, or 1
    X = rnd(M, K, seed=1).to(torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.05).to(torch.bfloat16)
    ref = X.double() @ W.double().t()
    out = gemm(X, W, M, N, K, 1, 1, L.BF16, L.F32, exact=0)                        # NT
    assert rel_err(out, ref) < 2e-5
    Wn = W.t().contiguous()
    out = gemm(X, Wn, M, N, K, 1, 0, L.BF16, L.F32, exact=0)                       # NN (tr-read B)
    assert rel_err(out, ref) < 2e-5
    out_bf = gemm(X, Wn, M, N, K, 1, 0, L.BF16, L.BF16, exact=0)
    assert rel_err(out_bf, ref) < 1e-2
    if M % 128 == 0:
        Xt = X.t().contiguous()
        base = rnd(M, N, seed=7)
        out = gemm(Xt, Wn, M, N, K, 0, 0, L.BF16, L.F32, exact=0, epi=64, C_init=base)   # TN + accumulate
        assert rel_err(out, ref + base.double()) < 2e-5
    lib.rmcl_tune_set(0, -1)


def test_gemm_fast_epilogues_match_exact_kernel():
    M, N, K = 1111, 256, 192
    X, W, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16), rnd(N, seed=3)
    R = rnd(M, N, seed=4)
    U = rnd(M, N, seed=6).to(torch.bfloat16)
    for epi, kw in ((1 | 2 | 4, dict(bias=b, want_c2=True)), (1 | 8, dict(bias=b, aux=R, ld_aux=N)), (16, dict(aux=U, ld_aux=N))):
        for dto in (L.F32, L.BF16):
            a = gemm(X, W, M, N, K, 1, 1, L.BF16, dto, epi=epi, exact=1, **kw)
            f = gemm(X, W, M, N, K, 1, 1, L.BF16, dto, epi=epi, exact=0, **kw)
            a, f = (a, f) if isinstance(a, tuple) else ((a,), (f,))
            for x, y in zip(a, f):
                assert rel_err(y, x) < (2e-5 if dto == L.F32 else 1e-2)


@pytest.mark.parametrize("b_kc", [1, 0])
def test_gemm_sample_tile_epilogues_match_exact_kernel(b_kc):
    """The 192x192 ping-pong kernel (tune cfg 60; rows per tile 185 for M = 2 * 185 + 7) against the exact-f32 kernel."""
    M, N, K = 377, 384, 320
    X, W, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16), rnd(N, seed=3)
    Wm = W if b_kc else W.t().contiguous()
    R = rnd(M, N, seed=4)
    U = rnd(M, N, seed=6).to(torch.bfloat16)
    base = rnd(M, N, seed=7)
    cases = [(1, dict(bias=b)), (1 | 2 | 4, dict(bias=b, want_c2=True)), (1 | 8, dict(bias=b, aux=R, ld_aux=N)), (16, dict(aux=U, ld_aux=N))]
    try:
        for epi, kw in cases:
            for dto in (L.F32, L.BF16):
                lib.rmcl_tune_set(0, -1)
                a = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, dto, epi=epi, exact=1, **kw)
                lib.rmcl_tune_set(0, 60)
                f = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, dto, epi=epi, exact=0, **kw)
                a, f = (a, f) if isinstance(a, tuple) else ((a,), (f,))
                tol = 2e-4 if epi & (2 | 16) else 2e-5                 # GELU / GELU': polynomial erf, |err| <= 1.3e-4 (rmcl_common.h)
                for x, y in zip(a, f):
                    assert rel_err(y, x) < (tol if dto == L.F32 else 1e-2), (epi, dto)
        lib.rmcl_tune_set(0, 60)
        out = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, L.F32, exact=0, epi=64, C_init=base)           # accumulate
        assert rel_err(out, X.double() @ W.double().t() + base.double()) < 2e-5
    finally:
        lib.rmcl_tune_set(0, -1)


@pytest.mark.parametrize("b_kc", [1, 0])
def test_gemm_wide_tile_epilogues_match_exact_kernel(b_kc):
    """The 192x384 kernel (tune cfg 70) against the exact-f32 kernel: bias, bias+GELU+stash, GELU' epilogues."""
    M, N, K = 377, 768, 320
    X, W, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16), rnd(N, seed=3)
    Wm = W if b_kc else W.t().contiguous()
    U = rnd(M, N, seed=6).to(torch.bfloat16)
    cases = [(0, dict()), (1, dict(bias=b)), (1 | 2 | 4, dict(bias=b, want_c2=True)), (16, dict(aux=U, ld_aux=N))]
    try:
        for epi, kw in cases:
            for dto in (L.F32, L.BF16):
                lib.rmcl_tune_set(0, -1)
                a = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, dto, epi=epi, exact=1, **kw)
                lib.rmcl_tune_set(0, 70)
                f = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, dto, epi=epi, exact=0, **kw)
                a, f = (a, f) if isinstance(a, tuple) else ((a,), (f,))
                tol = 2e-4 if epi & (2 | 16) else 2e-5
                for x, y in zip(a, f):
                    assert rel_err(y, x) < (tol if dto == L.F32 else 1e-2), (epi, dto)
    finally:
        lib.rmcl_tune_set(0, -1)


@pytest.mark.parametrize("seed", list(range(12)))
def test_gemm_pingpong_kernels_random_shapes(seed):
    """192x192 (cfg 60) and 192x384 (cfg 70) kernels on random shapes: ragged last row tiles (any M >= 1), the shortest
    k-loops (K = 128: two k-tiles), multi-round persistence (more tiles than CUs), both B layouts, and the [K][M] x [K][N]
    form; reference = fp64 matmul of the same bf16 operands."""
    import random
    rng = random.Random(1000 + seed)
    M = rng.choice([1, 7, 185, 191, 192, 193, 370, 555, 1000, 2368, rng.randint(1, 3000)])
    N = 192 * rng.choice([1, 2, 4, 6, 8, 12])
    K = 64 * rng.choice([2, 3, 4, 5, 12, 13, 24])
    if seed % 4 == 3:
        M = 192 * rng.choice([40, 62, 70])                       # more than one tile per CU: the persistent k-tile stream
        N = 192 * rng.choice([8, 12])
        K = 64 * rng.choice([2, 3, 4])
    X = rnd(M, K, seed=seed).to(torch.bfloat16)
    W = rnd(N, K, seed=seed + 50, scale=0.05).to(torch.bfloat16)
    b = rnd(N, seed=seed + 99)
    ref = X.double() @ W.double().t() + b.double()
    Wn = W.t().contiguous()
    try:
        for cfg in (60, 70):
            lib.rmcl_tune_set(0, cfg)
            out = gemm(X, W, M, N, K, 1, 1, L.BF16, L.F32, bias=b, epi=1, exact=0)
            assert rel_err(out, ref) < 2e-5, ("NT", cfg, M, N, K)
            out = gemm(X, Wn, M, N, K, 1, 0, L.BF16, L.BF16, bias=b, epi=1, exact=0)
            assert rel_err(out, ref) < 1e-2, ("NN", cfg, M, N, K)
        if M % 192 == 0:                                             # [K][M] x [K][N] (weight-gradient form), accumulate
            lib.rmcl_tune_set(0, 60)
            Xt = X.t().contiguous()
            base = rnd(M, N, seed=seed + 7)
            out = gemm(Xt, Wn, M, N, K, 0, 0, L.BF16, L.F32, exact=0, epi=64, C_init=base)
            assert rel_err(out, X.double() @ W.double().t() + base.double()) < 2e-5, ("TN", M, N, K)
    finally:
        lib.rmcl_tune_set(0, -1)


@pytest.mark.parametrize("cfg,N", [(60, 2304), (70, 3072)])
def test_gemm_pingpong_kernels_are_run_to_run_deterministic(cfg, N):
    """The LDS-DMA / ds_read ordering of the ping-pong kernels rests on counted vmcnt + barriers: a mis-placed wait shows up
    as rare torn tiles, so the same multi-round launch is repeated and every result must be bitwise identical."""
    M, K = 11840, 768
    X = rnd(M, K, seed=3).to(torch.bfloat16)
    W = rnd(N, K, seed=4, scale=0.05).to(torch.bfloat16)
    Wn = W.t().contiguous()
    try:
        lib.rmcl_tune_set(0, cfg)
        for b_kc, Wm in ((1, W), (0, Wn)):
            first = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, L.BF16, exact=0)
            for _ in range(15):
                again = gemm(X, Wm, M, N, K, 1, b_kc, L.BF16, L.BF16, exact=0)
                assert torch.equal(first, again)
        ref = X.double() @ W.double().t()
        assert rel_err(first, ref) < 1e-2
    finally:
        lib.rmcl_tune_set(0, -1)


def test_gemm_dual_tile_epilogues_match_exact_kernel():
    """The 192x192x32 kernel with two 4-wave workgroups per CU (gemm_dp.hip, tune cfg 80) against the exact-f32 kernel: every
    epilogue it takes, ragged last row tile (M = 2 * 185 + 7), K = 320 = 10 k-steps of 32."""
    M, N, K = 377, 384, 320
    X, W, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16), rnd(N, seed=3)
    R = rnd(M, N, seed=4)
    U = rnd(M, N, seed=6).to(torch.bfloat16)
    cases = [(0, dict(), (L.F32, L.BF16)), (1, dict(bias=b), (L.F32, L.BF16)), (1 | 2 | 4, dict(bias=b, want_c2=True), (L.F32, L.BF16)),
             (1 | 8, dict(bias=b, aux=R, ld_aux=N), (L.F32,)), (16, dict(aux=U, ld_aux=N), (L.F32, L.BF16))]
    try:
        for epi, kw, dtos in cases:
            for dto in dtos:
                lib.rmcl_tune_set(0, -1)
                a = gemm(X, W, M, N, K, 1, 1, L.BF16, dto, epi=epi, exact=1, **kw)
                lib.rmcl_tune_set(0, 80)
                f = gemm(X, W, M, N, K, 1, 1, L.BF16, dto, epi=epi, exact=0, **kw)
                a, f = (a, f) if isinstance(a, tuple) else ((a,), (f,))
                tol = 2e-4 if epi & (2 | 16) else 2e-5
                for x, y in zip(a, f):
                    assert rel_err(y, x) < (tol if dto == L.F32 else 1e-2), (epi, dto)
    finally:
        lib.rmcl_tune_set(0, -1)


@pytest.mark.parametrize("seed", list(range(10)))
def test_gemm_dual_tile_random_shapes(seed):
    """gemm_dp.hip (cfg 80) on random shapes: ragged row tiles (any M >= 1), the shortest k-loops (K = 64: two k-steps), one tile,
    fewer tiles than workgroups, and several tiles per workgroup (the k-step stream across tiles); fp64 reference."""
    import random
    rng = random.Random(2000 + seed)
    M = rng.choice([1, 7, 185, 191, 192, 193, 370, 555, 1000, 2368, rng.randint(1, 3000)])
    N = 192 * rng.choice([1, 2, 4, 6, 8, 12])
    K = 32 * rng.choice([2, 3, 4, 5, 12, 13, 24, 25])
    if seed % 3 == 2:
        M = 192 * rng.choice([40, 62, 70])
        N = 192 * rng.choice([8, 12, 16])
        K = 32 * rng.choice([2, 3, 8])
    X = rnd(M, K, seed=seed).to(torch.bfloat16)
    W = rnd(N, K, seed=seed + 50, scale=0.05).to(torch.bfloat16)
    b = rnd(N, seed=seed + 99)
    ref = X.double() @ W.double().t() + b.double()
    try:
        lib.rmcl_tune_set(0, 80)
        out = gemm(X, W, M, N, K, 1, 1, L.BF16, L.F32, bias=b, epi=1, exact=0)
        assert rel_err(out, ref) < 2e-5, ("f32", M, N, K)
        out = gemm(X, W, M, N, K, 1, 1, L.BF16, L.BF16, bias=b, epi=1, exact=0)
        assert rel_err(out, ref) < 1e-2, ("bf16", M, N, K)
    finally:
        lib.rmcl_tune_set(0, -1)


@pytest.mark.parametrize("kblk", [1, 2, 3])
def test_gemm_dual_tile_k_blocked_operands(kblk):
    """gemm_dp.hip with operands stored k-blocked [K/32][rows][32] (whole 128-byte lines per k-step): same result as row-major."""
    M, N, K = 1111, 576, 416
    X, W, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16), rnd(N, seed=3)
    blk = lambda t: t.view(t.shape[0], K // 32, 32).permute(1, 0, 2).contiguous()
    A = blk(X) if kblk & 1 else X
    Bw = blk(W) if kblk & 2 else W
    out = torch.empty(M, N, device=DEV)
    check(lib.rmcl_gemm_kblk(P(A), P(Bw), P(out), None, P(b), None, M, N, K, N, 0, 1, L.F32, kblk, stream()))
    assert rel_err(out, X.double() @ W.double().t() + b.double()) < 2e-5


@pytest.mark.parametrize("N,K", [(2304, 768), (3072, 768), (768, 3072)])
def test_gemm_dual_tile_is_run_to_run_deterministic(N, K):
    """Counted vmcnt + one barrier per k-step order the LDS-DMA against the fragment reads; a mis-placed wait shows as rare torn
    tiles, and here two workgroups per CU drift against each other: the step's launches repeated, bitwise identical every time."""
    M = 11840
    X = rnd(M, K, seed=3).to(torch.bfloat16)
    W = rnd(N, K, seed=4, scale=0.05).to(torch.bfloat16)
    try:
        lib.rmcl_tune_set(0, 80)
        first = gemm(X, W, M, N, K, 1, 1, L.BF16, L.BF16, exact=0)
        for _ in range(15):
            assert torch.equal(first, gemm(X, W, M, N, K, 1, 1, L.BF16, L.BF16, exact=0))
        assert rel_err(first, X.double() @ W.double().t()) < 1e-2
    finally:
        lib.rmcl_tune_set(0, -1)


def test_grad_ready_wait_rejects_bad_layers():
    """rmcl_grad_ready_wait: error codes, never a crash (layers beyond the last backward's depth / negative)."""
    import ctypes as C
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rmcl_grad_ready_wait(-1, st) != 0
    assert lib.rmcl_grad_ready_wait(64, st) != 0
    assert b"grad_ready_wait" in lib.rmcl_last_error()


def test_gemm_skinny_heads_shapes():
    """M <= 256 fp32 problems (pooler / MoCo head) run on the skinny 64x16-tile kernel."""
    for M, N, K in ((64, 768, 768), (64, 128, 768), (2, 768, 768), (70, 40, 256)):
        X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
        pre = X.double() @ W.double().t() + b.double()
        out = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, epi=1 | 128)
        assert rel_err(out, torch.tanh(pre)) < 2e-5
        base = rnd(M, N, seed=5)
        out = gemm(X, W.t().contiguous(), M, N, K, 1, 0, L.F32, L.F32, epi=64, C_init=base)
        assert rel_err(out, X.double() @ W.double().t() + base.double()) < 2e-5


@pytest.mark.parametrize("form", [0, 1])
def test_gemm_skinny_epilogues_and_forms(form):
    """Both forms of the skinny fp32 GEMM (rmcl_tune_set key 6: 0 = waves split the rows, else waves split K where K % 64 == 0)
    with the epilogues of the encoder's cls-only tail: bias + GELU + saved pre-activation, bias + residual, GELU' of a saved
    pre-activation (data gradient, B stored [K][N]), and the 8192-wide Barlow-Twins shape (32-column tiles)."""
    check(lib.rmcl_tune_set(6, form))
    try:
        gelu = lambda t: 0.5 * t * (1 + torch.erf(t / 2 ** 0.5))
        dgelu = lambda t: 0.5 * (1 + torch.erf(t / 2 ** 0.5)) + t * torch.exp(-0.5 * t * t) / (2 * torch.pi) ** 0.5
        for M, N, K in ((64, 3072, 768), (64, 768, 3072), (37, 768, 768), (64, 128, 768), (64, 8192, 2048), (130, 8192, 256)):
            X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
            pre = X.double() @ W.double().t() + b.double()
            out, c2 = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, epi=1 | 2 | 4, want_c2=True)          # bias, GELU, save pre-activation
            assert rel_err(c2, pre) < 2e-5 and rel_err(out, gelu(pre)) < 2e-5, (M, N, K)
            res = rnd(M, N, seed=7)
            out = gemm(X, W, M, N, K, 1, 1, L.F32, L.F32, bias=b, aux=res, ld_aux=N, epi=1 | 8)
            assert rel_err(out, pre + res.double()) < 2e-5, (M, N, K)
            u = rnd(M, N, seed=9)
            out = gemm(X, W.t().contiguous(), M, N, K, 1, 0, L.F32, L.F32, aux=u, ld_aux=N, epi=16)          # (X W^T) * gelu'(u), B as [K][N]
            assert rel_err(out, (X.double() @ W.double().t()) * dgelu(u.double())) < 2e-5, (M, N, K)
        # short-K outer-product form C += alpha A^T B (weight gradients over B rows, the Barlow-Twins correlation)
        for K, M, N in ((64, 768, 3072), (4, 256, 384), (37, 132, 200), (64, 1000, 520)):
            A, Bm, base = rnd(K, M, seed=11), rnd(K, N, seed=12), rnd(M, N, seed=13)
            out = gemm(A, Bm, M, N, K, 0, 0, L.F32, L.F32, lda=M, ldb=N, epi=64, alpha=0.5, C_init=base)
            assert rel_err(out, 0.5 * A.double().t() @ Bm.double() + base.double()) < 2e-5, (K, M, N)
            out = gemm(A, Bm, M, N, K, 0, 0, L.F32, L.F32, lda=M, ldb=N)
            assert rel_err(out, A.double().t() @ Bm.double()) < 2e-5, (K, M, N)
    finally:
        check(lib.rmcl_tune_set(6, -1))


# ------------------------------------------------------------------------- LayerNorm folded into the consuming GEMM
@pytest.mark.parametrize("cfg", [-1, 80])
@pytest.mark.parametrize("N2,gelu", [(2304, 0), (3072, 1)])
def test_layernorm_fold_gemm_pair_matches_layernorm_then_linear(N2, gelu, cfg):
    """Producer (fp32 output + residual, bf16 copy, per-row partial sums) and consumer (rstd * (xb W'^T - mean * s) + c) of
    the LayerNorm fold at the step's shapes, against  LN(y) W2^T + b2  (then GELU) computed in fp64 from the producer's
    own fp32 output."""
    M, D, K1 = 11840, 768, 768
    lib.rmcl_tune_set(0, cfg)                                    # 80: both GEMMs on gemm_dp.hip (two 4-wave workgroups per CU)
    A = rnd(M, K1, seed=1).to(torch.bfloat16)
    W1 = rnd(D, K1, seed=2, scale=0.05).to(torch.bfloat16)
    b1 = rnd(D, seed=3)
    res = rnd(M, D, seed=4, scale=2.0)
    out = torch.empty(M, D, device=DEV)
    outb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    nparts = 4 * (D // 192)
    part = torch.zeros(M, nparts, 2, device=DEV)
    check(lib.rmcl_linear_rowstat(P(A), P(W1), P(b1), P(res), P(out), P(outb), P(part), M, D, K1, stream()))
    y_ref = A.double() @ W1.double().t() + b1.double() + res.double()
    assert rel_err(out, y_ref) < 2e-5
    assert torch.equal(outb, out.to(torch.bfloat16))
    np.testing.assert_allclose(part[:, :, 0].sum(1).cpu().numpy(), out.double().sum(1).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(part[:, :, 1].sum(1).cpu().numpy(), (out.double() ** 2).sum(1).cpu().numpy(), rtol=1e-5)
    # consumer
    gamma, beta = 1.0 + rnd(D, seed=5, scale=0.1), rnd(D, seed=6, scale=0.1)
    W2, b2 = rnd(N2, D, seed=7, scale=0.05), rnd(N2, seed=8, scale=0.1)
    wf = (W2 * gamma).to(torch.bfloat16)
    s = wf.float().sum(1)
    c = W2 @ beta + b2
    o2 = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV) if gelu else None
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    check(lib.rmcl_linear_lnfold(P(outb), P(wf), P(s), P(c), P(part), nparts, P(o2), P(pre), M, N2, D, gelu, F(1e-6), P(mean), P(rstd),
                                 stream()))
    lib.rmcl_tune_set(0, -1)
    yd = out.double()
    mu, var = yd.mean(1, keepdim=True), yd.var(1, unbiased=False, keepdim=True)
    ln = (yd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    z = ln @ W2.double().t() + b2.double()
    np.testing.assert_allclose(mean.cpu().numpy(), mu[:, 0].cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(rstd.cpu().numpy(), (1 / torch.sqrt(var + 1e-6))[:, 0].cpu().numpy(), rtol=1e-4)
    if gelu:
        assert rel_err(pre, z) < 1.2e-2
        z = torch.nn.functional.gelu(z)
    assert rel_err(o2, z) < 1.2e-2          # bf16 operands and output (the separate LayerNorm -> bf16 -> GEMM path has the same class)


@pytest.mark.parametrize("case", ["init_like", "outlier_channel", "offset_10", "offset_50", "offset_50_drift"])
def test_layernorm_fold_precision_on_offset_rows(case):
    """The fold on residual streams that are NOT zero-mean (a trained checkpoint's can be: large common-mode offset, outlier
    channels).  Round 3's form fed bf16(x) of the RAW stream to the folded GEMM: its rounding scales with |x|, i.e. error
    ~ 2^-9 * |mean| / std in units of the normalised activations (1.8e-2 at 10 sigma, 7.1e-2 at 50 sigma against 3.5e-3 for the
    separate fp32 LayerNorm -> bf16 path).  Round 4: the producer stores bf16(x - c) and the partial sums of (x - c), c = the row
    mean of the row's PREVIOUS LayerNorm (LN(x) = LN(x - c) exactly), which is what the encoder passes run.  Here c = the mean of the
    producer's residual INPUT (`prev`) while the producer adds a branch of unit scale on top - in "offset_50_drift" a branch whose own
    row means drift by up to +-2 sigma, so c is visibly off the true mean.  Both paths against fp64; the centred fold must stay within
    2x of the separate path in EVERY case; the uncentred form is measured beside it and recorded."""
    M, D, N2, K1 = 1480, 768, 2304, 128
    g = torch.Generator().manual_seed(11)
    prev = torch.randn(M, D, generator=g)
    if case == "outlier_channel":
        prev[:, 17] += 60.0                                          # one massive-activation channel
    elif case.startswith("offset"):
        prev += float(case.split("_")[1])
    A = (torch.randn(M, K1, generator=g) * 0.3).to(torch.bfloat16)
    W1 = (torch.randn(D, K1, generator=g) * 0.3).to(torch.bfloat16)   # branch = A W1^T: rows of standard deviation ~1
    b1 = torch.zeros(D)
    prev, A, W1, b1 = prev.to(DEV), A.to(DEV), W1.to(DEV), b1.to(DEV)
    drift = ((torch.rand(M, 1, generator=g) * 4 - 2).to(DEV) if case.endswith("drift") else torch.zeros(M, 1, device=DEV))
    prev_in = prev + drift                                           # what the producer adds its branch to; c is taken BEFORE the drift
    center = prev.mean(1).contiguous()
    out, outb = torch.empty(M, D, device=DEV), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    nparts = 4 * (D // 192)
    part = torch.zeros(M, nparts, 2, device=DEV)
    gamma, beta = 1.0 + rnd(D, seed=5, scale=0.1), rnd(D, seed=6, scale=0.1)
    W2, b2 = rnd(N2, D, seed=7, scale=0.05), rnd(N2, seed=8, scale=0.1)
    wf = (W2 * gamma).to(torch.bfloat16)
    s_, c_ = wf.float().sum(1), W2 @ beta + b2
    res = {}
    lib.rmcl_tune_set(0, 60)                                         # M = 8 x 185: the 192-row tile kernels by request
    try:
        for name, cen in (("centred", center), ("uncentred", None)):
            check(lib.rmcl_linear_rowstat_c(P(A), P(W1), P(b1), P(prev_in), P(cen), P(out), P(outb), P(part), M, D, K1, stream()))
            x = out.clone()
            o_fold = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
            mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            check(lib.rmcl_linear_lnfold_c(P(outb), P(wf), P(s_), P(c_), P(part), nparts, P(cen), P(o_fold), None, M, N2, D, 0, F(1e-6),
                                           P(mean), P(rstd), stream()))
            res[name] = (o_fold, mean.clone(), rstd.clone())
        # separate path: fp32 LayerNorm -> bf16 -> bf16 GEMM + bias (what the FULL pass runs)
        ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        m2, r2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        check(lib.rmcl_layernorm_fwd(P(x), P(gamma), P(beta), F(1e-6), P(ln), L.BF16, P(m2), P(r2), M, D, 0, stream()))
        o_sep = gemm(ln, W2.to(torch.bfloat16), M, N2, D, 1, 1, L.BF16, L.BF16, bias=b2, epi=1, exact=0)
    finally:
        lib.rmcl_tune_set(0, -1)
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    ref = ((xd - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()) @ W2.double().t() + b2.double()
    e_fold, e_raw, e_sep = rel_err(res["centred"][0], ref), rel_err(res["uncentred"][0], ref), rel_err(o_sep, ref)
    ratio = float((mu.abs() / var.sqrt()).mean())
    # the stashed statistics (the backward's LayerNorm reads them) are those of the TRUE rows, whatever the centre
    np.testing.assert_allclose(res["centred"][1].cpu().numpy(), mu[:, 0].cpu().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(res["centred"][2].cpu().numpy(), (1 / torch.sqrt(var + 1e-6))[:, 0].cpu().numpy(), rtol=2e-4)
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lnfold_precision.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[case] = {"fold_centred_rel_err": e_fold, "fold_uncentred_rel_err": e_raw, "separate_rel_err": e_sep, "mean_over_std": ratio}
    json.dump(data, open(path, "w"), indent=1)
    assert e_sep < 1.2e-2
    assert e_fold < max(6e-3, 2.0 * e_sep), (case, e_fold, e_sep, e_raw)       # the round-3 review's bar: <= 2x the separate path
    if case in ("offset_10", "offset_50"):
        assert e_raw > 2.0 * e_fold, (case, e_raw, e_fold)                     # (the uncentred form is what loses the spread)


@pytest.mark.parametrize("release", [1, 0])
def test_chained_row_tile_gemms_equal_the_separate_launches(release):
    """gemm_chain_kernel (round-4 prototype, include/rmcl.h rmcl_gemm_chain): proj producer (+residual, centred row partials) -> fc1
    (LayerNorm-folded, GELU, pre-activation stash) -> fc2 producer -> qkv (LayerNorm-folded) of one row tile inside ONE launch, groups of
    four workgroups synchronised by global-memory tickets, against the same four GEMMs as separate launches: the same tiles, k order and
    epilogue code, so every output must be BIT-identical - with the agent-scope release per stage and with the same-XCD hand-off (the blocks'
    XCC ids are read back: the four blocks of every group must share an XCD for that form to be valid), over two consecutive launches on
    one ticket buffer, the second with three stages (tickets of unused stages advance too)."""
    import ctypes as C

    class Stage(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("A", "W", "bias", "residual", "out", "out2", "part", "center", "ln_s", "ln_c", "mean", "rstd")] + \
                   [("N", C.c_int), ("K", C.c_int), ("epi", C.c_int), ("nparts", C.c_int), ("ln_eps", C.c_float)]

    M, D, Hm = 11840, 768, 3072
    att = rnd(M, D, seed=1).to(torch.bfloat16)
    x = rnd(M, D, seed=2, scale=2.0) + 3.0
    Wo, bo = rnd(D, D, seed=3, scale=0.05).to(torch.bfloat16), rnd(D, seed=4)
    W2, b2 = rnd(D, Hm, seed=5, scale=0.03).to(torch.bfloat16), rnd(D, seed=6)
    g1, be1, W1, b1 = 1.0 + rnd(D, seed=7, scale=0.1), rnd(D, seed=8, scale=0.1), rnd(Hm, D, seed=9, scale=0.05), rnd(Hm, seed=10, scale=0.1)
    g2, be2, Wq, bq = 1.0 + rnd(D, seed=11, scale=0.1), rnd(D, seed=12, scale=0.1), rnd(3 * D, D, seed=13, scale=0.05), rnd(3 * D, seed=14, scale=0.1)
    wf1, wfq = (W1 * g1).to(torch.bfloat16), (Wq * g2).to(torch.bfloat16)
    s1, c1, sq, cq = wf1.float().sum(1), W1 @ be1 + b1, wfq.float().sum(1), Wq @ be2 + bq
    cen = x.mean(1).contiguous()                                    # centre of the first producer: the row mean of its residual input
    E = L
    nparts = 4 * (D // 192)

    def buffers():
        f32 = lambda *s: torch.zeros(*s, device=DEV)
        b16 = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV)
        return dict(y=f32(M, D), yb=b16(M, D), p1=f32(M, nparts, 2), h=b16(M, Hm), u=b16(M, Hm), m2=f32(M), r2=f32(M), xo=f32(M, D), xob=b16(M, D),
                    p2=f32(M, nparts, 2), qkv=b16(M, 3 * D), m1=f32(M), r1=f32(M))

    def separate(o, n):
        check(lib.rmcl_linear_rowstat_c(P(att), P(Wo), P(bo), P(x), P(cen), P(o["y"]), P(o["yb"]), P(o["p1"]), M, D, D, stream()))
        check(lib.rmcl_linear_lnfold_c(P(o["yb"]), P(wf1), P(s1), P(c1), P(o["p1"]), nparts, P(cen), P(o["h"]), P(o["u"]), M, Hm, D, 1, F(1e-6),
                                       P(o["m2"]), P(o["r2"]), stream()))
        check(lib.rmcl_linear_rowstat_c(P(o["h"]), P(W2), P(b2), P(o["y"]), P(o["m2"]), P(o["xo"]), P(o["xob"]), P(o["p2"]), M, D, Hm, stream()))
        if n == 4:
            check(lib.rmcl_linear_lnfold_c(P(o["xob"]), P(wfq), P(sq), P(cq), P(o["p2"]), nparts, P(o["m2"]), P(o["qkv"]), None, M, 3 * D, D, 0, F(1e-6),
                                           P(o["m1"]), P(o["r1"]), stream()))

    def stages(o):
        q = lambda t: None if t is None else t.data_ptr()
        return (Stage * 4)(
            Stage(q(att), q(Wo), q(bo), q(x), q(o["y"]), q(o["yb"]), q(o["p1"]), q(cen), None, None, None, None, D, D, E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTAT, 0, 1e-6),
            Stage(q(o["yb"]), q(wf1), None, None, q(o["h"]), q(o["u"]), q(o["p1"]), q(cen), q(s1), q(c1), q(o["m2"]), q(o["r2"]), Hm, D,
                  E.EPI_LNFOLD | E.EPI_GELU | E.EPI_SAVE_PREACT, nparts, 1e-6),
            Stage(q(o["h"]), q(W2), q(b2), q(o["y"]), q(o["xo"]), q(o["xob"]), q(o["p2"]), q(o["m2"]), None, None, None, None, D, Hm,
                  E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTAT, 0, 1e-6),
            Stage(q(o["xob"]), q(wfq), None, None, q(o["qkv"]), None, q(o["p2"]), q(o["m2"]), q(sq), q(cq), q(o["m1"]), q(o["r1"]), 3 * D, D, E.EPI_LNFOLD, nparts, 1e-6))

    tickets = torch.zeros(4 * 64 + 1, dtype=torch.int32, device=DEV)
    xcc = torch.full((256,), -1, dtype=torch.int32, device=DEV)
    pure = lambda: bool((xcc.cpu().view(8, 4, 8) == xcc.cpu().view(8, 4, 8)[:, :1, :]).all())     # [8 groups of a block row][member][xcd slot]
    if not release:
        # the same-XCD hand-off is valid only where blocks b, b + 8, b + 16, b + 24 share an XCD (observed dispatch order, not a HIP guarantee):
        # read the placement with a placement-independent launch first and skip this form where the box deals blocks differently
        probe = buffers()
        check(lib.rmcl_gemm_chain(stages(probe), 1, M, P(tickets), C.c_uint32(1), 1, P(xcc), None, 0, stream()), "gemm_chain")
        torch.cuda.synchronize()
        if not pure():
            pytest.skip("blocks b, b + 8, b + 16, b + 24 do not share an XCD on this box: the same-XCD hand-off form does not apply")
        tickets.zero_()
    lib.rmcl_tune_set(0, 60)                                         # the separate launches on the same 192 x 192 tiles
    try:
        for epoch, n in ((1, 4), (2, 3), (3, 4)):
            ref, got = buffers(), buffers()
            separate(ref, n)
            check(lib.rmcl_gemm_chain(stages(got), n, M, P(tickets), C.c_uint32(epoch), release, P(xcc), None, 0, stream()), "gemm_chain")
            torch.cuda.synchronize()
            assert int(tickets[4 * 64]) == 0, "a workgroup gave up waiting for a ticket"
            assert tickets[:4 * 64].view(4, 64)[:, :62].eq(4 * epoch).all(), tickets[:4 * 64].view(4, 64)[:, :8]
            for k in ref:
                if n == 3 and k in ("qkv", "m1", "r1"):
                    continue
                assert torch.equal(ref[k], got[k]), (epoch, n, k)
    finally:
        lib.rmcl_tune_set(0, -1)
    assert release or pure()


@pytest.mark.parametrize("wire", ["f32", "bf16"])
def test_shard_sum_is_the_rank_order_sum(wire):
    """rmcl_shard_sum (owner side of the direct reduce-scatter): W pieces of one slice -> fp32 sum in rank order, and the
    sum in the wire type."""
    W, n = 8, 256 * 37
    g = torch.Generator().manual_seed(3)
    pieces = torch.randn(W, n, generator=g).to(DEV)
    if wire == "bf16":
        pieces = pieces.to(torch.bfloat16)
    out32 = torch.empty(n, device=DEV)
    outw = torch.empty(n, dtype=pieces.dtype, device=DEV)
    check(lib.rmcl_shard_sum(P(pieces), L.BF16 if wire == "bf16" else L.F32, W, n, P(out32), P(outw), stream()))
    want = pieces[0].float().clone()
    for w in range(1, W):
        want += pieces[w].float()
    assert torch.equal(out32, want)
    assert torch.equal(outw, want.to(pieces.dtype))
