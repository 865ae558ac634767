"""Review item 2 of round 4: does a row tile CARRIED THROUGH proj -> fc1 -> fc2 -> next qkv inside one launch (gemm_chain_kernel, csrc/gemm_st.hip;
groups of four workgroups on one XCD, global-memory tickets per (stage, row tile)) beat the same GEMMs as separate launches?

Forward of 12 encoder layers through the C ABI at B = 64 (M = 11 840), random bf16 weights, as in tools/two_stream_test.py:
  separate launches : qkv, fused attention, proj producer (+residual, row statistics), fc1 (+GELU), fc2 producer          - one chain / two B = 32 lanes
  chained           : qkv_0, then per layer  [fused attention]  [ONE launch: proj -> fc1 -> fc2 -> qkv of the next layer]   - one chain / two lanes
with the hand-off in its placement-independent form (agent-scope release per stage) and in the same-XCD form (plain stores stay in the XCD's L2);
outputs of the chained forward are compared BIT FOR BIT with the separate launches (same tiles, same k order, same epilogue code), the blocks' XCC ids
are read back to check the placement assumption, and one workgroup's per-stage wall-clock stamps are printed.

Gate: chained >= 8 % faster than the two-lane separate-launch forward, else record and stop."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import L, lib, check, P, I64, F, DEV

D, H, N, LAYERS = 768, 12, 185, int(os.environ.get("CHAIN_LAYERS", "12"))
g = torch.Generator().manual_seed(0)
Wqkv = (torch.randn(3 * D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
Wo = (torch.randn(D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
W1 = (torch.randn(4 * D, D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
W2 = (torch.randn(D, 4 * D, generator=g) * 0.03).to(DEV).to(torch.bfloat16)
bq, bo, b1, b2 = ((torch.randn(n, generator=g) * 0.02).to(DEV) for n in (3 * D, D, 4 * D, D))
lib.rmcl_attention_scratch_elems.restype = ctypes.c_int64


class Stage(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("out2", ctypes.c_void_p), ("part", ctypes.c_void_p), ("center", ctypes.c_void_p), ("ln_s", ctypes.c_void_p), ("ln_c", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int), ("epi", ctypes.c_int),
                ("nparts", ctypes.c_int), ("ln_eps", ctypes.c_float)]


def ptr(t):
    return None if t is None else t.data_ptr()


class Chain:
    def __init__(self, B, seed):
        self.B, self.M = B, B * N
        M = self.M
        gg = torch.Generator().manual_seed(seed)
        self.x0 = torch.randn(M, D, generator=gg).to(DEV)
        self.xb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        self.x32 = torch.empty(M, D, device=DEV)
        self.qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
        self.att = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        self.h = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=DEV)
        self.y32 = torch.empty(M, D, device=DEV)
        self.yb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        self.prt = torch.empty(M, 16, 2, device=DEV)
        self.prt2 = torch.empty(M, 16, 2, device=DEV)
        self.mask = torch.ones(B, N, dtype=torch.int32, device=DEV)
        ne = lib.rmcl_attention_scratch_elems(B, H, N)
        self.probs = torch.empty(ne, dtype=torch.bfloat16, device=DEV)
        self.scores = torch.empty(max(ne, 1), dtype=torch.float32, device=DEV)
        self.tickets = torch.zeros(4 * 64 + 1, dtype=torch.int32, device=DEV)
        self.epoch = 0
        self.xcc = torch.full((256,), -1, dtype=torch.int32, device=DEV)
        self.stamps = torch.zeros(16, dtype=torch.int64, device=DEV)
        E = L
        mk = lambda A, W, b, res, out, out2, part, Nn, K, epi: Stage(ptr(A), ptr(W), ptr(b), ptr(res), ptr(out), ptr(out2), ptr(part), None, None, None,
                                                                     None, None, Nn, K, epi, 0, 1e-6)
        self.stages = (Stage * 4)(
            mk(self.att, Wo, bo, self.x32, self.y32, self.yb, self.prt, D, D, E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTAT),
            mk(self.yb, W1, b1, None, self.h, None, None, 4 * D, D, E.EPI_BIAS | E.EPI_GELU),
            mk(self.h, W2, b2, self.y32, self.x32, self.xb, self.prt2, D, 4 * D, E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTAT),
            mk(self.xb, Wqkv, bq, None, self.qkv, None, None, 3 * D, D, E.EPI_BIAS))

    def reset(self):
        self.x32.copy_(self.x0)
        self.xb.copy_(self.x0.to(torch.bfloat16))

    def st(self, s):
        return ctypes.c_void_p(s.cuda_stream)

    def qkv_gemm(self, s):
        check(lib.rmcl_gemm(P(self.xb), P(Wqkv), P(self.qkv), None, P(bq), None, self.M, 3 * D, D, I64(D), I64(D), 3 * D, 0, F(1.0), L.EPI_BIAS, 1, L.BF16,
                            L.BF16, 1, 1, 0, self.st(s)))

    def attention(self, s):
        check(lib.rmcl_attention_fwd(P(self.qkv), P(self.mask), P(self.att), P(self.probs), P(self.scores), self.B, N, H, L.BF16, 0, self.st(s)))

    def layer_separate(self, s):
        M, st = self.M, self.st(s)
        self.qkv_gemm(s)
        self.attention(s)
        check(lib.rmcl_linear_rowstat(P(self.att), P(Wo), P(bo), P(self.x32), P(self.y32), P(self.yb), P(self.prt), M, D, D, st))
        check(lib.rmcl_gemm(P(self.yb), P(W1), P(self.h), None, P(b1), None, M, 4 * D, D, I64(D), I64(D), 4 * D, 0, F(1.0), L.EPI_BIAS | L.EPI_GELU, 1,
                            L.BF16, L.BF16, 1, 1, 0, st))
        check(lib.rmcl_linear_rowstat(P(self.h), P(W2), P(b2), P(self.y32), P(self.x32), P(self.xb), P(self.prt2), M, D, 4 * D, st))

    def chain_launch(self, s, flags, n=4, stamp_wg=-1):
        self.epoch += 1
        check(lib.rmcl_gemm_chain(self.stages, n, self.M, P(self.tickets), ctypes.c_uint32(self.epoch), flags, P(self.xcc),
                                  P(self.stamps) if stamp_wg >= 0 else None, max(stamp_wg, 0), self.st(s)), "gemm_chain")

    def forward_separate(self, s):
        for _ in range(LAYERS):
            self.layer_separate(s)

    def forward_chained(self, s, flags, stamp_wg=-1):
        self.qkv_gemm(s)
        for l in range(LAYERS):
            self.attention(s)
            self.chain_launch(s, flags, 4 if l + 1 < LAYERS else 3, stamp_wg)      # (the last layer has no next qkv)


def timed(fn, reps=7):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], ts[0]


def two(fn0, fn1):
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur)
    s1.wait_stream(cur)
    fn0(s0)
    fn1(s1)
    cur.wait_stream(s0)
    cur.wait_stream(s1)


def interleaved(a, b, fa, fb):
    """enqueue the two lanes alternately, layer by layer, so that neither queue runs dry"""
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur)
    s1.wait_stream(cur)
    fa(a, s0, "head")
    fb(b, s1, "head")
    for l in range(LAYERS):
        fa(a, s0, l)
        fb(b, s1, l)
    cur.wait_stream(s0)
    cur.wait_stream(s1)


def sep_step(c, s, l):
    if l != "head":
        c.layer_separate(s)


def chain_step(flags):
    def f(c, s, l):
        if l == "head":
            c.qkv_gemm(s)
        else:
            c.attention(s)
            c.chain_launch(s, flags, 4 if l + 1 < LAYERS else 3)
    return f


full, h0, h1 = Chain(64, 1), Chain(32, 2), Chain(32, 3)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream()

# ---- correctness: chained == separate launches, bit for bit (192x192 tiles for every GEMM: rmcl_tune_set(0, 60)) ----
check(lib.rmcl_tune_set(0, 60))
results = {}
for name, fn in (("separate", lambda: full.forward_separate(cur)), ("chained, release per stage", lambda: full.forward_chained(cur, 1)),
                 ("chained, same-XCD hand-off", lambda: full.forward_chained(cur, 0))):
    full.reset()
    full.qkv.zero_(); full.h.zero_(); full.y32.zero_()
    fn()
    torch.cuda.synchronize()
    results[name] = (full.x32.clone(), full.xb.clone(), full.att.clone(), full.h.clone())
    err = int(full.tickets[4 * 64])
    print(f"{name:32s}: |x| = {float(full.x32.abs().mean()):.4f}  give-up word = {err}")
    if err:
        sys.exit("a chained launch gave up waiting for a ticket: hand-off bug")
ref = results["separate"]
for name in list(results)[1:]:
    same = all(torch.equal(a, b) for a, b in zip(ref, results[name]))
    worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ref, results[name]))
    print(f"  {name:32s} == separate launches: {same}  (max |diff| {worst:.3e})")
xcc = full.xcc.cpu().tolist()
groups = [[xcc[b], xcc[b + 8], xcc[b + 16], xcc[b + 24]] for base in range(0, 256, 32) for b in range(base, base + 8)]
pure = sum(1 for q in groups if len(set(q)) == 1)
print(f"placement: {pure} of {len(groups)} groups have all four blocks on one XCD; block b -> XCC {xcc[:16]} ...")

# ---- timing ----
rows = []


def line(name, fn):
    full.reset(); h0.reset(); h1.reset()
    med, best = timed(fn)
    rows.append((name, med, best))
    print(f"{name:78s} {med:7.3f} ms  (best {best:.3f})", flush=True)
    bad = [int(c.tickets[4 * 64]) for c in (full, h0, h1)]
    if any(bad):
        sys.exit(f"a chained launch gave up waiting for a ticket (give-up words {bad}): hand-off bug, not a timing")


line("separate launches, one chain B = 64, 192x192 tiles everywhere", lambda: full.forward_separate(cur))
line("separate launches, two B = 32 lanes (two streams), 192x192 tiles", lambda: interleaved(h0, h1, sep_step, sep_step))
check(lib.rmcl_tune_set(0, -1))
line("separate launches, one chain B = 64, default routing (fc1 on 192x384 tiles)", lambda: full.forward_separate(cur))
check(lib.rmcl_tune_set(10, 2))
line("separate launches, two B = 32 lanes, default routing", lambda: interleaved(h0, h1, sep_step, sep_step))
check(lib.rmcl_tune_set(10, 1))
check(lib.rmcl_tune_set(0, 60))
for flags, what in ((1, "release per stage"), (0, "same-XCD hand-off")):
    line(f"CHAINED (attention + one launch per layer), one chain B = 64, {what}", lambda: full.forward_chained(cur, flags))
    line(f"CHAINED, two B = 32 lanes (two streams), {what}", lambda: interleaved(h0, h1, chain_step(flags), chain_step(flags)))
check(lib.rmcl_tune_set(0, -1))
base2 = min(r[1] for r in rows if "separate" in r[0] and "two" in r[0])
best_chain = min(r[1] for r in rows if "CHAINED" in r[0])
print(f"\ngate: best chained {best_chain:.3f} ms vs best two-lane separate-launch forward {base2:.3f} ms -> {100 * (1 - best_chain / base2):+.1f} % "
      f"(needs >= +8 %)")

# ---- phase stamps of one workgroup (block 100: group 4 + 8 * 3, member 0), one chained launch in the middle of a forward ----
for flags, what in ((1, "release per stage"), (0, "same-XCD hand-off")):
    full.reset()
    full.qkv_gemm(cur)
    for l in range(3):
        full.attention(cur)
        full.chain_launch(cur, flags, 4, stamp_wg=100 if l == 2 else -1)
    torch.cuda.synchronize()
    st = full.stamps.cpu().view(4, 4).tolist()
    t0 = st[0][0]
    print(f"\nphase stamps, block 100, {what} (us from the block's start; 100 MHz wall clock):")
    for i, nm in enumerate(("proj  (K 768, 1 tile)", "fc1   (K 768, 4 tiles)", "fc2   (K 3072, 1 tile)", "qkv   (K 768, 3 tiles)")):
        e, w, b_, p_ = [(v - t0) / 100.0 for v in st[i]]
        print(f"  {nm:24s} entered {e:7.2f}  ticket seen {w:7.2f} (+{w - e:5.2f})  body done {b_:7.2f} (+{b_ - w:6.2f})  published {p_:7.2f} (+{p_ - b_:5.2f})")
