"""CPU, world_size 2 over gloo: the multi-GPU host logic of the RMCL step (rmcl_amd.vilt.modules.dist_utils):
rank-major key all-gather feeding the enqueue, the reference's skip rule, queue pointer bookkeeping, and
gradient averaging over the flat arena.  On MI355X the same code runs with backend "nccl" (RCCL over xGMI)."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rmcl_pkg  # noqa: F401
from oracle import rmcl_oracle as O


def _worker(rank, world, initfile, out_dir):
    import rmcl_pkg  # noqa: F401
    from rmcl_amd.vilt.modules import dist_utils
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        B, Pd, Kq = 4, 128, 64
        g = torch.Generator().manual_seed(100 + rank)
        k = torch.nn.functional.normalize(torch.randn(B, Pd, generator=g), dim=1)
        gather = dist_utils.KeyGather(k)                       # async; other work would overlap here
        filler = torch.randn(64, 64) @ torch.randn(64, 64)     # stands in for the PGD loop
        keys_all = gather.wait()
        assert keys_all.shape == (world * B, Pd) and filler is not None
        assert torch.equal(keys_all[rank * B:(rank + 1) * B], k)            # rank-major like torch.cat(tensors_gather)
        # enqueue bookkeeping: every rank applies the same block -> queues stay bit-identical
        queue = torch.zeros(Pd, Kq)
        ptr = 0
        for step in range(3):
            do, new_ptr = dist_utils.queue_advance(ptr, keys_all.shape[0], Kq, per_step_bs=world * B)
            assert do
            ptr = O.enqueue(queue, ptr, keys_all, world * B)                # oracle = reference semantics
            assert ptr == new_ptr
        assert ptr == (3 * world * B) % Kq
        do, same = dist_utils.queue_advance(ptr, keys_all.shape[0], Kq, per_step_bs=world * B + 1)   # skip rule (:242-243)
        assert not do and same == ptr
        # flat-arena gradient averaging in buckets
        grads = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        dist_utils.allreduce_mean_(grads, bucket_elems=256)
        expect = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        assert torch.allclose(grads, expect)
        # bucketed (per-layer) gradient sync, the order the overlapped GPU path issues it in
        layer0, stride, layers, total = 37, 50, 4, 300
        buckets = dist_utils.grad_buckets(layer0, stride, layers, total)
        g2 = torch.arange(total, dtype=torch.float32) * (rank + 1)
        sync = dist_utils.GradSync(g2, buckets, None, None)
        sync.wait()
        assert torch.allclose(g2, torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
        torch.save({"queue": queue, "keys": keys_all}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _worker_step_sync(rank, world, initfile, out_dir):
    """The shipped gradient path at world 2: every deferred-backward closure accumulates into ONE arena with the
    1/world prescale on its loss gradient; the arena is SUM-reduced exactly once per step, after the last closure."""
    import rmcl_pkg  # noqa: F401
    from rmcl_amd.vilt.modules import dist_utils
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        n = 300
        base = torch.arange(n, dtype=torch.float32)
        contrib = {"itm": base * (rank + 1), "moco": (base + 7.0) * (2 * rank + 1), "third": (base * base) * (rank + 3)}
        mean_of = lambda name: sum((base * (r + 1) if name == "itm" else (base + 7.0) * (2 * r + 1) if name == "moco"
                                    else (base * base) * (r + 3)) for r in range(world)) / world

        def run(names, accumulate_micro=1, overlap=False):
            arena = torch.zeros(n)
            sync = dist_utils.StepGradSync()
            for micro in range(accumulate_micro):
                sync.begin_step()
                last_micro = micro == accumulate_micro - 1
                scales = [sync.register() for _ in names]                       # forward: one deferred loss per objective
                assert all(abs(sc - 1.0 / world) < 1e-12 for sc in scales)
                for name, sc in zip(names, scales):                             # backward: closures run one after another
                    arena += contrib[name] * sc                                 # the HIP backward accumulates (dW +=)
                    factory = None
                    if overlap:
                        buckets = dist_utils.grad_buckets(37, 50, 4, n)
                        factory = lambda: dist_utils.GradSync(arena, buckets, None, None, prescaled=True)
                    sync.closure_done(arena, enabled=last_micro, overlap=factory)
            sync.wait()
            return arena

        # three closures in one step (BASELINE configs[4]: text, image and both views) -> the mean, not W^2 R_a + W R_b + R_c
        got = run(["itm", "moco", "third"])
        assert torch.allclose(got, mean_of("itm") + mean_of("moco") + mean_of("third"), rtol=1e-6), "3 closures"
        # two closures (itm + moco)
        assert torch.allclose(run(["itm", "moco"]), mean_of("itm") + mean_of("moco"), rtol=1e-6)
        # one closure: the overlapped per-layer path (prescaled=True)
        assert torch.allclose(run(["moco"], overlap=True), mean_of("moco"), rtol=1e-6)
        # bf16-compressed buckets (config grad_allreduce_dtype="bf16"): same mean within bf16 rounding of the bucket values
        arena = contrib["moco"] * (1.0 / world)
        sync = dist_utils.GradSync(arena, dist_utils.grad_buckets(37, 50, 4, n), None, None, prescaled=True, compress="bf16")
        sync.wait()
        assert torch.allclose(arena, mean_of("moco"), rtol=2e-2), "bf16 buckets"
        assert not torch.equal(arena, mean_of("moco"))                           # (it really went through bf16)
        # gradient accumulation over 3 micro-steps, reduced only on the last: mean over ranks of the SUM over micro-steps
        assert torch.allclose(run(["moco"], accumulate_micro=3), 3 * mean_of("moco"), rtol=1e-6)
        assert torch.allclose(run(["itm", "moco"], accumulate_micro=2), 2 * (mean_of("itm") + mean_of("moco")), rtol=1e-6)
    finally:
        dist.destroy_process_group()


def _worker_direct(rank, world, initfile, outdir):
    import rmcl_pkg  # noqa: F401
    from rmcl_amd.vilt.modules import dist_utils
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        for n in (1000, 256 * world, 7):                                        # padded, exactly sliceable, shorter than a slice
            parts = [torch.randn(n, generator=g) for _ in range(world)]         # every rank draws all contributions
            want = parts[0].clone()
            for r in range(1, world):
                want += parts[r]                                                 # rank-order fp32 sum = what the owner computes
            mine = parts[rank].clone()
            ref = parts[rank].clone()
            dist.all_reduce(ref)
            red = dist_utils.DirectReduce(mine)
            red.finish()
            assert torch.equal(mine, want), f"direct reduce n={n}"
            assert torch.allclose(mine, ref, rtol=1e-5, atol=1e-6)               # and the ring all-reduce up to summation order
            # bf16 wire: pieces rounded once, sum rounded once
            mine16 = parts[rank].clone()
            dist_utils.DirectReduce(mine16, torch.bfloat16).finish()
            want16 = parts[0].bfloat16().float()
            for r in range(1, world):
                want16 += parts[r].bfloat16().float()
            assert torch.equal(mine16, want16.bfloat16().float()), f"bf16 wire n={n}"
        # through the step reducer: 2 closures, blocking pass, direct form; and the overlapped per-layer buckets
        n = 300
        base = torch.arange(n, dtype=torch.float32)
        sync = dist_utils.StepGradSync(algo="direct")
        sync.begin_step()
        sc = [sync.register(), sync.register()]
        arena = torch.zeros(n)
        for i, s_ in enumerate(sc):
            arena += base * (rank + 1 + i) * s_
            sync.closure_done(arena, enabled=True)
        sync.wait()
        want = sum(base * (r + 1) + base * (r + 2) for r in range(world)) / world
        assert torch.allclose(arena, want, rtol=1e-6)
        arena = base * (rank + 1) / world
        h = dist_utils.GradSync(arena, dist_utils.grad_buckets(37, 50, 4, n), None, None, prescaled=True, algo="direct")
        h.wait()
        assert torch.allclose(arena, sum(base * (r + 1) for r in range(world)) / world, rtol=1e-6)
        # every rank ends with the SAME bits (the owner computes each slice once)
        torch.save(arena, os.path.join(outdir, f"d{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_reduce_scatter_all_gather_matches_allreduce(world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_direct, args=(world, os.path.join(d, "init"), d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"d{r}.pt")) for r in range(world)]
        assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_two_rank_step_grad_sync_reduces_once():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_step_sync, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)


def test_two_rank_gather_enqueue_allreduce():
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, initfile, d), nprocs=2, join=True)
        a, b = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
        assert torch.equal(a["queue"], b["queue"]) and torch.equal(a["keys"], b["keys"])


def test_eight_rank_gather_enqueue_and_gradient_paths():
    """World size 8 - the size the product targets (BASELINE configs[3]) - on the CPU over gloo: rank-major gather of 8 x B keys, three
    enqueues of 8 x B-key blocks leaving all eight queues bit-identical, the skip rule, bucketed mean, the per-layer buckets; then the
    step reducer (1, 2, 3 closures, accumulation, bf16 buckets) and the one-hop reduce-scatter / all-gather with its padded slices."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(8, os.path.join(d, "init"), d), nprocs=8, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(8)]
        assert all(torch.equal(outs[0]["queue"], o["queue"]) and torch.equal(outs[0]["keys"], o["keys"]) for o in outs[1:])
        assert outs[0]["keys"].shape[0] == 8 * 4
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_step_sync, args=(8, os.path.join(d, "init"), d), nprocs=8, join=True)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_direct, args=(8, os.path.join(d, "init"), d), nprocs=8, join=True)
        outs = [torch.load(os.path.join(d, f"d{r}.pt")) for r in range(8)]
        assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_queue_advance_rejects_wraparound():
    from rmcl_amd.vilt.modules import dist_utils
    with pytest.raises(RuntimeError):
        dist_utils.queue_advance(60, 8, 64, per_step_bs=8)
    assert dist_utils.queue_advance(56, 8, 64, per_step_bs=8) == (True, 0)


def test_grad_buckets_tile_the_arena_in_backward_order():
    from rmcl_amd.vilt.modules import dist_utils
    layer0, stride, layers, total = 1000, 700, 12, 1000 + 12 * 700 + 321
    b = dist_utils.grad_buckets(layer0, stride, layers, total)
    assert [x[0] for x in b] == list(range(11, -1, -1)) + [-1]                 # last layer first, embeddings last
    assert b[0] == (11, layer0 + 11 * stride, total)                           # + final norm / pooler / heads behind the layers
    covered = sorted((s, e) for _, s, e in b)
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))


def test_single_process_is_a_no_op():
    from rmcl_amd.vilt.modules import dist_utils
    k = torch.randn(3, 128)
    assert dist_utils.KeyGather(k).wait() is k
    assert dist_utils.allreduce_mean_(torch.ones(4)) == []
