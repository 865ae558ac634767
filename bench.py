#!/usr/bin/env python3
"""bench.py - RMCL training step throughput on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete RMCL training step on a synthetic batch (BASELINE.json configs[2]/[3]):
momentum update, key encoder, clean query, PGD K=3 image attack (3 x forward + data-gradient backward +
L-inf-normalised step), attacked-view forward + full backward, InfoNCE against the 65 536-entry queue,
key all-gather + enqueue, gradient all-reduce (N>1) and the fused AdamW step.  bs=64/GPU, 384x384 + 40
tokens, bf16 operands / fp32 accumulate, inputs resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  roofline     - the dominant kernel class (the encoder MLP forward GEMMs, fc1 +bias+GELU+stash and fc2
                 +bias+residual, exactly as the step launches them at M = bs*185): right after the timed
                 steps the two launches are replayed back to back on the launch stream between HIP events
                 (per-launch event brackets inside the step add ~20 us of marker/dispatch latency each, so
                 they over-state the duration rocprofv3 reports); algorithmic FLOPs / average duration vs
                 the dense bf16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s);
  step_mfma_frac - whole-step EXECUTED FLOPs / step time / peak: (4+2K) forward-equivalents per pair with dropout off (the
                 clean query forward IS PGD step 0's forward and runs once; SURVEY 8d counts (5+2K)F), each F less the part of
                 the last block the cls-only tail does not execute (5.9 % of F; F_TAIL_SKIPPED) - work that is skipped is not
                 credited; `value` (pairs/s) is what the skipping buys;
  cpu_baseline - the CPU oracle (port of the reference algorithm, oracle/rmcl_oracle.py) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N=1 only).

--config rmcl_pgd (default) is BASELINE configs[2] (the metric's configuration; configs[3] at N=8);
--config itm_clean is BASELINE configs[1] ("clean ITM + contrastive": ITM + word-patch alignment + CE on the clean
InfoNCE logits, 7F per pair, SURVEY 8d Config 2).

--gpus N > 1 without a torchrun environment: bench.py launches the N ranks itself (torch.distributed.run as a child
process, before anything touches the GPU) and exits with the child's code.  A WORLD_SIZE that disagrees with --gpus is an
error.  The JSON carries the process group's actual size (`world_size`).

Developer switches (never the measured configuration): RMCL_BENCH_SHARE_GPU=1 + RMCL_BENCH_BACKEND=gloo rehearse the N > 1 code
path with all ranks on cuda:0 (the JSON is marked `rehearsal` and reports `ranks_bit_identical`); RMCL_BENCH_FORCE_DIST=1 runs the
1-rank RCCL group; RMCL_LIB=<path> loads another build of the library (A/B runs, tools/ab_bench.sh).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15          # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
F_PER_PAIR = 33.386e9       # one encoder forward, SURVEY.md 8(d)
# cls-only tail (include/rmcl.h RMCL_MODE_CLS_TAIL): in a contrastive pass the last block's proj / fc1 / fc2 (9 of its 12 D x D
# linear units) run on the cls row only - FLOPs NOT executed per forward-equivalent and pair (N = 185 tokens, D = 768)
F_TAIL_SKIPPED = 0.75 * (2 * 185 * 12 * 768 * 768) * 184 / 185


def synthetic_batch(cfg, B, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    S, Lt = cfg["image_size"], cfg["max_text_len"]
    img = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    ids = torch.randint(1000, cfg["vocab_size"], (B, Lt), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    return {"image": [img.to(device)], "text": ["synthetic"] * B, "text_ids": ids.to(device),
            "text_masks": torch.ones(B, Lt, dtype=torch.int64, device=device),
            "text_labels": torch.full((B, Lt), -100, dtype=torch.int64, device=device)}


def mlp_gemm_replay(L, M, dtype, device, reps=40):
    """The step's two dominant GEMM launches (same C-ABI entry, shapes, epilogues and dtypes as encoder.cpp issues
    them), `reps` x (fc1, fc2) back to back between two events on the launch stream.  Returns (ms, launches, flops)."""
    D, H = 768, 3072
    lp = torch.bfloat16 if dtype == "bf16" else torch.float32
    dt = L.BF16 if dtype == "bf16" else L.F32
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(M, D, generator=g).to(device).to(lp)
    w1 = (torch.randn(H, D, generator=g) * 0.02).to(device).to(lp)
    w2 = (torch.randn(D, H, generator=g) * 0.02).to(device).to(lp)
    b1, b2 = torch.zeros(H, device=device), torch.zeros(D, device=device)
    h, u = torch.empty(M, H, dtype=lp, device=device), torch.empty(M, H, dtype=lp, device=device)
    res, y = torch.randn(M, D, generator=g).to(device), torch.empty(M, D, device=device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    exact = 0 if dtype == "bf16" else 1

    def pair():
        L.check(L.lib.rmcl_gemm(L.P(x), L.P(w1), L.P(h), L.P(u), L.P(b1), None, M, H, D, L.I64(D), L.I64(D), H, 0, L.F(1.0), 1 | 2 | 4, 1,
                                dt, dt, 1, 1, exact, st), "fc1")
        L.check(L.lib.rmcl_gemm(L.P(h), L.P(w2), L.P(y), None, L.P(b2), L.P(res), M, D, H, L.I64(H), L.I64(H), D, D, L.F(1.0), 1 | 8, 1,
                                dt, L.F32, 1, 1, exact, st), "fc2")
    for _ in range(3):
        pair()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        pair()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), 2 * reps, reps * 2 * (2.0 * M * D * H)


def cpu_baseline(K_adv, config="rmcl_pgd"):
    """Oracle (kind "port") on the host cores: one RMCL step (image view, PGD K) + backward; for --config itm_clean
    the ITM + word-patch-alignment + clean-InfoNCE step + backward."""
    from oracle import rmcl_oracle as O
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                           # cgroup v2 CPU quota (the GPU box gives a share of the host)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = min(threads, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    threads = min(threads, 32)
    torch.set_num_threads(threads)
    Bc, nsteps = 8, 3                              # B = 8, 1 warm-up + 3 timed steps
    clean = config == "itm_clean"
    ocfg = O.default_config(per_gpu_batchsize=Bc, adv_steps_img=K_adv, image_view=not clean, clean_view=clean)
    labels = (torch.arange(Bc) % 2).float()
    p = O.init_params(ocfg, 1)
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    queue = O.init_queue(ocfg, 0)
    batch = O.synthetic_batch(ocfg, Bc, 2)
    ptr, dt = 0, 0.0
    for i in range(nsteps + 1):                       # 1 warm-up + nsteps timed
        t0 = time.time()
        ret = O.compute_moco_contrastive(p, ocfg, batch, queue, ptr, training=True)
        loss = ret["moco_loss"]
        if clean:
            ri = O.compute_itm_wpa(p, ocfg, batch, labels)
            loss = loss + ri["itm_loss"] + ri["itm_wpa_loss"]
        loss.backward()
        ptr = ret["ptr"]
        if i > 0:
            dt += time.time() - t0
    return {"value": Bc * nsteps / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{nsteps} steps of the same workload at bs={Bc} (12 layers, queue 65536, "
                      + ("ITM+WPA + clean InfoNCE" if clean else f"PGD K={K_adv}") + f", fwd+bwd, no optimizer), fp32 CPU oracle, {dt:.1f} s"}


def self_launch(args):
    """--gpus N > 1 outside torchrun: start the N ranks as a child torch.distributed.run (nothing has touched the GPU in
    this process) and leave with its exit code."""
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


ROOFLINE_KERNEL_SOURCES = ("gemm_st.hip", "gemm_sw.hip", "gemm_st_epi.h", "gemm_fast.hip", "gemm.h", "rmcl_common.h")


def roofline_kernel_build_id():
    """Identity of the code the roofline kernels are built from: sha256 over the sources of the two GEMM kernels and their routing
    (tools/pmc_traffic.py stores the same value with the PMC record it writes)."""
    import hashlib
    h = hashlib.sha256()
    for f in ROOFLINE_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "robust-multimodal-contrastive-learning_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_record(kernel_key):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC passes (profiles/roofline_traffic.json,
    written by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload,
    gfx950 correction applied: FETCH_SIZE x 2).  The record is used only when it was measured on THIS build of the kernels
    (`kernel_build_id`); otherwise (None, reason)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
    except Exception:
        return None, "no PMC record (profiles/roofline_traffic.json)"
    r = rec.get(kernel_key)
    if r is None:
        return None, "no PMC record for this kernel"
    have, want = rec.get("kernel_build_id"), roofline_kernel_build_id()
    if have != want:
        return None, f"PMC record is stale: measured on kernel build {have}, this build is {want} (re-run tools/pmc_traffic.py)"
    return r, r.get("note")


def realistic_leg(model, cfg, batch, args, device, ViLTransformerSS, steps=15, warmup=5):
    """The reference's real recipe beside the headline (SURVEY 8d "a separate training-realistic throughput line"): `task_moco` is
    trained from `load_path=...vilt_200k_mlm_itm.ckpt` (TRAIN.md:21) with `drop_rate = 0.1` (config.py:57).  The headline model's state
    is written to a checkpoint file and a SECOND module is constructed from it through config["load_path"] with drop_rate 0.1 - i.e.
    through the code path a real run takes (what stays on after the load is reported) - then timed on the same synthetic batch:
    dropout live in every train-mode forward incl. the key encoder and the PGD passes, the clean query forward separate from PGD
    step 0 ((5 + 2K) F per pair instead of (4 + 2K) F: the two draw different masks in the reference)."""
    import tempfile
    import shutil
    tmp = tempfile.mkdtemp(prefix="rmcl_bench_")
    try:
        path = os.path.join(tmp, "state.ckpt")
        torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}}, path)
        cfg2 = dict(cfg, drop_rate=0.1, load_path=path)
        cfg2["loss_names"] = dict(cfg["loss_names"])
        m2 = ViLTransformerSS(cfg2, device=device, compute_dtype=args.dtype)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    m2.train()
    (opt2,), (sched2,) = m2.configure_optimizers()

    def step(i):
        loss = m2.training_step(batch, i)
        loss.backward()
        opt2.step()
        sched2["scheduler"].step()
        opt2.zero_grad()
        return loss

    def run():
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(warmup + i)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    dt, loss = run()
    # opt-in variant: PGD step 0's forward also serves as the clean query forward under dropout (config["share_clean_forward"]; it changes
    # only the LOGGED clean prediction, objectives.py): (4 + 2K) F per pair like the dropout-free step
    m2.hparams.config["share_clean_forward"] = True
    dt_shared, _ = run()
    m2.hparams.config["share_clean_forward"] = False
    eng = m2.engine
    B = batch["text_ids"].shape[0]
    pb = eng.bufs(B)
    return {"drop_rate": 0.1, "load_path": "checkpoint of the headline model's state, loaded through config['load_path']",
            "ln_fold": m2.load_report.get("ln_fold"), "half_batch_lanes": "on" if getattr(pb, "_lanes", None) is not None else "off",
            "cls_only_tail": "on" if any(pb.tail.values()) else "off",
            "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3), "pairs_per_s": round(B * steps / dt, 2),
            "executed_F_per_pair": 5 + 2 * cfg["adv_steps_img"], "final_loss": round(float(loss.detach()), 4),
            "opt_in_share_clean_forward": {"ms_per_step": round(1e3 * dt_shared / steps, 3), "pairs_per_s": round(B * steps / dt_shared, 2),
                                           "executed_F_per_pair": 4 + 2 * cfg["adv_steps_img"],
                                           "note": "config['share_clean_forward'] = True: the clean query's logged prediction comes from PGD step 0's forward "
                                                   "(same dropout mask); loss / gradients / perturbation unchanged; default off = the reference's two draws"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--adv-steps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--drop-rate", type=float, default=0.0,
                    help="0 = the parity configuration (headline); 0.1 = the reference's training default (config.py:57)")
    ap.add_argument("--config", default="rmcl_pgd", choices=["rmcl_pgd", "itm_clean", "full_rmcl", "barlowtwins"],
                    help="rmcl_pgd = BASELINE configs[2]/[3] (the metric); itm_clean = BASELINE configs[1]; full_rmcl = configs[4] "
                         "(PGD K=5 + greedy text attack with synthetic candidates + the three views); barlowtwins = the Barlow-Twins "
                         "variant of the step (8192-wide head, cross-correlation loss, image view)")
    ap.add_argument("--grad-sync", default="ring", choices=["ring", "direct"],
                    help="N > 1 gradient reduction: ring = RCCL all-reduce per layer bucket; direct = one-hop reduce-scatter "
                         "(all-to-all + owner sum) + all-gather over the xGMI mesh (dist_utils.DirectReduce)")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="transport type of the gradient buckets")
    ap.add_argument("--grad-overlap", default="on", choices=["on", "off"],
                    help="N > 1: on = per-layer buckets reduced on the communication stream while the backward runs; off = one blocking "
                         "pass over the arena after the backward (the comparison run)")
    ap.add_argument("--no-feed-bench", action="store_true", help="skip the input-pipeline feed-rate measurement (tools/feed_bench.py, N = 1 only)")
    ap.add_argument("--rehearse-short-batch", action="store_true",
                    help="rehearsal runs only: one extra, untimed step on a HALF-size last batch after the timed steps - the gathered keys then "
                         "number world * B / 2 != per_step_bs and the enqueue must be skipped on every rank (objectives.py:242-243)")
    ap.add_argument("--no-realistic", action="store_true",
                    help="skip the second, short leg of the default run: the same step in the reference's real recipe (drop_rate 0.1 after a "
                         "load_path-style checkpoint load, TRAIN.md:21 / config.py:57) -> `training_realistic` in the JSON line")
    ap.add_argument("--padded-images", action="store_true",
                    help="run the general visual_embed path (pixel-mask patch selection + its device-to-host count read per step) on the "
                         "same full-size synthetic batch instead of the dense fast path")
    args = ap.parse_args()

    if args.grad_overlap == "off":
        os.environ["RMCL_NO_GRAD_OVERLAP"] = "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    feed = None
    if world == 1 and not args.no_feed_bench and args.config == "rmcl_pgd":
        # row f3: what the host pipeline can deliver, measured BEFORE this process touches the GPU (the child and its DataLoader
        # workers never do): reference default of 4 workers... and every core of this box's share
        feed = {}
        cores = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = min(cores, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
        for w in sorted({4, max(4, min(cores, 32))}):
            try:
                # decode_uint8: workers only decode (MinMaxResize + normalisation on the device, round 4); pixelbert_uint8: workers also resize
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "feed_bench.py"), "--json", "--workers", str(w), "--images", "128",
                                    "--seconds", "4", "--paths", "decode_uint8,pixelbert_uint8"] + (["--split"] if w == 4 else []),
                                   capture_output=True, text=True, timeout=120)
                feed[f"workers_{w}"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            except Exception as e:                              # the step measurement does not depend on it
                feed[f"workers_{w}"] = {"error": repr(e)[:200]}
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or without a torchrun environment, then bench.py starts the ranks itself)")
    # rehearsal of the N > 1 code path on a ONE-GPU box (developer use): RMCL_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # RMCL_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); never the measured configuration
    share_gpu = os.environ.get("RMCL_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("RMCL_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    use_dist = world > 1 or os.environ.get("RMCL_BENCH_FORCE_DIST", "0") == "1"   # (1-rank rehearsal of the N > 1 code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL's channel workgroups share the CUs with one-workgroup-per-CU GEMMs that leave 8 CUs free (rmcl_tune_set key 1);
        # 8 channels move the 447 MB of gradients well inside the ~12 ms backward window they overlap with
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import rmcl_pkg  # noqa: F401
    from rmcl_amd import _lib as L
    from rmcl_amd.vilt.config import task_moco, task_barlowtwins
    from rmcl_amd.vilt.modules import ViLTransformerSS

    if world > 1:
        L.check(L.lib.rmcl_tune_set(1, 8), "tune_set")     # leave 8 CUs to RCCL's channels (include/rmcl.h)
    for kv in filter(None, os.environ.get("RMCL_BENCH_TUNE", "").split(",")):      # developer switch: "key:value,..." for rmcl_tune_set
        k_, v_ = kv.split(":")
        L.check(L.lib.rmcl_tune_set(int(k_), int(v_)), "tune_set")
    B, K = args.batch, args.adv_steps
    clean = args.config == "itm_clean"
    full = args.config == "full_rmcl"
    if full:
        K = 5
    cfg = task_moco(per_gpu_batchsize=B, num_gpus=world, num_nodes=1, adv_steps_img=K, drop_rate=args.drop_rate, image_view=not clean,
                    text_view=full, clean_view=clean, max_steps=100000, max_loops=10, n_candidates=5,
                    dense_images=not args.padded_images)   # synthetic full-size 384x384 images: skip the per-batch padded-image check
    if clean:
        cfg["loss_names"]["itm"] = 1
    barlow = args.config == "barlowtwins"
    if barlow:
        cfg = task_barlowtwins(per_gpu_batchsize=B, num_gpus=world, num_nodes=1, adv_steps_img=K, drop_rate=args.drop_rate, image_view=True,
                               text_view=False, max_steps=100000, dense_images=not args.padded_images)
    cfg["grad_allreduce_algo"] = args.grad_sync
    cfg["grad_allreduce_dtype"] = "bf16" if args.grad_dtype == "bf16" else None
    torch.manual_seed(0)
    model = ViLTransformerSS(cfg, device=device, compute_dtype=args.dtype)
    model.train()
    (opt,), (sched,) = model.configure_optimizers()
    batch = synthetic_batch(cfg, B, 1234 + rank, device)
    if clean:
        batch["false_image_0"] = [torch.roll(batch["image"][0], shifts=1, dims=0)]    # SURVEY 8d Config 2
    if use_dist:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    def step(i):
        loss = model.training_step(batch, i)
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        opt.zero_grad()
        return loss

    for i in range(args.warmup):
        step(i)
    from rmcl_amd.vilt.modules.dist_utils import StepTimers
    StepTimers.enabled = use_dist                       # event brackets around the two waits on communication (N > 1 only)
    StepTimers.reset()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    comm = {k: v / args.steps for k, v in StepTimers.totals_ms().items()} if use_dist else {}
    StepTimers.enabled = False
    kern_ms, kern_n, kern_fl = mlp_gemm_replay(L, B * 185, args.dtype, device) if rank == 0 else (0.0, 0, 0.0)
    ingest = None
    if feed is not None:
        # device side of the byte path: pinned uint8 [B,384,384,3] -> H2D -> ONE kernel (normalise + pad + patch rows), per batch
        from rmcl_amd.vilt.datasets import Uint8Batch
        u8 = Uint8Batch(torch.randint(0, 256, (B, 384, 384, 3), dtype=torch.uint8).pin_memory(), torch.full((B, 2), 384, dtype=torch.int32))
        eng = model.engine
        for _ in range(2):
            eng.bind_batch(batch["text_ids"], batch["text_masks"], u8, tag="feed")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            eng.bind_batch(batch["text_ids"], batch["text_masks"], u8, tag="feed")
        e1.record()
        torch.cuda.synchronize()
        ingest = {"pairs_per_s": round(10 * B / (e0.elapsed_time(e1) * 1e-3), 1), "bytes_over_pcie_per_batch": int(u8.data.numel()),
                  "what": "pinned uint8 batch -> device -> rmcl_image_u8_to_patches (normalise + zero-pad + patch rows), stream time per batch"}
        # the decode-only path: decoded 640 x 480 / 480 x 640 bytes -> device -> MinMaxResize (PIL's integer bicubic, two passes) -> the above
        from rmcl_amd.vilt.datasets import RawUint8Batch
        rsz = torch.tensor([(480, 640) if i % 3 else (640, 480) for i in range(B)], dtype=torch.int32)
        raw = RawUint8Batch(torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).pin_memory(), rsz, 384, 640)
        for _ in range(2):
            eng.bind_batch(batch["text_ids"], batch["text_masks"], raw, tag="feed_raw")
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            eng.bind_batch(batch["text_ids"], batch["text_masks"], raw, tag="feed_raw")
        e1.record()
        torch.cuda.synchronize()
        ingest["decode_only"] = {"pairs_per_s": round(10 * B / (e0.elapsed_time(e1) * 1e-3), 1), "bytes_over_pcie_per_batch": int(raw.data.numel()),
                                 "what": "pinned decoded bytes at original size -> device -> rmcl_image_resize_u8 (MinMaxResize, PIL's integer "
                                         "bicubic) -> rmcl_image_u8_to_patches, stream time per batch (incl. the host's table packing)"}
    per_rank = None
    if use_dist:
        mine = {"step_ms": 1e3 * elapsed / args.steps, "comm_exposed_ms": comm.get("comm_exposed", 0.0),
                "key_gather_wait_ms": comm.get("key_gather_wait", 0.0)}
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    short = None
    if args.rehearse_short_batch:
        half = synthetic_batch(cfg, max(2, B // 2), 4321 + rank, device)
        ptr0 = model.queue_ptr
        lossh = model.training_step(half, args.warmup + args.steps)
        lossh.backward()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        short = {"batch": max(2, B // 2), "queue_ptr_before": ptr0, "queue_ptr_after": model.queue_ptr, "enqueue_skipped": model.queue_ptr == ptr0,
                 "loss_finite": bool(torch.isfinite(lossh.detach()))}
    consistent, digest = None, None
    if use_dist and (share_gpu or backend != "nccl"):          # rehearsal: the ranks' parameters and queues must be bit-identical
        eng = model.engine
        mine = [float(eng.q32.double().sum()), float(eng.q32.double().abs().sum()), float(eng.queue.double().sum())]
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        consistent = all(v == allv[0] for v in allv)
        digest = mine

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        # algorithmic FLOPs per GPU per step: SURVEY 8(d) counts (5+2K)F; with dropout off the clean query forward
        # and PGD step 0's forward are the same computation and run once -> (4+2K)F are executed and credited
        tail = (5 * B if full else B) <= 1024 and os.environ.get("RMCL_NO_CLS_TAIL", "0") != "1"   # (candidate batch: 5 B rows)
        F_c = F_PER_PAIR - (F_TAIL_SKIPPED if tail else 0.0)           # executed FLOPs of one contrastive forward-equivalent
        step_flops = ((4 if args.drop_rate == 0 else 5) + 2 * K) * F_c * B
        workload = (f"RMCL step, PGD K={K} image attack + MoCo InfoNCE (queue 65536) + full backward + AdamW, "
                    f"ViLT-B/32, bs={B}/GPU, 384x384 img + 40 tok (BASELINE configs[2]; [3] when n_gpus=8)")
        metric = f"image-text pairs/sec, ViLT-B/32 RMCL step (PGD K={K})"
        if clean:                                        # key 1F + clean query fwd+bwd 3F + ITM fwd+bwd 3F (SURVEY 8d Config 2)
            step_flops = (4 * F_c + 3 * F_PER_PAIR) * B               # the ITM pass needs every token of the last block
            workload = (f"clean ITM + contrastive step: ITM + word-patch alignment (IPOT) + CE on the clean InfoNCE logits (queue 65536) "
                        f"+ full backward + AdamW, ViLT-B/32, bs={B}/GPU, 384x384 img + 40 tok (BASELINE configs[1])")
            metric = "image-text pairs/sec, ViLT-B/32 clean ITM+contrastive step"
        if full:
            # SURVEY 8d Config 5: 11F (key, clean, three attacked views fwd+bwd) + 2K F (PGD) + the text attack: per loop 2F
            # (saliency fwd + data backward) + 5F (five candidate sentences per sample) = 70F for 10 loops; K = 5
            step_flops = (11 + 2 * K + 10 * (2 + 5)) * F_c * B
            workload = (f"full RMCL step: PGD K={K} image attack + greedy text attack (10 loops x 5 synthetic candidates per sample) + text / "
                        f"image / both views + MoCo InfoNCE (queue 65536) + full backward + AdamW, ViLT-B/32, bs={B}/GPU (BASELINE configs[4])")
            metric = f"image-text pairs/sec, ViLT-B/32 full RMCL step (PGD K={K} + text attack)"
        if barlow:
            # clean forward 1F + K PGD steps 2F each + attacked forward/backward 3F; the head adds 3 x 2 x B x 8192^2-ish FLOPs per
            # pass (weight-streaming, HBM-bound) and the 8192 x 8192 x B correlation - not counted in F
            step_flops = (4 + 2 * K) * F_c * B
            workload = (f"Barlow-Twins variant of the RMCL step: clean projection + PGD K={K} on the cross-correlation loss + attacked view + "
                        f"full backward + AdamW (ViLT-B/32 + 768-8192-8192-8192 BatchNorm head), bs={B}/GPU (objectives.py:449-602)")
            metric = f"image-text pairs/sec, ViLT-B/32 Barlow-Twins step (PGD K={K})"
        kern_tf = kern_fl / (kern_ms * 1e-3) / 1e12 if kern_n else 0.0
        traffic, traffic_note = traffic_record("mlp_fwd_pair") if args.dtype == "bf16" and B == 64 else (None, "PMC record exists for bf16, bs=64 only")
        out = {
            "metric": metric, "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "world_size": dist.get_world_size() if use_dist else 1,
            **({"rehearsal": f"{backend} backend, ranks share one GPU: NOT a scaling measurement", "ranks_bit_identical": consistent,
                "param_digest": digest}
               if (share_gpu or backend != "nccl") else {}),
            "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}", "drop_rate": args.drop_rate,
                       "grad_sync": f"{args.grad_sync}/{args.grad_dtype}", "final_loss": round(final_loss, 4),
                       "images": ("dense fast path (every synthetic image is full-size: no patch selection, no per-step count read)"
                                  if not args.padded_images else "general visual_embed path (pixel-mask selection + count read per step)")},
            "roofline": {"bound": "mfma", "achieved": round(kern_tf, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(kern_tf * 1e12 / PEAK_BF16, 4),
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_note": traffic_note, "kernel_build_id": roofline_kernel_build_id(),
                         "kernel": "encoder MLP forward GEMMs (fc1 768->3072 +bias+GELU, fc2 3072->768 +bias+residual), "
                                   f"M={B * 185}; {kern_n} replayed launches, avg {kern_ms / max(kern_n, 1):.4f} ms"},
            "step_mfma_frac": round(step_flops / (elapsed / args.steps) / PEAK_BF16, 4),
        }
        if short is not None:
            out["short_last_batch"] = short
        out["queue_ptr"] = model.queue_ptr if hasattr(model, "proj_queue_ptr") else None
        out["lanes"] = "on" if getattr(model.engine.bufs(B), "_lanes", None) is not None else "off"
        if per_rank is not None:
            # what an N > 1 run needs to explain itself: per-rank step time, the stream time the step stood still waiting for
            # communication, and what was sent (max over ranks of the waits: the slowest rank sets the step)
            sm = [r["step_ms"] for r in per_rank]
            out["multi_gpu"] = {
                "ranks": dist.get_world_size(), "backend": backend, "step_ms_per_rank": {"min": round(min(sm), 3), "max": round(max(sm), 3)},
                "comm_exposed_ms": round(max(r["comm_exposed_ms"] for r in per_rank), 3),
                "key_gather_wait_ms": round(max(r["key_gather_wait_ms"] for r in per_rank), 3),
                "grad_overlap": args.grad_overlap, "grad_sync": StepTimers.info.get("grad_sync"),
                "key_gather_bytes_per_rank": B * 128 * 4, "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                "cus_left_to_rccl": 8 if world > 1 else 0}
        if feed is not None:
            out["feed"] = {"host": feed, "device_ingest": ingest,
                           "note": "row f3: host pairs/s of the arrow -> batch pipeline (byte path) beside the step's `value`; the step itself is "
                                   "measured on a batch already resident in HBM"}
        out["config"]["ln_fold"] = "on (shift-robust form)" if model.engine.fold else "off"
        if world == 1 and args.config == "rmcl_pgd" and args.drop_rate == 0 and not args.no_realistic:
            out["training_realistic"] = realistic_leg(model, cfg, batch, args, device, ViLTransformerSS)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(K, args.config)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()                                     # rank 0 is still replaying / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
