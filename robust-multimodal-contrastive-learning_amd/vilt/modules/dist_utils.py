"""Multi-GPU host logic of the RMCL step: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).  Device-agnostic on purpose so the
world_size-2 gloo tests exercise exactly this code.

Replaces: ``_concat_all_gather`` / ``_dequeue_and_enqueue`` (vilt/modules/objectives.py:226-248)
and the implicit DDP gradient all-reduce of run.py:96.  The reference's detectron2-style pickle
gathers (vilt/modules/dist_utils.py) serve IR/TR recall only and are out of scope."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class StepTimers:
    """Event brackets around the step's two waits on communication (bench.py --gpus N reports them per step so that a scaling
    run explains itself): `comm_exposed` = how long the main stream stood still in the optimizer's wait for the gradient
    all-reduces, `key_gather_wait` = the same for the key all-gather at the enqueue.  Off unless bench.py switches it on; CUDA
    streams only (two event records per wait, no host synchronisation)."""
    enabled = False
    pairs: dict = {}
    info: dict = {}

    @classmethod
    def reset(cls):
        cls.pairs = {}

    class _Bracket:
        def __init__(self, name):
            self.name = name
            self.e0 = None

        def __enter__(self):
            if StepTimers.enabled and torch.cuda.is_available():
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                StepTimers.pairs.setdefault(self.name, []).append((self.e0, e1))
            return False

    @classmethod
    def bracket(cls, name):
        return cls._Bracket(name)

    @classmethod
    def totals_ms(cls) -> dict:
        """Sum of the bracketed stream time per name (call after a device synchronise)."""
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in cls.pairs.items()}


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class KeyGather:
    """Asynchronous all-gather of the momentum keys [B,128] fp32 (32 KiB per rank), rank-major
    like ``torch.cat(tensors_gather, 0)`` (objectives.py:231-234).  Launched right after the key
    head; RCCL runs it on its own stream while the clean forward, the PGD loop and the attacked
    forward proceed; ``wait()`` is called just before the enqueue (objectives.py:394-395)."""

    def __init__(self, k: torch.Tensor):
        self.k = k
        self.out: Optional[torch.Tensor] = None
        self.work = None
        ws = world_size()
        if ws == 1:
            self.out = k
            return
        self.out = torch.empty(ws * k.shape[0], k.shape[1], dtype=k.dtype, device=k.device)
        self.work = dist.all_gather_into_tensor(self.out, k.contiguous(), async_op=True)

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            with StepTimers.bracket("key_gather_wait"):
                self.work.wait()
            self.work = None
        return self.out


def queue_advance(ptr: int, n_keys: int, num_negative: int, per_step_bs: int) -> Tuple[bool, int]:
    """Bookkeeping of ``_dequeue_and_enqueue`` (objectives.py:241-248): skip when the gathered batch
    differs from per_step_bs; no wrap-around handling (needs num_negative % n_keys == 0)."""
    if n_keys != per_step_bs:
        return False, ptr
    if ptr + n_keys > num_negative:
        raise RuntimeError(f"queue block [{ptr}, {ptr + n_keys}) runs past num_negative={num_negative} "
                           "(the reference requires num_negative % batch == 0)")
    return True, (ptr + n_keys) % num_negative


class DirectReduce:
    """SUM over ranks of one flat fp32 chunk as a ONE-HOP reduce-scatter + all-gather (replaces
    the ring all-reduce inside DDP, run.py:96).  MI355X xGMI is a full mesh of point-to-point links (7 per GPU): a ring
    moves 2(W-1)/W of the bytes through W-1 hops, this form sends slice j straight to its owner j over the direct link
    (``all_to_all_single``: W-1 concurrent sends per rank), the owner adds the W pieces (``rmcl_shard_sum``: fp32
    accumulation in rank order, so the result does not depend on arrival order) and the finished slices return with one
    ``all_gather_into_tensor``.  Same bytes per rank as the ring, one hop each way, every link busy at once.

    ``wire``: torch.bfloat16 sends both legs in bf16 (half the bytes; the pieces are rounded once before the sum and the
    sum once after it); default fp32 is exact up to the summation order.  Slices are padded to a multiple of 256 elements.
    On a CUDA chunk everything is enqueued on the CURRENT stream (the collectives' completion is a stream dependency,
    not a host wait); ``finish()`` waits for the all-gather and writes the chunk back."""

    def __init__(self, chunk: torch.Tensor, wire: Optional[torch.dtype] = None):
        ws = world_size()
        n = chunk.numel()
        s = -(-n // ws)
        s = (s + 255) // 256 * 256
        wire = wire or chunk.dtype
        self.chunk, self.n = chunk, n
        if ws * s == n and wire == chunk.dtype:
            send = chunk
        else:
            send = torch.zeros(ws * s, dtype=wire, device=chunk.device)
            send[:n].copy_(chunk)
        recv = torch.empty(ws * s, dtype=wire, device=chunk.device)
        dist.all_to_all_single(recv, send)
        mine = self._owner_sum(recv, ws, s)
        self.gathered = chunk if send is chunk else torch.empty(ws * s, dtype=wire, device=chunk.device)
        self.work = dist.all_gather_into_tensor(self.gathered, mine, async_op=True)
        self._keep = (send, recv, mine)                     # alive until the collectives have run

    @staticmethod
    def _owner_sum(recv: torch.Tensor, ws: int, s: int) -> torch.Tensor:
        if recv.is_cuda:
            from ... import _lib as L                       # the HIP library: loud failure when it is missing
            mine = torch.empty(s, dtype=recv.dtype, device=recv.device)
            L.check(L.lib.rmcl_shard_sum(L.C.c_void_p(recv.data_ptr()), L.BF16 if recv.dtype == torch.bfloat16 else L.F32, ws, s, None,
                                         L.C.c_void_p(mine.data_ptr()), L.C.c_void_p(torch.cuda.current_stream().cuda_stream)), "shard_sum")
            return mine
        acc = recv[:s].float().clone()                      # CPU ranks of the gloo tests: same rank-order fp32 sum
        for w in range(1, ws):
            acc += recv[w * s:(w + 1) * s].float()
        return acc.to(recv.dtype)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.gathered is not self.chunk:
                self.chunk.copy_(self.gathered[:self.n])
            self._keep = None

    wait = finish


def allreduce_mean_(flat: torch.Tensor, bucket_elems: int = 32 * 1024 * 1024, async_op: bool = False):
    """Gradient averaging over a flat fp32 arena in a few large buckets (xGMI is point-to-point:
    few, large collectives).  Returns the list of work handles when async_op."""
    ws = world_size()
    if ws == 1:
        return []
    works = []
    n = flat.numel()
    for s in range(0, n, bucket_elems):
        chunk = flat[s:min(n, s + bucket_elems)]
        chunk.div_(ws)
        w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def allreduce_sum_(flat: torch.Tensor, bucket_elems: int = 32 * 1024 * 1024, force: bool = False, algo: str = "ring",
                   wire: Optional[torch.dtype] = None):
    """Blocking bucketed SUM over the flat gradient arena (the 1/world_size is applied to the loss gradient beforehand).
    ``force``: issue the collectives even in a 1-rank group (single-GPU rehearsal of the communication path).
    ``algo``: "ring" = RCCL all-reduce per bucket; "direct" = DirectReduce per bucket (wire: its transport dtype)."""
    if world_size() == 1 and not force:
        return
    n = flat.numel()
    pending = []
    for s in range(0, n, bucket_elems):
        chunk = flat[s:min(n, s + bucket_elems)]
        if algo == "direct":
            pending.append(DirectReduce(chunk, wire))
        elif algo == "ring":
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
        else:
            raise ValueError(f"grad_allreduce_algo must be 'ring' or 'direct', got {algo!r}")
    for p in pending:
        p.finish()


def grad_buckets(layer0: int, layer_stride: int, layers: int, total: int):
    """Buckets of the flat gradient arena in the order their gradients become final during the backward:
    (gate_layer, start, end).  gate_layer = l: ready once encoder layer l's backward has finished; -1: ready only
    when the whole backward has (embeddings).  Layer L-1's bucket also carries everything behind the layers
    (final norm, pooler, heads), which the backward writes first."""
    out = []
    end_layers = layer0 + layers * layer_stride
    for l in range(layers - 1, -1, -1):
        s = layer0 + l * layer_stride
        e = s + layer_stride if l < layers - 1 else max(total, end_layers)
        out.append((l, s, e))
    if layer0 > 0:
        out.append((-1, 0, layer0))
    return out


class GradSync:
    """Overlapped data-parallel gradient reduction (replaces DDP's bucketed all-reduce hooks, run.py:96; SUM - the
    1/world_size is applied to the loss gradient before the backward, `prescaled=True`): one all-reduce
    per encoder layer (28 MB fp32 - xGMI rings want few, large messages), issued on a communication stream that waits
    on that layer's gradient-ready events (rmcl_grad_ready_wait), so RCCL runs while the layers below are still in
    their backward.  ``wait()`` makes the current stream wait for all of them (call before the optimizer step)."""

    def __init__(self, flat: torch.Tensor, buckets, comm_stream, gate, prescaled: bool = False, compress: str = None,
                 algo: str = "ring"):
        """compress="bf16": every bucket travels as bf16 (half the bytes over the xGMI links: 224 instead of 449 MB per step
        for ViLT-B/32) - cast, SUM all-reduce, cast back into the fp32 arena; the sum itself is then rounded to bf16 (the
        usual mixed-precision DDP trade; default None keeps fp32 buckets = the reference's DDP arithmetic).
        algo="direct": every bucket goes through DirectReduce (one-hop reduce-scatter + all-gather) instead of RCCL's
        all-reduce; with compress="bf16" both of its legs travel as bf16."""
        if algo not in ("ring", "direct"):
            raise ValueError(f"grad_allreduce_algo must be 'ring' or 'direct', got {algo!r}")
        self.works = []
        self.unpack = []                       # (bf16 buffer, fp32 destination) pairs copied back in wait()
        self.comm_stream = comm_stream if flat.is_cuda else None
        StepTimers.info["grad_sync"] = {"mode": "overlapped per-layer buckets", "algo": algo, "wire": compress or "f32", "buckets": len(buckets),
                                        "bucket_bytes_max": 4 * max((e - s) for _, s, e in buckets) if buckets else 0,
                                        "bytes_total": 4 * sum((e - s) for _, s, e in buckets)}
        if not (dist.is_available() and dist.is_initialized()):
            return
        ws = world_size()
        main = torch.cuda.current_stream() if flat.is_cuda else None

        def reduce(chunk):
            if not prescaled:
                chunk.div_(ws)
            if algo == "direct":
                self.works.append(DirectReduce(chunk, torch.bfloat16 if compress == "bf16" else None))
            elif compress == "bf16":
                buf = chunk.to(torch.bfloat16)
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
                self.unpack.append((buf, chunk))
            else:
                self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

        for layer, s, e in buckets:
            chunk = flat[s:e]
            if flat.is_cuda:
                if layer >= 0:
                    gate(layer, comm_stream)
                else:
                    comm_stream.wait_stream(main)
                with torch.cuda.stream(comm_stream):
                    reduce(chunk)
            else:
                reduce(chunk)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        for buf, dst in self.unpack:
            dst.copy_(buf)
        self.unpack = []


class StepGradSync:
    """Reduces the gradient arena over ranks EXACTLY ONCE per backward pass of a training step, whatever the number of
    deferred-backward closures that accumulate into it (itm + moco, the three RMCL views, gradient accumulation).

    Why: the HIP backward kernels accumulate (``dW +=``) into one flat arena.  Reducing after every closure would sum
    the already-reduced part over ranks again (g = W^2 R_a + W R_b ... instead of the mean).  So closures only report
    that they ran; the SUM all-reduce is launched when the last closure of the step has been enqueued.  The 1/world_size
    of DDP's mean (run.py:96) is applied to every closure's incoming loss gradient (``prescale``), including micro-steps
    whose reduction is skipped (``enabled=False``: gradient accumulation, DDP ``no_sync`` semantics).

    ``overlap``: a factory returning a ``GradSync`` (per-layer buckets gated on the backward's events); used only when the
    step has a single closure, because the per-layer events belong to ONE backward."""

    def __init__(self, algo: str = "ring", compress: Optional[str] = None):
        """algo / compress: config["grad_allreduce_algo"] ("ring" | "direct") and config["grad_allreduce_dtype"] (None |
        "bf16", honoured by the direct form and by the overlapped per-layer buckets) for the blocking pass."""
        self.open = 0
        self.created = 0
        self.handle = None
        self.algo = algo
        self.compress = compress

    def begin_step(self):
        self.open = 0
        self.created = 0

    def register(self) -> float:
        """A loss with a deferred backward was created; returns the gradient prescale for its closure."""
        self.open += 1
        self.created += 1
        return 1.0 / world_size()

    def closure_done(self, flat: torch.Tensor, enabled: bool = True, overlap=None):
        if self.open <= 0:
            # two forwards before their backwards: begin_step() of the second forgot the first one's closure, the first backward
            # has already reduced the arena, and this one would add the summed arena over the ranks again (gradients x world size)
            why = ("the first backward has already reduced the gradient arena over the ranks and this one would add the summed arena again "
                   "(gradients x world size)") if (dist.is_available() and dist.is_initialized() and world_size() > 1) else \
                  "the step's closure count no longer matches its backwards"
            raise RuntimeError("StepGradSync: a deferred backward ran that the current step did not register - a loss of an EARLIER "
                               f"forward was back-propagated after the next forward began ({why}); run every loss.backward() of a step "
                               "before the next training_step (or accumulate with sync_grads=False)")
        self.open -= 1
        if self.open > 0 or not enabled:
            return
        if not (dist.is_available() and dist.is_initialized()):
            return
        self.wait()
        if overlap is not None and self.created == 1:
            self.handle = overlap()
        else:
            StepTimers.info["grad_sync"] = {"mode": "one blocking pass", "algo": self.algo, "wire": self.compress or "f32",
                                            "buckets": -(-flat.numel() // (32 * 1024 * 1024)), "bytes_total": 4 * flat.numel()}
            with StepTimers.bracket("comm_exposed"):         # blocking pass on the compute stream: all of it is exposed
                allreduce_sum_(flat, force=True, algo=self.algo,
                               wire=torch.bfloat16 if (self.algo == "direct" and self.compress == "bf16") else None)

    def wait(self):
        if self.handle is not None:
            with StepTimers.bracket("comm_exposed"):
                self.handle.wait()
            self.handle = None
