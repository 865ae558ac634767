"""PGD loop (K = 3) of the benchmark step alone, as one chain and as two half-batch lanes (Engine.lanes); also the lanes' forward and
backward passes alone.  RMCL_LANES / RMCL_LANE_LAG_US as in the product path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmcl_pkg  # noqa: F401
from rmcl_amd import _lib as L
from rmcl_amd.vilt.config import task_moco
from rmcl_amd.vilt.modules import ViLTransformerSS
from bench import synthetic_batch

cfg = task_moco(per_gpu_batchsize=64, num_gpus=1, num_nodes=1, adv_steps_img=3, drop_rate=0.0, image_view=True, max_steps=100000, dense_images=True)
m = ViLTransformerSS(cfg, device="cuda:0", compute_dtype="bf16"); m.train()
batch = synthetic_batch(cfg, 64, 1, "cuda:0")
eng = m.engine
eng.dropout_on = False
pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
pb.k.normal_()
pb.k.div_(pb.k.norm(dim=1, keepdim=True))
atk = m.pgd_attacker


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pgd():
    atk.attack_patches(m, pb, None, clean_op=eng.make_operand(pb))


if os.environ.get("LANES_ATTN_WG"):
    L.check(L.lib.rmcl_tune_set(8, int(os.environ["LANES_ATTN_WG"])))       # persistent attention backward with this many workgroups
for lanes, cnt in (("0", 2), ("1", 2), ("1", 4), ("0", 2), ("1", 2), ("1", 4)):
    os.environ["RMCL_LANES"] = lanes
    os.environ["RMCL_LANE_COUNT"] = str(cnt)
    print(f"PGD loop, lanes={lanes} x {cnt}: {timed(pgd):.3f} ms")
os.environ["RMCL_LANE_COUNT"] = "2"

# forward / backward alone
os.environ["RMCL_LANES"] = "1"
ls = eng.lanes(pb)
op = eng.make_operand(pb)
per = ls[0].B * pb.d.P
main, side = torch.cuda.current_stream(), eng.side_stream
L.check(L.lib.rmcl_tune_set(10, 1))


def fwd_one():
    eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op, cls_tail=True)


def bwd_one():
    eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=pb.gpatch)


def fwd_lanes():
    side.wait_stream(main)
    for i, ln in enumerate(ls):
        with torch.cuda.stream((main, side)[i]):
            eng.encoder_forward(ln, key=False, mode=L.MODE_DATA, patchesT=op[i * per:(i + 1) * per], cls_tail=True)
    main.wait_stream(side)


def bwd_lanes():
    side.wait_stream(main)
    for i, ln in enumerate(ls):
        with torch.cuda.stream((main, side)[i]):
            eng.encoder_backward(ln, L.MODE_DATA, op[i * per:(i + 1) * per], ln.dcls, cls_only=True, dpatches=ln.gpatch)
    main.wait_stream(side)


fwd_one(); eng.heads_forward(pb, key=False, wgrad=False); eng.infonce(pb, 1.0, want_dq=True, metrics=False); eng.heads_backward(pb, pb.dq, None, with_grads=False)
print(f"forward, one chain: {timed(fwd_one):.3f} ms   backward: {timed(bwd_one):.3f} ms")
L.check(L.lib.rmcl_tune_set(10, 2))
fwd_lanes()
for ln in ls:
    eng.heads_forward(ln, key=False, wgrad=False); eng.infonce(ln, 1.0, want_dq=True, metrics=False); eng.heads_backward(ln, ln.dq, None, with_grads=False)
print(f"forward, two lanes: {timed(fwd_lanes):.3f} ms   backward: {timed(bwd_lanes):.3f} ms")
