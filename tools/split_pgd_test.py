"""Experiment: the PGD loop (K x [forward, InfoNCE, data-gradient backward, update]) on the full batch in one stream vs as
two independent half-batches on two HIP streams (samples are independent through the whole loop)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmcl_pkg  # noqa
from rmcl_amd import _lib as L
from rmcl_amd.vilt.config import task_moco
from rmcl_amd.vilt.modules import ViLTransformerSS
from bench import synthetic_batch

dev = "cuda:0"
B, K = 64, 3
cfg = task_moco(per_gpu_batchsize=B, num_gpus=1, num_nodes=1, adv_steps_img=K, drop_rate=0.0, image_view=True, text_view=False, max_steps=1000)
m = ViLTransformerSS(cfg, device=dev, compute_dtype="bf16")
m.train()
eng = m.engine
batch = synthetic_batch(cfg, B, 1, dev)
att = m.pgd_attacker
kk = torch.nn.functional.normalize(torch.randn(B, 128, device=dev), dim=1)


def full():
    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
    att.attack_patches(m, pb, kk)
    return pb.delta


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def split(nsplit=2):
    h = B // nsplit
    main = torch.cuda.current_stream()
    pbs = []
    for i in range(nsplit):
        sl = slice(i * h, (i + 1) * h)
        pbs.append(eng.bind_batch(batch["text_ids"][sl], batch["text_masks"][sl], batch["image"][0][sl], tag=f"h{i}"))
        pbs[-1].k.copy_(kk[sl])
    for i in range(nsplit):
        streams[i].wait_stream(main)
    Kk = att.adv_steps_img
    for pb in pbs:
        pb.delta.zero_(); pb.delta_prev.zero_()
    for i in range(nsplit):
        streams[i].wait_stream(main)
    for step in range(Kk):
        for i, pb in enumerate(pbs):
            with torch.cuda.stream(streams[i]):
                op = eng.make_operand(pb, pb.delta)
                eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op)
                eng.heads_forward(pb, key=False)
                eng.infonce(pb, grad_scale=1.0 / (B * Kk), want_dq=True)
                eng.heads_backward(pb, pb.dq, None, with_grads=False)
                eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=pb.gpatch)
                eng.pgd_step(pb, att.adv_lr_img, att.adv_max_norm_img)
    for i in range(nsplit):
        main.wait_stream(streams[i])
    return torch.cat([pb.delta for pb in pbs])


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


t_full, d_full = timeit(full)
t_split, d_split = timeit(split)
print(f"full batch, one stream : {t_full:.3f} ms")
print(f"two halves, two streams: {t_split:.3f} ms   max |delta diff| {float((d_full - d_split).abs().max()):.3e} "
      f"same-saturation {(float(((d_full.abs() > 0.00499) == (d_split.abs() > 0.00499)).float().mean())):.4f}")
for res in (0, 8):
    L.lib.rmcl_tune_set(1, res)
    t, _ = timeit(split)
    print(f"two halves, reserve {res} CUs: {t:.3f} ms")
