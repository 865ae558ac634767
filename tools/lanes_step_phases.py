"""Where a training step spends its time with and without the half-batch lanes: events around the PGD front (key forward + K iterations)
and the rest (attacked view forward, full backward, optimizer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmcl_pkg  # noqa: F401
from rmcl_amd.vilt.config import task_moco
from rmcl_amd.vilt.modules import ViLTransformerSS
from bench import synthetic_batch

cfg = task_moco(per_gpu_batchsize=64, num_gpus=1, num_nodes=1, adv_steps_img=3, drop_rate=0.0, image_view=True, max_steps=100000, dense_images=True)
m = ViLTransformerSS(cfg, device="cuda:0", compute_dtype="bf16"); m.train()
(opt,), _ = m.configure_optimizers()
batch = synthetic_batch(cfg, 64, 1, "cuda:0")
atk = m.pgd_attacker
orig = atk.attack_patches
marks = []


def wrapped(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    r = orig(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    marks.append((e0, e1))
    return r


atk.attack_patches = wrapped


def step(i):
    s = torch.cuda.Event(enable_timing=True); s.record()
    loss = m.training_step(batch, i); loss.backward(); opt.step(); opt.zero_grad()
    e = torch.cuda.Event(enable_timing=True); e.record()
    return s, e


for lanes in ("0", "auto", "0", "auto"):
    os.environ["RMCL_LANES"] = lanes
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    marks.clear()
    evs = [step(i) for i in range(10)]
    torch.cuda.synchronize()
    tot = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    front = sum(s.elapsed_time(m1) for (s, _), (_, m1) in zip(evs, marks)) / len(evs)
    pgd = sum(a.elapsed_time(b) for a, b in marks) / len(marks)
    print(f"lanes={lanes:4s}: step {tot:.2f} ms = start -> end of PGD {front:.2f} (attack_patches itself {pgd:.2f}) + rest {tot - front:.2f}")
