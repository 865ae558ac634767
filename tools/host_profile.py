"""Where the HOST spends the step's enqueue time (tools/host_overhead.py measures how much: 29 ms per step against 35 ms of GPU time
in round 3): cProfile over 5 enqueue-only steps, cumulative time per Python function, plus wall time per engine entry point."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmcl_pkg  # noqa: F401
from rmcl_amd.vilt.config import task_moco
from rmcl_amd.vilt.modules import ViLTransformerSS
from bench import synthetic_batch

cfg = task_moco(per_gpu_batchsize=64, num_gpus=1, num_nodes=1, adv_steps_img=3, drop_rate=0.0, image_view=True, max_steps=100000, dense_images=True)
m = ViLTransformerSS(cfg, device="cuda:0", compute_dtype="bf16")
m.train()
(opt,), (sched,) = m.configure_optimizers()
batch = synthetic_batch(cfg, 64, 1, "cuda:0")


def step(i):
    loss = m.training_step(batch, i)
    loss.backward()
    opt.step()
    opt.zero_grad()


for i in range(3):
    step(i)
torch.cuda.synchronize()
# wall time of every Engine method (enqueue only)
eng = m.engine
acc = {}
for name in ("bind_batch", "make_operand", "encoder_forward", "heads_forward", "infonce", "heads_backward", "encoder_backward", "pgd_step", "ema",
             "enqueue", "refresh_shadows", "weights_T", "fold_of", "zero_grads"):
    fn = getattr(eng, name)

    def wrap(fn=fn, name=name):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] = acc.get(name, [0, 0.0])
            acc[name][0] += 1
            acc[name][1] += time.perf_counter() - t0
            return r
        return w
    setattr(eng, name, wrap())
t0 = time.perf_counter()
for i in range(5):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:18s} {n / 5:6.1f} calls/step  {1e3 * t / 5:7.3f} ms/step  {1e6 * t / n:8.1f} us/call")
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
