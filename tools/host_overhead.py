"""How long does the host take to ENQUEUE one step (no sync) vs the GPU to execute it?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmcl_pkg
from rmcl_amd.vilt.config import task_moco
from rmcl_amd.vilt.modules import ViLTransformerSS
from bench import synthetic_batch
# dense_images=True like bench.py: the general visual_embed path reads the patch counts back once per step (a host sync that would
# make "enqueue time" equal the GPU time)
cfg = task_moco(per_gpu_batchsize=64, num_gpus=1, num_nodes=1, adv_steps_img=3, drop_rate=0.0, image_view=True, max_steps=100000, dense_images=True)
m = ViLTransformerSS(cfg, device="cuda:0", compute_dtype="bf16"); m.train()
(opt,), (sched,) = m.configure_optimizers()
batch = synthetic_batch(cfg, 64, 1, "cuda:0")
def step(i):
    loss = m.training_step(batch, i); loss.backward(); opt.step(); opt.zero_grad()
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/5:.1f} ms/step; total {1e3*(t2-t0)/5:.1f} ms/step")
