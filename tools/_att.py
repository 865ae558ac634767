import sys; sys.path.insert(0, ".")
import torch, ctypes as C
from tests.gpu_util import DEV, L, lib, check, P, stream
B, N, H = 64, 185, 12
g = torch.Generator(device="cpu").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(DEV).to(torch.bfloat16)
mask = torch.ones(B, N, dtype=torch.int32, device=DEV)
out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=DEV)
n = int(lib.rmcl_attention_scratch_elems(B, H, N))
probs = torch.empty(n, dtype=torch.float32, device=DEV); scores = torch.empty(n, dtype=torch.float32, device=DEV)
def run():
    check(lib.rmcl_attention_fwd(P(qkv), P(mask), P(out), P(probs), P(scores), B, N, H, L.BF16, 0, stream()))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"attn fwd {e0.elapsed_time(e1)/50*1e3:.1f} us", flush=True)
