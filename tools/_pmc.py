import sys; sys.path.insert(0, ".")
from tools.gemm_bench import bench
from tests.gpu_util import L, lib
lib.rmcl_tune_set(0, 60)
for name, N, K, dto, epi in [("fc2", 768, 3072, L.F32, 1), ("fc1", 3072, 768, L.BF16, 3)]:
    ms, tf = bench(11840, N, K, 1, 1, dto, epi, iters=5)
    print(f"{name} {ms*1e3:.1f}us {tf:.0f}TF", flush=True)
