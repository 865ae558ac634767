"""Run one GEMM shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from tests.gpu_util import L, lib
import torch
M, N, K, akc, bkc = [int(x) for x in sys.argv[1:6]]
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else 2
lib.rmcl_tune_set(0, cfg)
ms, tf = bench(M, N, K, akc, bkc, L.BF16, 1, iters=5)
print(ms, tf)
