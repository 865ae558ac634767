"""Host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (csrc/build_host_asan.sh; SURVEY section 5
"Race detection / sanitizers": GPU sanitizers are not available on this pool, so the sanitizers run on the CPU build of the two
pure-host translation units - the pass orchestration / workspace carving of encoder.cpp and the C ABI of api.cpp).  A child
process preloads the ASan runtime, loads lib/librmcl_hip_asan.so and drives every entry point that needs no GPU: layout and size
queries over a range of shapes, GEMM routing, tuning switches, and the argument-validation path of the compute entry points (NULL
operands, bad shapes: they must return an error code and a message, never touch memory).  Any sanitizer report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "robust-multimodal-contrastive-learning_amd", "lib", "librmcl_hip_asan.so")

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.environ["RMCL_ROOT"])
import rmcl_pkg
from rmcl_amd import _lib as L
lib = L.lib
assert os.path.basename(L.LIB_PATH) == "librmcl_hip_asan.so", L.LIB_PATH
n_err = 0
def expect_error(rc, what):
    global n_err
    assert rc != 0, what
    msg = lib.rmcl_last_error()
    assert msg and len(msg) > 4, what
    n_err += 1
sizes = []
for B, Lt, P, layers, dt in ((64, 40, 144, 12, L.BF16), (2, 40, 144, 12, L.F32), (3, 40, 132, 2, L.BF16), (1, 8, 4, 1, L.F32), (64, 40, 200, 12, L.BF16)):
    d = L.Dims(B=B, L=Lt, P=P, D=768, H=12, layers=layers, mlp=3072, patch_k=3072, proj=128, vocab=30522, dtype=dt, exact=0, Pp=144)
    lay = L.Layout()
    lib.rmcl_param_layout(C.byref(d), C.byref(lay))
    assert 0 < lay.ema_end < lay.total and lay.layer_stride > 0
    for mode in (L.MODE_INFER, L.MODE_DATA, L.MODE_FULL):
        sizes.append(lib.rmcl_stash_bytes(C.byref(d), mode))
    sizes.append(lib.rmcl_workspace_bytes(C.byref(d)))
    sizes.append(lib.rmcl_heads_stash_bytes(C.byref(d)))
    sizes.append(lib.rmcl_ln_fold_elems(C.byref(d), 0) + lib.rmcl_ln_fold_elems(C.byref(d), 1))
    # argument validation of the pass entry points: every pointer NULL -> error before anything is dereferenced
    expect_error(lib.rmcl_encoder_forward(C.byref(d), L.MODE_FULL, None, None, None, None, None, None, None, None, None, 0, C.c_float(0.0), None, None, None), "encoder_forward")
    expect_error(lib.rmcl_encoder_backward(C.byref(d), L.MODE_FULL, None, None, None, None, None, None, None, None, 0, None, None, None, 0, C.c_float(0.0), None, None, None), "encoder_backward")
    expect_error(lib.rmcl_heads_forward(C.byref(d), None, None, None, None, None, None, None), "heads_forward")
assert all(s >= 0 for s in sizes) and max(sizes) > 1 << 30
bad = L.Dims(B=1, L=40, P=144, D=700, H=12, layers=12, mlp=3072, patch_k=3072, proj=128, vocab=30522, dtype=L.BF16, exact=0, Pp=144)
one = (C.c_float * 4)()
expect_error(lib.rmcl_encoder_forward(C.byref(bad), L.MODE_INFER, one, one, one, one, one, one, None, one, one, 0, C.c_float(0.0), None, None, None), "bad dims")
for M, N, K, epi, dto in ((11840, 2304, 768, L.EPI_LNFOLD, L.BF16), (11840, 768, 3072, L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_ROWSTAT, L.F32),
                          (740, 768, 768, L.EPI_BIAS, L.BF16), (11840, 3072, 768, L.EPI_DGELU, L.BF16), (64, 128, 768, 0, L.F32), (1, 192, 64, 0, L.BF16)):
    for cfg in (-1, 60, 70, 80, 1):
        lib.rmcl_tune_set(0, cfg)
        assert 0 <= lib.rmcl_gemm_route(M, N, K, epi, dto, 1, 1) <= 5
lib.rmcl_tune_set(0, -1)
assert lib.rmcl_gemm_route(11840, 2304, 768, L.EPI_LNFOLD, L.BF16, 1, 1) == 1 and lib.rmcl_gemm_route(11840, 3072, 768, L.EPI_BIAS | L.EPI_GELU, L.BF16, 1, 1) == 2
for key, val in ((1, 8), (1, -5), (1, 1000), (2, 0), (2, 1), (3, 1), (4, 8), (5, 1), (6, 1), (7, 0)):
    assert lib.rmcl_tune_set(key, val) == 0
expect_error(lib.rmcl_tune_set(99, 1), "unknown tune key")
expect_error(lib.rmcl_gemm(None, None, None, None, None, None, 4, 4, 4, C.c_int64(4), C.c_int64(4), 4, 0, C.c_float(1.0), 0, 1, 0, 0, 1, 1, 1, None), "gemm NULL")
expect_error(lib.rmcl_infonce_f32(None, None, None, 4, 128, C.c_int64(1024), C.c_float(0.07), C.c_float(1.0), None, None, None, None, None), "infonce NULL")
expect_error(lib.rmcl_infonce_split_bf16(None, None, None, 4, 128, C.c_int64(1024), C.c_float(0.07), C.c_float(1.0), None, None, None, None, 1, None), "infonce NULL")
expect_error(lib.rmcl_image_u8_to_patches(None, None, None, None, 0, 4, 144, 384, 384, 32, None, None, None), "u8 NULL")
expect_error(lib.rmcl_grad_ready_wait(-1, None), "grad_ready_wait")
expect_error(lib.rmcl_pgd_step(None, 0, None, None, 4, C.c_int64(16), C.c_float(0.1), C.c_float(0.1), None), "pgd NULL")
assert lib.rmcl_infonce_ws_bytes(64, C.c_int64(65536)) > 0 and lib.rmcl_attention_scratch_elems(64, 12, 185) > 0
print("SANITIZED_HOST_OK", n_err)
'''


def _asan_runtime():
    try:
        out = subprocess.run(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, timeout=60).stdout.strip()
        return out if os.path.isfile(out) else None
    except Exception:
        return None


def test_host_code_is_clean_under_asan_and_ubsan():
    rt = _asan_runtime()
    if not os.path.isfile(LIB) or rt is None:
        pytest.skip("lib/librmcl_hip_asan.so not built (csrc/build_host_asan.sh; __graft_entry__.build() builds it where hipcc is present)")
    env = dict(os.environ, LD_PRELOAD=rt, RMCL_LIB=LIB, RMCL_ROOT=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_HOST_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
