"""CPU: the C-ABI library loads and exports every symbol include/rmcl.h declares (no compute calls)."""
import os
import re

import rmcl_pkg  # noqa: F401
from rmcl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rmcl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmcl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(_lib.lib, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == syms


def test_version_and_layout_are_host_only_calls():
    import ctypes as C
    assert _lib.lib.rmcl_version() == 1
    d = _lib.Dims(B=2, L=40, P=144, D=768, H=12, layers=12, mlp=3072, patch_k=3072, proj=128, vocab=30522, dtype=0, exact=1)
    lay = _lib.Layout()
    _lib.lib.rmcl_param_layout(C.byref(d), C.byref(lay))
    # 112 285 440 trainable parameters of ViLT-B/32 + heads plus <=63 pad elements per tensor
    n_tensors = 10 + 12 * 12 + 2 + 5 + 2 + 2
    exact = 112285440 + 2 * 768 + 2                                  # + itm head (not in the SURVEY count)
    assert exact <= lay.total <= exact + 64 * n_tensors
    assert lay.ema_end == lay.pool_w < lay.itm_w < lay.total and lay.ema_end == 111694848
    assert _lib.lib.rmcl_stash_bytes(C.byref(d), _lib.MODE_FULL) > _lib.lib.rmcl_stash_bytes(C.byref(d), _lib.MODE_DATA) > 0
    assert _lib.lib.rmcl_stash_bytes(C.byref(d), _lib.MODE_INFER) == 256


def test_argument_errors_are_reported_not_crashed():
    import ctypes as C
    rc = _lib.lib.rmcl_gemm(None, None, None, None, None, None, 1, 1, 1, C.c_int64(1), C.c_int64(1), 1, 1,
                            C.c_float(1.0), 0, 1, 0, 0, 1, 1, 1, None)
    assert rc == -1
    assert b"NULL" in _lib.lib.rmcl_last_error()
