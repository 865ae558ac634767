"""Helpers for the -m gpu parity tests: call the C ABI on torch device tensors."""
import ctypes as C

import torch

import rmcl_pkg  # noqa: F401
from rmcl_amd import _lib as L
from rmcl_amd._lib import lib, check, P, I64, F

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tdt(dt):
    return torch.float32 if dt == L.F32 else torch.bfloat16


def gemm(A, B, M, N, K, a_kc, b_kc, dt_in, dt_out, lda=None, ldb=None, bias=None, aux=None, ld_aux=0, epi=0, splitk=1,
         alpha=1.0, exact=1, C_init=None, want_c2=False):
    lda = lda if lda is not None else A.shape[-1]
    ldb = ldb if ldb is not None else B.shape[-1]
    Cm = torch.zeros(M, N, dtype=tdt(dt_out), device=DEV) if C_init is None else C_init.clone()
    C2 = torch.zeros(M, N, dtype=tdt(dt_out), device=DEV) if want_c2 else None
    check(lib.rmcl_gemm(P(A), P(B), P(Cm), P(C2), P(bias), P(aux), M, N, K, I64(lda), I64(ldb), N, ld_aux, F(alpha), epi,
                        splitk, dt_in, dt_out, int(a_kc), int(b_kc), int(exact), stream()), "gemm")
    return (Cm, C2) if want_c2 else Cm
