"""CPU (hipcc cross-compiles gfx950 without a GPU): resource budget of the hot kernels, read from the compiler's own assembly.

Round 4 found two things in the ISA that three rounds of counter tables had not shown: epilogues that were instruction-bound (value pairs
mis-grouped by the vectoriser, addresses recomputed per store: DESIGN section 3, Round 4 (5)) and an instantiation that spilled (the folded fc1 form
under dropout: 256 VGPRs + scratch).  This test keeps both from coming back unnoticed:
  * no kernel of the GEMM / LayerNorm / attention sources uses scratch memory;
  * the fc1 form of the 192x384 kernel (LayerNorm-folded, bf16 out, no dropout) stays under a static VALU budget behind its last MFMA.
The budget is a regression guard for THIS toolchain (ROCm 7.2 hipcc), not a portable number."""
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robust-multimodal-contrastive-learning_amd", "csrc")
FILES = ["gemm_sw.hip", "gemm_st.hip", "norm_softmax.hip", "attention.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-x", "hip", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include")]


@pytest.fixture(scope="module")
def isa():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        def one(f):
            dst = os.path.join(tmp, f + ".s")
            r = subprocess.run(["hipcc", *FLAGS, os.path.join(CSRC, f), "-o", dst], capture_output=True, text=True, timeout=1500)
            assert r.returncode == 0, r.stderr[-2000:]
            return f, open(dst).read()
        with ThreadPoolExecutor(max_workers=4) as ex:
            for f, text in ex.map(one, FILES):
                out[f] = text
    return out


def test_no_hot_kernel_uses_scratch(isa):
    spilled = []
    for f, text in isa.items():
        for name, size in re.findall(r"\.set (_Z\S+)\.private_seg_size, (\d+)", text):
            if int(size) != 0:
                spilled.append((f, name[:80], int(size)))
    assert not spilled, spilled


def test_fc1_epilogue_stays_inside_its_instruction_budget(isa):
    text = isa["gemm_sw.hip"].split("\n")
    # gemm_sw_kernel<B_KC = true, AUX = 0, TO = bf16 (t), LNF = 1, DROP = false>
    start = next(i for i, l in enumerate(text) if re.match(r"_Z14gemm_sw_kernelILb1ELi0EtLi1ELb0EE\S*:", l))
    body = []
    for l in text[start + 1:]:
        if l.strip().startswith(".Lfunc_end"):
            break
        body.append(l.strip())
    last = max(i for i, l in enumerate(body) if l.startswith("v_mfma"))
    valu = sum(1 for l in body[last:] if re.match(r"v_", l))
    # four (GELU, stash) instantiations of the chunk loop: 1 620 + 1 300 + 400 + 300 VALU instructions at the time of writing (2 370 for ONE before round 4)
    assert valu < 4500, valu
    assert sum(1 for l in body[last:] if l.startswith("v_pk_fma_f32")) > 500          # the arithmetic is packed
    assert sum(1 for l in body[last:] if l.startswith("v_or_b32_sdwa")) < 40           # no bf16 re-pairing fix-ups (144 before)
