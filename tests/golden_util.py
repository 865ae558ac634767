"""Helpers shared by the golden-fixture tests (data only: the .npz files were written by
oracle/gen_golden.py from the reference's own arithmetic)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def digest(t: torch.Tensor) -> np.ndarray:
    f = t.detach().double().flatten().cpu()
    head = torch.zeros(8, dtype=torch.float64)
    head[: min(8, f.numel())] = f[:8]
    return torch.cat([torch.stack([f.sum(), f.norm(), f.abs().max()]), head]).numpy()


def cfg_from_meta(O, meta, kind="moco"):
    if kind == "moco":
        B, seed_w, seed_b, ragged, L, Kq, K = [int(x) for x in meta]
        cfg = O.default_config(num_layers=L, num_negative=Kq, adv_steps_img=K, per_gpu_batchsize=B)
    else:
        B, seed_w, seed_b, ragged, L = [int(x) for x in meta]
        cfg = O.default_config(num_layers=L, per_gpu_batchsize=B)
    return cfg, B, seed_w, seed_b, bool(ragged)
