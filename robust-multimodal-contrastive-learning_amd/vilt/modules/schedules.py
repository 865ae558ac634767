"""Learning-rate schedules of ``vilt_utils.set_schedule`` (vilt/modules/vilt_utils.py:404-432): the reference calls
``transformers.get_polynomial_decay_schedule_with_warmup`` (``decay_power`` numeric) or
``get_cosine_schedule_with_warmup`` (``decay_power == "cosine"``), stepped once per iteration.  Pure functions of the step
index; pinned against transformers.optimization's own curves in tests/golden/schedules.npz."""
from __future__ import annotations

import math


def poly_lr(step: int, base_lr: float, warmup: int, total: int, end_lr: float, power: float) -> float:
    if step < warmup:
        return base_lr * (float(step) / float(max(1, warmup)))
    if step > total:
        return base_lr * (end_lr / base_lr)
    lr_range = base_lr - end_lr
    decay_steps = total - warmup
    pct_remaining = 1 - (step - warmup) / decay_steps
    decay = lr_range * pct_remaining ** power + end_lr
    return base_lr * (decay / base_lr)


def cosine_lr(step: int, base_lr: float, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    if step < warmup:
        return base_lr * (float(step) / float(max(1, warmup)))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
