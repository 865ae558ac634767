"""Training utilities on the hot path's edge: ``set_task`` (vilt/modules/vilt_utils.py:325-329) and
``set_schedule`` (:331-437) = HF AdamW (beta (0.9,0.98), eps 1e-8, 4 parameter groups) + polynomial
decay with warm-up, restated as ONE fused multi-tensor kernel over the flat parameter arena."""
from __future__ import annotations

import ctypes as C
import math

import torch

from ... import _lib as L
from ..._lib import lib, check, P, I64, F
from ...runtime import stream_ptr
from . import schedules

NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight"]
HEAD_NAMES = ["vqa_classifier", "nlvr2_classifier", "moco_head", "barlowtwinshead"]


def set_task(pl_module):
    pl_module.current_tasks = [k for k, v in pl_module.hparams.config["loss_names"].items() if v >= 1]
    return


def epoch_wrapup(pl_module):
    """vilt_utils.epoch_wrapup (:227-323) for the tasks on this path: the reference computes and logs the epoch value of
    every task metric and resets it.  Metrics here are the per-step values kept in ``pl_module.logged``; the wrap-up
    returns them as ``the_metric`` inputs and clears them.  (As written, the reference's own wrap-up reads undefined
    ``text_attack`` / ``image_attack`` attributes, :236 - not imitated.)"""
    phase = "train" if pl_module.training else "val"
    out = {k: (float(v) if torch.is_tensor(v) and v.numel() == 1 else v) for k, v in pl_module.logged.items()}
    pl_module.logged = {}
    pl_module.last_epoch_metrics = {"phase": phase, **out}
    return pl_module.last_epoch_metrics


class FusedAdamW:
    """optimizer-like object (``step`` / ``zero_grad`` / ``param_groups``) over the engine's arenas."""

    def __init__(self, pl_module, lr, wd, lr_mult, betas=(0.9, 0.98), eps=1e-8):
        eng = pl_module.engine
        self.eng = eng
        self.module = pl_module
        self.betas, self.eps = betas, eps
        self.m = torch.zeros_like(eng.q32)
        self.v = torch.zeros_like(eng.q32)
        self.t = 0
        ends, mults, wds = [], [], []
        specs = sorted(eng.specs, key=lambda s: s[1])
        for i, (name, off, shape) in enumerate(specs):
            end = specs[i + 1][1] if i + 1 < len(specs) else eng.total
            decay = not any(nd in name for nd in NO_DECAY)
            head = any(bb in name for bb in HEAD_NAMES)
            ends.append(end)
            mults.append(lr_mult if head else 1.0)
            wds.append(wd if decay else 0.0)
        dev = eng.device
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
        self.seg_mult = torch.tensor(mults, dtype=torch.float32, device=dev)
        self.seg_wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": lr, "initial_lr": lr}]
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none: bool = False):
        self.eng.zero_grads()

    def step(self):
        self.t += 1
        e = self.eng
        self.module.wait_grad_sync()                       # overlapped data-parallel all-reduces (N > 1)
        check(lib.rmcl_adamw_f32(P(e.q32), P(e.g32), P(self.m), P(self.v), P(e.q_lp), P(self.seg_end), P(self.seg_mult),
                                 P(self.seg_wd), int(self.seg_end.numel()), F(self.param_groups[0]["lr"]), F(self.betas[0]),
                                 F(self.betas[1]), F(self.eps), self.t, F(self.grad_scale), I64(e.q32.numel()), stream_ptr()),
              "adamw")
        e.fold_stale["q"] = True
        e.lpT_stale = True


class PolySchedule:
    """get_polynomial_decay_schedule_with_warmup as called at vilt_utils.py:423-430; step() per iteration."""

    def __init__(self, opt, warmup, total, end_lr, power):
        self.opt, self.warmup, self.total, self.end_lr, self.power = opt, warmup, total, end_lr, power
        self.base = opt.param_groups[0]["initial_lr"]
        self.n = 0
        self._apply()

    def lr_at(self, step):
        return schedules.poly_lr(step, self.base, self.warmup, self.total, self.end_lr, self.power)

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.lr_at(self.n)

    def step(self):
        self.n += 1
        self._apply()


class CosineSchedule(PolySchedule):
    """get_cosine_schedule_with_warmup (vilt_utils.py:417-421)."""

    def lr_at(self, step):
        return schedules.cosine_lr(step, self.base, self.warmup, self.total)


def set_schedule(pl_module):
    cfg = pl_module.hparams.config
    if cfg["optim_type"] != "adamw":
        raise NotImplementedError("only optim_type='adamw' (the reference default) is built")
    opt = FusedAdamW(pl_module, cfg["learning_rate"], cfg["weight_decay"], cfg["lr_mult"])
    max_steps = cfg["max_steps"]
    if max_steps is None:
        # the reference derives it from the Lightning trainer's dataloader (vilt_utils.py:404-411); there is no
        # trainer here, so the caller states it
        raise ValueError("set_schedule: config['max_steps'] is None - set it to the number of optimizer steps "
                         "(the reference computes it from len(train_dataloader) * max_epoch // accumulate_grad_batches)")
    warmup = cfg["warmup_steps"]
    if isinstance(warmup, float):
        warmup = int(max_steps * warmup)
    if cfg["decay_power"] == "cosine":
        sched = CosineSchedule(opt, warmup, max_steps, cfg["end_lr"], 1)
    else:
        sched = PolySchedule(opt, warmup, max_steps, cfg["end_lr"], cfg["decay_power"])
    return [opt], [{"scheduler": sched, "interval": "step"}]
