"""Objectives of the RMCL hot path.  Mirrors the free-function protocol of the reference's
vilt/modules/objectives.py (``compute_<task>(pl_module, batch) -> dict``); all tensor math runs in
librmcl_hip.so through ``pl_module.engine``."""
from __future__ import annotations

from copy import copy

import torch

from ... import _lib as L
from ..._lib import lib, check, P, I64
from ...runtime import stream_ptr
from . import dist_utils


class _DeferredBackward(torch.autograd.Function):
    """Gives a loss value computed by HIP kernels a ``.backward()``: the closure launches the HIP
    backward, which accumulates into the gradient arena that every ``param.grad`` is a view of."""

    @staticmethod
    def forward(ctx, anchor, value, closure):
        ctx.closure = closure
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.closure(grad_out)
        return None, None, None


def _scalar(t):
    return t.reshape(())


def compute_pgd(pl_module, batch, loss_name, k_modality=None):
    """objectives.py:160-188 for loss_name == "moco"."""
    img_delta = pl_module.pgd_attacker.pgd_attack(pl_module, batch, k_modality=k_modality)
    batch["image"][0] = batch["image"][0] + img_delta                     # :176 (on top of img + delta_{K-1})
    phase = "train" if pl_module.training else "val"
    pl_module.log(f"{loss_name}_attack/{phase}/delta", torch.linalg.norm(img_delta, dim=1).mean())
    return batch


def compute_moco_contrastive(pl_module, batch):
    """objectives.py:217-447 (image view).  Returns {"moco_loss", pos_/neg_{dist,cosine,dot}_attacked_img}."""
    eng = pl_module.engine
    if not (pl_module.image_view or pl_module.text_view):
        raise ZeroDivisionError("division by zero: loss / loss_num with both views off (objectives.py:250-251,397)")
    if pl_module.text_view:
        raise NotImplementedError("text view (greedy synonym attack) needs nltk/counter-fitted resources; SURVEY 8(f4)")
    if pl_module.augmentation:
        raise NotImplementedError("augmentation views are out of scope (SURVEY 2.1 #17)")
    phase = "train" if pl_module.training else "val"
    ret = {}

    eng.ema(pl_module.momentum)                                            # :257-260
    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
    B = pb.B

    op = eng.make_operand(pb)
    eng.encoder_forward(pb, key=True, mode=L.MODE_INFER, patchesT=op)      # infer_k under no_grad (:262-265)
    eng.heads_forward(pb, key=True)
    k = pb.k
    gather = dist_utils.KeyGather(k.clone()) if pl_module.training else None   # overlaps everything below

    eng.encoder_forward(pb, key=False, mode=L.MODE_INFER, patchesT=op)     # clean query (:267-275)
    eng.heads_forward(pb, key=False)
    eng.infonce(pb, 0.0, want_dq=False)
    prediction_original = pb.rows[:, 1].clone()
    ret["q_original"] = pb.q.clone()

    loss = 0
    loss_num = 0
    if pl_module.image_view:
        pl_module.pgd_attacker.attack_patches(pl_module, pb, k)            # compute_pgd (:319-323)
        check(lib.rmcl_delta_channel_norm(P(pb.delta), P(pb.loss_sum.zero_()), I64(pb.delta.shape[0]), 3,
                                          pb.d.patch_k // 3, stream_ptr()), "delta_norm")
        pl_module.log(f"moco_attack/{phase}/delta", _scalar(pb.loss_sum / float(pb.delta.numel() // 3)))
        need_grad = torch.is_grad_enabled() and pl_module.training
        # attacked view = img + delta_{K-1} + delta_K  (pgd_attack_vilt.py:144 + objectives.py:176)
        op_att = eng.make_operand(pb, pb.delta_prev, pb.delta, out=pb.patchesT_full)
        eng.encoder_forward(pb, key=False, mode=L.MODE_FULL if need_grad else L.MODE_INFER, patchesT=op_att)
        eng.heads_forward(pb, key=False)
        eng.infonce(pb, 1.0 / B, want_dq=need_grad)
        rows = pb.rows
        if phase == "train":
            pl_module.log("moco_attack/PGD_success_rate", (rows[:, 1] != prediction_original).float().mean())
        for j, name in ((3, "pos_dist"), (4, "pos_cosine"), (5, "pos_dot"), (6, "neg_dist"), (7, "neg_cosine"), (8, "neg_dot")):
            ret[f"{name}_attacked_img"] = rows[:, j].mean()
        ret["q_img_attack"] = pb.q.clone()
        ret["logit_pos_img_attack"] = rows[:, 2].clone()
        ret["lse_img_attack"] = rows[:, 9].clone()
        value = _scalar(pb.loss_sum.clone())
        if need_grad:
            dq_saved = pb.dq.clone()

            def backward(grad_out, pb=pb, dq_saved=dq_saved, op_att=op_att):
                dq = dq_saved * grad_out.to(dq_saved.dtype)
                eng.heads_backward(pb, dq, None, with_grads=True)
                eng.encoder_backward(pb, L.MODE_FULL, op_att, pb.dcls, cls_only=True, dpatches=None)
                pl_module.after_backward()

            loss_attacked_img = _DeferredBackward.apply(pl_module.grad_anchor, value, backward)
        else:
            loss_attacked_img = value
        pl_module.log("moco_loss/attacked_img_loss", loss_attacked_img.detach())
        loss = loss + loss_attacked_img
        loss_num += 1

    if pl_module.training:                                                  # _dequeue_and_enqueue (:394-395)
        keys_all = gather.wait()
        do, new_ptr = dist_utils.queue_advance(pl_module.queue_ptr, keys_all.shape[0], pl_module.num_negative,
                                               pl_module.per_step_bs)
        if do:
            eng.enqueue(keys_all, pl_module.queue_ptr)
            pl_module.queue_ptr = new_ptr

    ret["moco_loss"] = loss / loss_num
    ret["k"] = k.clone()
    pl_module.log(f"moco_loss/step/{phase}", ret["moco_loss"].detach())
    if pl_module.image_view:
        for kind, tag in (("dist", "L2"), ("cosine", "Cosine"), ("dot", "Dot")):
            pos, neg = ret[f"pos_{kind}_attacked_img"], ret[f"neg_{kind}_attacked_img"]
            pl_module.log(f"moco_dist_{phase}_{tag}/Pos_attacked_img", pos)
            pl_module.log(f"moco_dist_{phase}_{tag}/Neg_attacked_img", neg)
            pl_module.log(f"moco_dist_{phase}_{tag}/Neg-Pos_attacked_img", neg - pos)
    return ret
