"""Objectives of the RMCL hot path.  Mirrors the free-function protocol of the reference's
vilt/modules/objectives.py (``compute_<task>(pl_module, batch) -> dict``); all tensor math runs in
librmcl_hip.so through ``pl_module.engine``."""
from __future__ import annotations

import os
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
    def forward(ctx, anchor, value, closure, prescale=1.0):
        ctx.closure = closure
        ctx.prescale = float(prescale)
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        # data-parallel mean: the backward is linear in grad_out, so the 1/world_size of DDP's gradient averaging is applied
        # here (exact for power-of-two world sizes) and the collectives are plain SUMs - no pre-multiplied reduction op
        ctx.closure(grad_out * ctx.prescale if ctx.prescale != 1.0 else grad_out)
        return None, None, None, None


def _scalar(t):
    return t.reshape(())


def compute_pgd(pl_module, batch, loss_name, k_modality=None):
    """objectives.py:160-188 for loss_name == "moco"."""
    img_delta = pl_module.pgd_attacker.pgd_attack(pl_module, batch, k_modality=k_modality)
    batch["image"][0] = batch["image"][0] + img_delta                     # :176 (on top of img + delta_{K-1})
    phase = "train" if pl_module.training else "val"
    pl_module.log(f"{loss_name}_attack/{phase}/delta", torch.linalg.norm(img_delta, dim=1).mean())
    return batch


def compute_geometric(pl_module, batch, loss_name, k_modality=None):
    """objectives.py:190-215."""
    attack_words = pl_module.greedy_attacker.adv_attack_samples(pl_module, batch, k_modality)
    batch["text"] = attack_words["text"]
    batch["text_ids"] = attack_words["txt_input_ids"]
    batch["text_masks"] = attack_words["text_masks"]
    phase = "train" if pl_module.training else "val"
    pl_module.log(f"{loss_name}_attack/{phase}/num_changes", attack_words["num_changes"])
    pl_module.log(f"{loss_name}_attack/{phase}/change_rate", attack_words["change_rate"])
    return batch


def _only_moco(pl_module):
    ln = pl_module.hparams.config["loss_names"]
    return ln.get("moco", 0) > 0 and sum(1 for v in ln.values() if v > 0) == 1


def _attacked_view(pl_module, pv, op, k, suffix, success_name, prediction_original, ret, phase, before_loss=None):
    """One loss view (objectives.py:287-317 / :324-354 / :362-392; suffix "clean": the clean query of :267-275 as a
    loss, BASELINE configs[1]): encoder forward on `op` with the text in `pv`, InfoNCE against the queue, metrics, and a
    loss tensor whose backward runs the HIP backward."""
    eng = pl_module.engine
    B = pv.B
    need_grad = torch.is_grad_enabled() and pl_module.training
    eng.encoder_forward(pv, key=False, mode=L.MODE_FULL if need_grad else L.MODE_INFER, patchesT=op, cls_tail=True)
    eng.heads_forward(pv, key=False)
    if before_loss is not None:
        before_loss()                                  # e.g. join the key-encoder stream: k is needed from here on
    if pv.k.data_ptr() != k.data_ptr():
        pv.k.copy_(k)
    eng.infonce(pv, 1.0 / B, want_dq=need_grad)
    rows = pv.rows
    if phase == "train" and success_name is not None:
        pl_module.log(f"moco_attack/{success_name}", (rows[:, 1] != prediction_original).float().mean())
    means = rows.mean(dim=0)                           # one reduction launch for the six batch means (the values are views of it)
    for j, name in ((3, "pos_dist"), (4, "pos_cosine"), (5, "pos_dot"), (6, "neg_dist"), (7, "neg_cosine"), (8, "neg_dot")):
        ret[f"{name}_attacked_{suffix}"] = means[j]
    ret[f"q_{suffix}_attack"] = pv.q.clone()
    value = _scalar(pv.loss_sum.clone())
    if not need_grad:
        return value
    dq_saved = pv.dq.clone()

    def backward(grad_out, pv=pv, dq_saved=dq_saved, op=op):
        dq = dq_saved * grad_out.to(dq_saved.dtype)
        eng.heads_backward(pv, dq, None, with_grads=True)
        eng.encoder_backward(pv, L.MODE_FULL, op, pv.dcls, cls_only=True, dpatches=None)
        # one attacked view = one backward per step: its gradient all-reduces can start layer by layer right away
        # reduced over ranks once per step, after the step's last closure; a single closure (the image-view step) gets
        # per-layer all-reduces that start while the layers below are still in their backward
        pl_module.after_backward(overlap=True)

    return _DeferredBackward.apply(pl_module.grad_anchor, value, backward, pl_module.grad_prescale())


def compute_itm_wpa(pl_module, batch):
    """objectives.compute_itm_wpa (:714-787): ITM head + CE, and the word-patch-alignment OT distance
    (cosine cost :24-34, IPOT :46-76 with beta=0.5 / 50 iterations, trace :37-43)."""
    import ctypes as C
    from ..._lib import F
    eng = pl_module.engine
    dev = eng.device
    Bn = len(batch["text"])
    pos_len = Bn // 2
    itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(Bn - pos_len)]).to(dev)
    forced = getattr(pl_module, "itm_labels_override", None)            # test hook: fix the 50/50 draw
    itm_labels = forced.to(dev).float() if forced is not None else itm_labels[torch.randperm(Bn, device=dev)]
    img, fimg = batch["image"][0].to(dev), batch["false_image_0"][0].to(dev)
    if hasattr(img, "tables"):                                          # decoded bytes (collate_raw_uint8): MinMaxResize on the device first
        img, fimg = eng.resize_raw(img), eng.resize_raw(fimg)
    if hasattr(img, "float_image"):                                     # byte batches (collate_uint8): the mix of the two views needs pixels
        img, fimg = img.float_image(), fimg.float_image()
    images = torch.where(itm_labels.view(-1, 1, 1, 1) == 1, img, fimg)                 # :722-730

    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], images, tag="itm")
    d = pb.d
    Lt, Li, D = d.L, d.P + 1, d.D
    N = Lt + Li
    need_grad = torch.is_grad_enabled() and pl_module.training
    op = eng.make_operand(pb, out=pb.patchesT_full)
    eng.encoder_forward(pb, key=False, mode=L.MODE_FULL if need_grad else L.MODE_INFER, patchesT=op)
    eng.heads_forward(pb, key=False, want_q=False)
    st = stream_ptr()
    lay = eng.layout

    labels_i = itm_labels.to(torch.int32)
    logits = torch.empty(Bn, 2, device=dev)
    dlogits = torch.empty(Bn, 2, device=dev)
    losses = torch.zeros(2, device=dev)
    w_itm, b_itm = eng.q32[lay.itm_w:lay.itm_w + 2 * D], eng.q32[lay.itm_b:lay.itm_b + 2]
    check(lib.rmcl_itm_fwd(P(pb.cls), P(w_itm), P(b_itm), P(labels_i), P(logits), P(dlogits), P(losses), Bn, D, F(1.0 / Bn), st), "itm_fwd")

    # masks (:739-745): drop [CLS], the last valid text token and the image cls slot
    txt_mask = pb.text_mask.bool().clone()
    lens = txt_mask.sum(dim=1)
    txt_mask[torch.arange(Bn, device=dev), lens - 1] = False
    txt_mask[:, 0] = False
    img_mask = pb.co_mask[:, Lt:].bool().clone()
    img_mask[:, 0] = False
    txt_valid, img_valid = txt_mask.to(torch.int32).contiguous(), img_mask.to(torch.int32).contiguous()

    ld = (Li + 3) // 4 * 4
    M = Bn * N
    xhat, norms = torch.empty(M, D, device=dev), torch.empty(M, device=dev)
    check(lib.rmcl_l2norm_rows_fwd(P(pb.xn), P(xhat), P(norms), M, D, F(1e-5), st), "l2norm")
    cost = torch.zeros(Bn, Lt, ld, device=dev)
    img_rows = xhat[Lt:]                                                   # first image row of sample 0
    check(lib.rmcl_gemm_batched(P(xhat), P(img_rows), P(cost), Lt, Li, D, I64(D), I64(D), ld, F(1.0), Bn, I64(N * D), I64(N * D),
                                I64(Lt * ld), L.F32, L.F32, 1, 1, st), "cosine sim")
    check(lib.rmcl_wpa_cost_finish(P(cost), P(txt_valid), P(img_valid), Bn, Lt, Li, ld, st), "cost_finish")
    T = torch.empty(Bn, Li, Lt, device=dev)
    check(lib.rmcl_ipot_f32(P(cost), P(txt_valid), P(img_valid), P(T), Bn, Lt, Li, ld, F(0.5), 50, st), "ipot")
    w = (itm_labels * 2 - 1) * (0.1 / Bn)                                  # 0.1 * (sum_pos - sum_neg) / B  (:763-765,771)
    dist = torch.empty(Bn, device=dev)
    dsim = torch.empty(Bn, Lt, ld, device=dev) if need_grad else None
    check(lib.rmcl_wpa_distance(P(cost), P(T), P(w), P(dist), P(dsim), Bn, Lt, Li, ld, st), "wpa_distance")
    losses[1] = (dist * w).sum()
    value = losses.clone()

    if need_grad:
        def backward(grad_out, pb=pb, dlogits=dlogits, dsim=dsim, xhat=xhat, norms=norms, op=op):
            st2 = stream_ptr()
            g = grad_out.to(torch.float32)
            dl = dlogits * g[0]
            dcls_itm = torch.empty(Bn, D, device=dev)
            gw, gb = eng.g32[lay.itm_w:lay.itm_w + 2 * D], eng.g32[lay.itm_b:lay.itm_b + 2]
            check(lib.rmcl_itm_bwd(P(dl), P(pb.cls), P(w_itm), P(dcls_itm), P(gw), P(gb), Bn, D, F(1.0), st2), "itm_bwd")
            eng.heads_backward(pb, None, dcls_itm, with_grads=True)
            ds = dsim * g[1]
            dxhat = torch.zeros(M, D, device=dev)
            check(lib.rmcl_gemm_batched(P(ds), P(xhat[Lt:]), P(dxhat), Lt, D, Li, I64(ld), I64(D), D, F(1.0), Bn, I64(Lt * ld),
                                        I64(N * D), I64(N * D), L.F32, L.F32, 1, 0, st2), "d txt")
            check(lib.rmcl_gemm_batched(P(ds), P(xhat), P(dxhat[Lt:]), Li, D, Lt, I64(ld), I64(D), D, F(1.0), Bn, I64(Lt * ld),
                                        I64(N * D), I64(N * D), L.F32, L.F32, 0, 0, st2), "d img")
            dxn = torch.empty(M, D, device=dev)
            check(lib.rmcl_l2norm_rows_bwd(P(dxhat), P(xhat), P(norms), P(dxn), M, D, st2), "l2norm_bwd")
            dxn.view(Bn, N, D)[:, 0] += pb.dcls
            eng.encoder_backward(pb, L.MODE_FULL, op, dxn, cls_only=False, dpatches=None)
            pl_module.after_backward()

        value = _DeferredBackward.apply(pl_module.grad_anchor, value, backward, pl_module.grad_prescale())
    ret = {"itm_loss": value[0], "itm_wpa_loss": value[1], "itm_logits": logits, "itm_labels": itm_labels}
    phase = "train" if pl_module.training else "val"
    pl_module.log(f"itm/{phase}/loss", ret["itm_loss"].detach())
    pl_module.log(f"itm/{phase}/wpa_loss", ret["itm_wpa_loss"].detach())
    pl_module.log(f"itm/{phase}/accuracy", (logits.argmax(-1) == itm_labels.long()).float().mean())
    return ret


def compute_moco_contrastive(pl_module, batch):
    """objectives.py:217-447 (image view).  Returns {"moco_loss", pos_/neg_{dist,cosine,dot}_attacked_img}."""
    eng = pl_module.engine
    clean_view = bool(pl_module.hparams.config.get("clean_view", False))
    if not (pl_module.image_view or pl_module.text_view or clean_view):
        raise ZeroDivisionError("division by zero: loss / loss_num with both views off (objectives.py:250-251,397)")
    if pl_module.augmentation:
        raise NotImplementedError("augmentation views are out of scope (SURVEY 2.1 #17)")
    phase = "train" if pl_module.training else "val"
    ret = {}

    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
    B = pb.B

    op = eng.make_operand(pb)
    # The momentum-encoder forward (infer_k under no_grad, :262-265) and the clean query forward (:267-275)
    # are independent: they run on two HIP streams so each fills the other's tile-quantisation tails.
    pk = eng.twin(pb, "key")
    pk.text_ids, pk.text_mask = pb.text_ids, pb.text_mask
    pk.k = pb.k                            # the key head writes the keys where the query passes read them (no copy launch after the join)
    main = torch.cuda.current_stream()
    side = eng.side_stream
    if eng.lp_stale:                       # (after a checkpoint load / manual weight edit) refresh the bf16 shadows on the MAIN
        eng.refresh_shadows()              # stream before forking: both streams read them
    # momentum update (:257-260).  (Moving the 1.6 GB sweep onto the key encoder's stream, beside the query encoder's first
    # forward, measured no gain: 37.5 vs 37.6 ms per step - both are HBM-bound there.)
    eng.ema(pl_module.momentum)
    # PGD step 0 runs the query encoder on img + delta_0 = img: with dropout off that IS the clean query forward
    # (:267-275), so it is computed once (common sub-expression) and its logits give prediction_original.
    # Under dropout the reference draws a fresh mask for the clean forward (:267) and another for PGD step 0's (pgd_attack_vilt.py:145), so
    # the two are different computations and both run.  config["share_clean_forward"] = True (default False) lets PGD step 0's forward stand in
    # for the clean one under dropout too: loss, gradients and the perturbation are untouched (the clean logits never enter the loss,
    # SURVEY quirk 3) - only the LOGGED prediction_original / q_original then come from the forward that shares step 0's mask.
    share = bool(pl_module.hparams.config.get("share_clean_forward", False))
    fuse_clean = pl_module.image_view and not pl_module.text_view and (not eng.dropout_on or share) and not clean_view
    # two half-batch chains (Engine.lanes) where the PGD loop is the step's front: each lane runs its own K iterations, lane 0 on this
    # stream, lane 1 on the side stream.  With dropout ON the clean query forward cannot be shared with PGD step 0 (the reference draws
    # a fresh mask for each, objectives.py:267 vs pgd_attack_vilt.py:145): it then follows the key forward on the key stream, in its own
    # buffers, beside the lanes
    use_lanes = (pl_module.image_view and not pl_module.text_view and not clean_view and eng.pgd_bufs(pb) is pb and eng.lanes(pb) is not None)
    # with the lanes on this stream and the side stream, the key forward takes the (idle) weight-gradient stream: three chains
    key_stream = eng.dw_stream if (use_lanes and (os.environ.get("RMCL_KEY_LANES", "0") != "1" or not fuse_clean)) else side
    key_lanes = eng.lanes(pk) if (use_lanes and key_stream is side) else None
    k_ready = None                          # event: the keys are in pb.k (None: wait for the whole key stream)
    if key_lanes is not None:
        eng.fold_of(True)                                   # (refreshed on the main stream before the fork)
        side.wait_stream(main)
        check(lib.rmcl_tune_set(10, len(key_lanes)), "tune_set")
        try:
            per = key_lanes[0].B * pb.d.P
            for i, kl in enumerate(key_lanes):
                with torch.cuda.stream(side if i else main):
                    eng.encoder_forward(kl, key=True, mode=L.MODE_INFER, patchesT=op[i * per:(i + 1) * per], cls_tail=True)
                    eng.heads_forward(kl, key=True, wgrad=False)
        finally:
            check(lib.rmcl_tune_set(10, 1), "tune_set")      # (process-global routing state: restored on every path)
    else:
        if use_lanes and not fuse_clean:
            eng.fold_of(False)                              # the clean query pass will read the query fold on the key stream: refresh it before the fork
        key_stream.wait_stream(main)
        with torch.cuda.stream(key_stream):
            eng.encoder_forward(pk, key=True, mode=L.MODE_INFER, patchesT=op, cls_tail=True)
            eng.heads_forward(pk, key=True, wgrad=False)
            k_ready = torch.cuda.Event()
            k_ready.record(key_stream)
    gather_box = {}

    def join_key_stream():
        if k_ready is not None:
            main.wait_event(k_ready)
        else:
            main.wait_stream(key_stream)
        # asynchronous key all-gather (RCCL's own stream): overlaps everything until the enqueue (one rank: the keys themselves)
        gather_box["g"] = dist_utils.KeyGather(pb.k.clone() if dist_utils.world_size() > 1 else pb.k) if pl_module.training else None

    k = pb.k
    clean = {}
    clean_on_key = False
    loss = 0
    loss_num = 0
    if clean_view:
        # build extension (config key "clean_view", default off): CE on the clean logits of :267-275 as a loss term -
        # BASELINE configs[1] "clean ITM + contrastive" (SURVEY 8d Config 2).  The reference forms these logits in
        # every step but never turns them into a loss (quirk 3).
        pc = eng.bind_text(pb, pb.text_ids, pb.text_mask, tag="moco_clean")
        op_c = eng.make_operand(pb, out=pc.patchesT_full)
        loss_c = _attacked_view(pl_module, pc, op_c, k, "clean", None, None, ret, phase, before_loss=join_key_stream)
        clean = {"prediction": pc.rows[:, 1].clone(), "q": pc.q.clone()}
        pl_module.log("moco_loss/clean_loss", loss_c.detach())
        loss = loss + loss_c
        loss_num += 1
    elif not fuse_clean and use_lanes and k_ready is not None:
        # clean query (:267-275) behind the key forward on the key stream, in buffers of its own (the lanes' passes use pb's)
        pc = eng.twin(pb, "clean_q")
        pc.text_ids, pc.text_mask, pc.k = pb.text_ids, pb.text_mask, pb.k
        clean = {"prediction": torch.empty(B, dtype=torch.float32, device=eng.device), "q": torch.empty_like(pb.q)}
        with torch.cuda.stream(key_stream):
            eng.encoder_forward(pc, key=False, mode=L.MODE_INFER, patchesT=op, cls_tail=True)
            eng.heads_forward(pc, key=False, wgrad=False)
            eng.infonce(pc, 0.0, want_dq=False, metrics=False)
            clean["prediction"].copy_(pc.rows[:, 1])
            clean["q"].copy_(pc.q)
        clean_on_key = True
    elif not fuse_clean:
        eng.encoder_forward(pb, key=False, mode=L.MODE_INFER, patchesT=op, cls_tail=True)     # clean query
        eng.heads_forward(pb, key=False)
        join_key_stream()
        eng.infonce(pb, 0.0, want_dq=False, metrics=False)
        clean = {"prediction": pb.rows[:, 1].clone(), "q": pb.q.clone()}
    prediction_original = clean.get("prediction")

    txt_ids = txt_masks = None
    if pl_module.text_view:                                                 # :277-317
        aug = compute_geometric(pl_module, copy(batch), "moco", k_modality=k)
        txt_ids, txt_masks = aug["text_ids"], aug["text_masks"]
        pt = eng.bind_text(pb, txt_ids, txt_masks, tag="moco_txt")
        op_t = eng.make_operand(pb, out=pt.patchesT_full)                   # clean image, attacked text
        loss_t = _attacked_view(pl_module, pt, op_t, k, "txt", "Geom_success_rate", prediction_original, ret, phase)
        pl_module.log("moco_loss/attacked_txt_loss", loss_t.detach())
        loss = loss + loss_t
        loss_num += 1
    if pl_module.image_view:                                                # :319-354
        if key_lanes is not None:
            join_key_stream()                               # lane 0 waits for lane 1's key forward only; every lane reads its own keys
            pl_module.pgd_attacker.attack_patches(pl_module, pb, None, clean_out=clean, clean_op=op)
            prediction_original = clean["prediction"]
        elif fuse_clean:
            pl_module.pgd_attacker.attack_patches(pl_module, pb, None, before_first_loss=join_key_stream, clean_out=clean, clean_op=op,
                                                  key_stream=key_stream, key_event=k_ready)
            prediction_original = clean["prediction"]
        elif clean_on_key:
            pl_module.pgd_attacker.attack_patches(pl_module, pb, None, before_first_loss=join_key_stream, clean_op=op, key_stream=key_stream,
                                                  key_event=k_ready)
            main.wait_stream(key_stream)                    # the clean query's prediction / q (key stream) are read from here on
        else:
            pl_module.pgd_attacker.attack_patches(pl_module, pb, k, clean_op=op)        # compute_pgd (:319-323)
        check(lib.rmcl_delta_channel_norm(P(pb.delta), P(eng.zero_scalar(pb)), I64(pb.delta.shape[0]), 3,
                                          pb.d.patch_k // 3, stream_ptr()), "delta_norm")
        # mean over ALL pixels of the (padded) batch image like torch.linalg.norm(delta, dim=1).mean() (:184); the pad pixels of
        # a zero-padded batch carry delta = 0 and are not stored in the patch layout
        n_pix = pb.delta.numel() // 3 if pb.geom is None else pb.B * pb.geom.shape[2] * pb.geom.shape[3]
        pl_module.log(f"moco_attack/{phase}/delta", _scalar(pb.loss_sum / float(n_pix)))
        # attacked view = img + delta_{K-1} + delta_K  (pgd_attack_vilt.py:144 + objectives.py:176): written by the last PGD update
        op_att = pb.patchesT_full
        loss_i = _attacked_view(pl_module, pb, op_att, k, "img", "PGD_success_rate", prediction_original, ret, phase)
        ret["logit_pos_img_attack"] = pb.rows[:, 2].clone()
        ret["lse_img_attack"] = pb.rows[:, 9].clone()
        pl_module.log("moco_loss/attacked_img_loss", loss_i.detach())
        loss = loss + loss_i
        loss_num += 1
    if pl_module.image_view and pl_module.text_view:                        # :356-392 attacked image AND attacked text
        pbo = eng.bind_text(pb, txt_ids, txt_masks, tag="moco_both")
        op_b = pbo.patchesT_full.copy_(pb.patchesT_full)                    # the same attacked image (own buffer: read again by the backward)
        loss_b = _attacked_view(pl_module, pbo, op_b, k, "both", "Both_success_rate", prediction_original, ret, phase)
        pl_module.log("moco_loss/attacked_both_loss", loss_b.detach())
        loss = loss + loss_b
        loss_num += 1

    if pl_module.training:                                                  # _dequeue_and_enqueue (:394-395)
        keys_all = gather_box["g"].wait()
        do, new_ptr = dist_utils.queue_advance(pl_module.queue_ptr, keys_all.shape[0], pl_module.num_negative,
                                               pl_module.per_step_bs)
        if do:
            eng.enqueue(keys_all, pl_module.queue_ptr)
            pl_module.queue_ptr = new_ptr

    ret["moco_loss"] = loss / loss_num
    ret["k"] = k.clone()
    ret["q_original"] = clean["q"]
    pl_module.log(f"moco_loss/step/{phase}", ret["moco_loss"].detach())
    views = (["img"] if pl_module.image_view else []) + (["txt"] if pl_module.text_view else []) + \
            (["both"] if pl_module.image_view and pl_module.text_view else [])
    for v in views:                                                         # :402-445
        for kind, tag in (("dist", "L2"), ("cosine", "Cosine"), ("dot", "Dot")):
            pos, neg = ret[f"pos_{kind}_attacked_{v}"], ret[f"neg_{kind}_attacked_{v}"]
            pl_module.log(f"moco_dist_{phase}_{tag}/Pos_attacked_{v}", pos)
            pl_module.log(f"moco_dist_{phase}_{tag}/Neg_attacked_{v}", neg)
            pl_module.log(f"moco_dist_{phase}_{tag}/Neg-Pos_attacked_{v}", neg - pos)
    return ret


def _bt_view(pl_module, pv, op, zk, suffix, ret, phase, need_grad, training):
    """One Barlow-Twins loss view (objectives.py:464-498 text / :500-525 image / :527-546 both): forward of the view, the
    head (running estimates updated in training), c = q^T k / per_step_bs summed over ranks, on/off-diagonal loss, distance
    logs; returns (loss value with a deferred HIP backward, on_diag, adv_lr * off_diag)."""
    eng = pl_module.engine
    eng.encoder_forward(pv, key=False, mode=L.MODE_FULL if need_grad else L.MODE_INFER, patchesT=op, cls_tail=True)
    eng.heads_forward(pv, key=False, want_q=False)
    bq = eng.bt_bufs(pv.B, "q_" + suffix)
    eng.bt_forward(bq, pv.cls, training, track=training)

    def reduce_c(c):                                                            # torch.distributed.all_reduce(c) (:480,:507,:535)
        if dist_utils.world_size() > 1:
            torch.distributed.all_reduce(c)

    loss2 = eng.bt_loss(bq, zk, float(pl_module.per_step_bs), pl_module.adv_lr, 1.0, want_dz=need_grad, reduce_c=reduce_c)
    on_diag, red = loss2[0].clone(), pl_module.adv_lr * loss2[1]
    rows = eng.bt_pair_metrics(bq, zk)
    ret[f"pos_dist_attacked_{suffix}"], ret[f"pos_cosine_attacked_{suffix}"], ret[f"pos_dot_attacked_{suffix}"] = \
        rows[:, 0].mean(), rows[:, 1].mean(), rows[:, 2].mean()
    ret[f"q_{suffix}"] = bq.z.clone()
    value = _scalar(on_diag + red)
    if need_grad:
        dz_saved = bq.dz.clone()

        def backward(grad_out, pv=pv, bq=bq, dz_saved=dz_saved, op=op):
            dz = dz_saved * grad_out.to(dz_saved.dtype)
            dcls = eng.bt_backward(bq, dz, training=True, with_grads=True)
            eng.heads_backward(pv, None, dcls, with_grads=True)
            eng.encoder_backward(pv, L.MODE_FULL, op, pv.dcls, cls_only=True, dpatches=None)
            pl_module.after_backward(overlap=True)

        # NOTE on ranks: the reference all-reduces c WITHOUT autograd support, so each rank backpropagates d loss(c_global) / dq
        # of its own rows and DDP then AVERAGES the gradients - the 1/world_size of that average is the prescale
        value = _DeferredBackward.apply(pl_module.grad_anchor, value, backward, pl_module.grad_prescale())
        # Reference behaviour: training_step sums every returned value whose key contains "loss" (vilt_module.py:475), and the
        # logged components ARE live graph tensors there (:486-487) - each view is optimised with weight 1/loss_num + 1.  The
        # invariance component therefore carries the same deferred backward (as a zero-valued term).
        on_diag = on_diag + (value - value.detach())
    return value, on_diag, red


def compute_barlowtwins_contrastive(pl_module, batch):
    """objectives.py:449-602: ONE encoder and one head.  k = head(infer(clean)) under no_grad; per view
    q = head(infer(view)), c = q^T k / per_step_bs summed over ranks, loss = sum_i (c_ii - 1)^2 + adv_lr * sum_{i != j} c_ij^2
    (the reference uses its `adv_lr` hyper-parameter as the redundancy weight).  Views like the MoCo objective: attacked text
    (greedy attack on this loss), attacked image (PGD on this loss; img + delta_{K-1} + delta_K), both.  The head's BatchNorms
    run in the module's mode and - in training - update their running estimates in EVERY call, also under no_grad."""
    eng = pl_module.engine
    if pl_module.augmentation:
        raise NotImplementedError("augmentation views are out of scope (SURVEY 2.1 #17)")
    if not (pl_module.image_view or pl_module.text_view):
        raise ZeroDivisionError("division by zero: loss / loss_num with both views off (objectives.py:451-452,548)")
    phase = "train" if pl_module.training else "val"
    training = bool(pl_module.training)
    need_grad = torch.is_grad_enabled() and training
    ret = {}
    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0], tag="bt")
    B = pb.B
    op = eng.make_operand(pb)
    bk = eng.bt_bufs(B, "k")
    eng.encoder_forward(pb, key=False, mode=L.MODE_INFER, patchesT=op, cls_tail=True)          # :460-462
    eng.heads_forward(pb, key=False, want_q=False)
    zk = eng.bt_forward(bk, pb.cls, training, track=training)
    loss, loss_num = 0, 0
    views = []
    txt_ids = txt_masks = None
    if pl_module.text_view:                                                     # :464-498
        aug = compute_geometric(pl_module, copy(batch), "barlowtwins", k_modality=zk)
        txt_ids, txt_masks = aug["text_ids"], aug["text_masks"]
        pt = eng.bind_text(pb, txt_ids, txt_masks, tag="bt_txt")
        op_t = eng.make_operand(pb, out=pt.patchesT_full)
        v, on_diag, red = _bt_view(pl_module, pt, op_t, zk, "txt", ret, phase, need_grad, training)
        ret["barlowtwins_loss_invariance_text"], ret["barlowtwins_loss_redundancy_text"] = on_diag, red
        loss, loss_num = loss + v, loss_num + 1
        views.append(("txt", "text"))
    if pl_module.image_view:                                                    # :500-525
        pl_module.pgd_attacker.attack_patches(pl_module, pb, zk, clean_op=op)   # compute_pgd (:503)
        check(lib.rmcl_delta_channel_norm(P(pb.delta), P(eng.zero_scalar(pb)), I64(pb.delta.shape[0]), 3, pb.d.patch_k // 3,
                                          stream_ptr()), "delta_norm")
        n_pix = pb.delta.numel() // 3 if pb.geom is None else pb.B * pb.geom.shape[2] * pb.geom.shape[3]
        pl_module.log(f"barlowtwins_attack/{phase}/delta", _scalar(pb.loss_sum / float(n_pix)))
        op_att = pb.patchesT_full                                               # written by the last PGD update
        v, on_diag, red = _bt_view(pl_module, pb, op_att, zk, "img", ret, phase, need_grad, training)
        ret["barlowtwins_loss_invariance_img"], ret["barlowtwins_loss_redundancy_img"] = on_diag, red
        loss, loss_num = loss + v, loss_num + 1
        views.append(("img", "img"))
    if pl_module.image_view and pl_module.text_view:                            # :527-546
        pbo = eng.bind_text(pb, txt_ids, txt_masks, tag="bt_both")
        op_b = pbo.patchesT_full.copy_(pb.patchesT_full)
        v, on_diag, red = _bt_view(pl_module, pbo, op_b, zk, "both", ret, phase, need_grad, training)
        ret["barlowtwins_loss_invariance_both"], ret["barlowtwins_loss_redundancy_both"] = on_diag, red
        loss, loss_num = loss + v, loss_num + 1
        views.append(("both", "both"))
    ret["k"] = zk.clone()
    ret["barlowtwins_loss"] = loss / loss_num                                   # :548
    pl_module.log(f"barlowtwins/{phase}/loss", ret["barlowtwins_loss"].detach())
    for suffix, name in views:                                                  # :555-600
        pl_module.log(f"barlowtwins_dist_{phase}_L2/Pos_attacked_{suffix}", ret[f"pos_dist_attacked_{suffix}"])
        pl_module.log(f"barlowtwins_dist_{phase}_Cosine/Pos_attacked_{suffix}", ret[f"pos_cosine_attacked_{suffix}"])
        pl_module.log(f"barlowtwins_dist_{phase}_Dot/Pos_attacked_{suffix}", ret[f"pos_dot_attacked_{suffix}"])
        pl_module.log(f"barlowtwins/{phase}/barlowtwins_loss_invariance_{name}", ret[f"barlowtwins_loss_invariance_{name}"].detach())
        pl_module.log(f"barlowtwins/{phase}/barlowtwins_loss_redundancy_{name}", ret[f"barlowtwins_loss_redundancy_{name}"].detach())
    return ret
