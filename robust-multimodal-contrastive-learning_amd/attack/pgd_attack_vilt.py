"""PGD image attack on the MoCo objective.  Mirrors attack/pgd_attack_vilt.py:7-175 of the
reference (class names, constructor keys, ``pgd_attack(pl_module, batch, k_modality)`` -> delta),
but runs K x (encoder forward, InfoNCE, data-gradient backward, L-inf-normalised ascent step,
eps-projection) as HIP kernels with no deepcopy of the encoder and no weight-gradient work.
``PGDAttack_bartlowtwins`` (sic, :178-236) is the same loop on the Barlow-Twins cross-correlation loss."""
from __future__ import annotations

import torch

from .. import _lib as L


class PGDAttack:
    def __init__(self, config, contrastive_framework):
        self.contrastive_framework = contrastive_framework
        self.adv_steps_img = config["adv_steps_img"]
        self.adv_lr_img = config["adv_lr_img"]
        self.adv_max_norm_img = config["adv_max_norm_img"]
        self.max_image_len = config["max_image_len"]

    def pgd_attack(self, pl_module, batch, k_image):
        raise NotImplementedError(f"pgd_attack of {self.contrastive_framework} isn't implemented.")


class PGDAttack_moco(PGDAttack):
    def __init__(self, config):
        super().__init__(config, "moco")

    def attack_patches(self, pl_module, pb, k, before_first_loss=None, clean_out=None, keep_prev=False, clean_op=None, key_stream=None, key_event=None):
        """K-step attack in patch layout.  Leaves delta_K in ``pb.delta`` and the ATTACKED VIEW's operand
        cast(img + delta_{K-1} + delta_K) (see compute_pgd / objectives.py:176) in ``pb.patchesT_full``; with ``keep_prev``
        also delta_{K-1} in ``pb.delta_prev`` (the public ``pgd_attack`` needs it for the batch image it leaves behind).

        Every step's update kernel also writes the next forward's operand (``Engine.pgd_step``), step 0 starts from the
        implicit delta_0 = 0: no zero fills, no separate add + cast passes, no delta copy inside the loop.

        ``before_first_loss``: callback run once between the first encoder forward and the first InfoNCE (the
        caller joins the key-encoder stream there).  ``clean_out``: dict that receives the clean-query statistics:
        step 0 evaluates the query encoder at img + delta_0 = img, i.e. it IS the clean forward of
        objectives.py:267-275, so that forward is not computed twice when dropout is off."""
        eng = pl_module.engine
        K = self.adv_steps_img
        if k is not None and k.data_ptr() != pb.k.data_ptr():
            pb.k.copy_(k)
        pb0 = pb
        pb = eng.pgd_bufs(pb)                                 # fp32 twin when the engine runs PGD in fp32 (:141)
        lanes = eng.lanes(pb) if pb is pb0 else None
        if lanes is not None:
            return self._attack_lanes(eng, pb, lanes, K, before_first_loss, clean_out, keep_prev, clean_op, key_stream, key_event)
        # img_init + delta_0, delta_0 = 0 (:136,144); ``clean_op``: the caller's cast of the clean image in pb.patchesT, if it has one
        op = clean_op if (clean_op is not None and pb is pb0) else eng.make_operand(pb)
        for step in range(K):
            last = step == K - 1
            eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op, cls_tail=True)
            eng.heads_forward(pb, key=False, wgrad=False)
            if step == 0 and before_first_loss is not None:
                before_first_loss()
            # CE(label 0) / K, mean over the batch (:152-158); gradient wrt q only
            eng.infonce(pb, grad_scale=1.0 / (pb.B * K), want_dq=True, metrics=False)     # (only dq and the prediction are read)
            if step == 0 and clean_out is not None:
                clean_out["prediction"] = pb.rows[:, 1].clone()
                clean_out["q"] = pb.q.clone()
            eng.heads_backward(pb, pb.dq, None, with_grads=False)
            eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=pb.gpatch)
            if last and keep_prev:
                if K > 1:
                    pb.delta_prev.copy_(pb.delta)
                else:
                    pb.delta_prev.zero_()
            op = pb0.patchesT_full if last else pb.patchesT
            eng.pgd_step(pb, self.adv_lr_img, self.adv_max_norm_img, first=step == 0, out=op, sum_prev=last)   # :162-173
        return pb.delta

    def _attack_lanes(self, eng, pb, lanes, K, before_first_loss, clean_out, keep_prev, clean_op, key_stream, key_event=None):
        """The same K steps as two independent half-batch chains (Engine.lanes): lane 0 on the current stream, lane 1 on
        ``eng.side_stream`` (RMCL_LANE_LAG_US: an extra start delay of lane 1; the host's call-by-call enqueue order already staggers them).
        Every per-sample buffer of a lane is a view of ``pb``'s, so ``pb`` ends up exactly as the one-chain loop leaves it.
        ``key_stream`` / ``key_event``: where the keys are produced, if not on the current stream (lane 1 waits for the event - or the
        whole stream - by itself before its first InfoNCE)."""
        import os
        from .._lib import lib, check
        main = torch.cuda.current_stream()
        # lane 1 runs on the engine's side stream, BEHIND the key-encoder forward the caller may have put there: HIP multiplexes its
        # streams onto four hardware queues, and a fifth stream of this process shared the main stream's queue - the lanes then ran
        # one after the other (measured: no change of the step at all)
        side = eng.side_stream
        streams = [main, side, eng.comm_stream, eng.dw_stream][:len(lanes)]     # (four lanes: experiment, RMCL_LANE_COUNT=4)
        op_full = clean_op if clean_op is not None else eng.make_operand(pb)
        eng.fold_of(False)                                    # parameter-derived operands are refreshed on the main stream BEFORE the fork
        eng.weights_T()
        if clean_out is not None:
            clean_out["prediction"] = torch.empty(pb.B, dtype=torch.float32, device=eng.device)
            clean_out["q"] = torch.empty_like(pb.q)
        for st in streams[1:]:                                # every lane stream forks AFTER the operand / fold / transposed-weight refresh
            st.wait_stream(main)
        lag = float(os.environ.get("RMCL_LANE_LAG_US", "0"))
        if lag > 0:
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(lag * 2400))            # (device spin clock: 2.4 cycles per ns on MI355X)
        per = lanes[0].B * pb.d.P
        ops = [op_full[i * per:(i + 1) * per] for i in range(len(lanes))]
        def on(i):
            return torch.cuda.stream(streams[i])

        check(lib.rmcl_tune_set(10, len(lanes)), "tune_set")  # the GEMM routing sizes a launch against its share of the CUs
        try:
            for step in range(K):
                last = step == K - 1
                # the host alternates between the lanes call by call, so that neither queue runs dry while the other is being filled
                for i, ln in enumerate(lanes):
                    with on(i):
                        eng.encoder_forward(ln, key=False, mode=L.MODE_DATA, patchesT=ops[i], cls_tail=True)
                        eng.heads_forward(ln, key=False, wgrad=False)
                for i, ln in enumerate(lanes):
                    with on(i):
                        if step == 0:
                            if i == 0 and before_first_loss is not None:
                                before_first_loss()
                            if i > 0 and key_event is not None:
                                streams[i].wait_event(key_event)
                            elif i > 0 and key_stream is not None and key_stream is not streams[i]:
                                streams[i].wait_stream(key_stream)
                        eng.infonce(ln, grad_scale=1.0 / (pb.B * K), want_dq=True, metrics=False)       # 1 / B of the WHOLE batch (:152-158)
                        if step == 0 and clean_out is not None:
                            clean_out["prediction"][i * ln.B:(i + 1) * ln.B].copy_(ln.rows[:, 1])
                            clean_out["q"][i * ln.B:(i + 1) * ln.B].copy_(ln.q)
                        eng.heads_backward(ln, ln.dq, None, with_grads=False)
                for i, ln in enumerate(lanes):
                    with on(i):
                        eng.encoder_backward(ln, L.MODE_DATA, ops[i], ln.dcls, cls_only=True, dpatches=ln.gpatch)
                for i, ln in enumerate(lanes):
                    with on(i):
                        if last and keep_prev:
                            if K > 1:
                                ln.delta_prev.copy_(ln.delta)
                            else:
                                ln.delta_prev.zero_()
                        ops[i] = ln.patchesT_full if last else ln.patchesT
                        eng.pgd_step(ln, self.adv_lr_img, self.adv_max_norm_img, first=step == 0, out=ops[i], sum_prev=last)
        finally:
            # process-global routing state and the forked streams are put back on EVERY path: an exception inside the loop must not
            # leave every later GEMM sized for half the chip or the side streams unjoined
            check(lib.rmcl_tune_set(10, 1), "tune_set")
            for st in streams[1:]:
                main.wait_stream(st)
        return pb.delta

    def pgd_attack(self, pl_module, batch, k_modality=None):
        eng = pl_module.engine
        img_init = batch["image"][0]
        if hasattr(img_init, "tables"):                      # decoded bytes (collate_raw_uint8): MinMaxResize on the device first
            img_init = eng.resize_raw(img_init)
        if hasattr(img_init, "float_image"):                 # byte batch (collate_uint8): this public API returns / leaves behind images
            img_init = img_init.to(eng.device).float_image()
        pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], img_init)
        delta_p = self.attack_patches(pl_module, pb, k_modality, keep_prev=True)
        B = img_init.shape[0]
        # the reference leaves batch['image'][0] = img_init + delta_{K-1} behind (:144)
        batch["image"][0] = img_init.to(eng.device) + eng.patches_to_image(pb.delta_prev, pb)
        return eng.patches_to_image(delta_p, pb)


class PGDAttack_bartlowtwins(PGDAttack):
    """attack/pgd_attack_vilt.py:178-236 (the reference's spelling of the class name is kept): loss =
    (on_diag + adv_lr * off_diag) / K on c = q^T k / B over the LOCAL batch, q = barlowtwins_head(cls_feats) of a deep copy of
    the head: batch statistics in training (the module's running estimates stay untouched), the running estimates in
    validation."""

    def __init__(self, config):
        super().__init__(config, "barlowtwins")

    def attack_patches(self, pl_module, pb, zk, keep_prev=False, clean_op=None):
        eng = pl_module.engine
        K = self.adv_steps_img
        bb = eng.bt_bufs(pb.B, "pgd")
        mode = bool(pl_module.training)           # deepcopy(pl_module.barlowtwins_head) keeps the train / eval flag (:189)
        op = clean_op if clean_op is not None else eng.make_operand(pb)                 # delta_0 = 0
        for step in range(K):
            last = step == K - 1
            eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op, cls_tail=True)
            eng.heads_forward(pb, key=False, want_q=False, wgrad=False)
            eng.bt_forward(bb, pb.cls, training=mode, track=False)
            eng.bt_loss(bb, zk, float(pb.B), pl_module.adv_lr, 1.0 / K, want_dz=True)
            dcls = eng.bt_backward(bb, bb.dz, training=mode, with_grads=False)
            eng.heads_backward(pb, None, dcls, with_grads=False)
            eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=pb.gpatch)
            if last and keep_prev:
                if K > 1:
                    pb.delta_prev.copy_(pb.delta)
                else:
                    pb.delta_prev.zero_()
            op = pb.patchesT_full if last else pb.patchesT
            eng.pgd_step(pb, self.adv_lr_img, self.adv_max_norm_img, first=step == 0, out=op, sum_prev=last)
        return pb.delta

    def pgd_attack(self, pl_module, batch, k_modality=None):
        eng = pl_module.engine
        img_init = batch["image"][0]
        if hasattr(img_init, "tables"):
            img_init = eng.resize_raw(img_init)
        if hasattr(img_init, "float_image"):
            img_init = img_init.to(eng.device).float_image()
        pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], img_init, tag="bt")
        delta_p = self.attack_patches(pl_module, pb, k_modality.to(eng.device, torch.float32).contiguous(), keep_prev=True)
        batch["image"][0] = img_init.to(eng.device) + eng.patches_to_image(pb.delta_prev, pb)
        return eng.patches_to_image(delta_p, pb)
