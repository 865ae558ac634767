"""``ViLTransformerSS``: the reference's LightningModule API (vilt/modules/vilt_module.py:20-507) over
the MI355X engine.  Same constructor keys, state-dict names, ``infer`` / ``infer_k`` / ``forward`` /
``training_step`` / ``configure_optimizers``; pytorch_lightning is not required (plain nn.Module with a
minimal ``log`` / ``hparams`` shim), the caller may be any loop that does
``loss = m.training_step(batch, i); loss.backward(); opt.step()``."""
from __future__ import annotations

import ctypes as C
import math
import os
import types

import torch
import torch.nn as nn

from ... import _lib as L
from ..._lib import lib, check
from ...runtime import Engine, EMA_GROUPS
from ...attack.pgd_attack_vilt import PGDAttack_moco, PGDAttack_bartlowtwins
from ...attack.greedy_attack_vilt import GreedyAttack_moco, GreedyAttack_barlowtwins
from . import objectives, vilt_utils, dist_utils


def config_patch(module) -> int:
    return int(module.hparams.config["patch_size"])


class _Node(nn.Module):
    """anonymous container so parameters get the reference's dotted state-dict names"""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _InferFeatures(torch.autograd.Function):
    """The four feature tensors of ``infer`` with a backward: their gradients are handed to the closure, which launches the HIP
    heads / encoder backward into the gradient arena (every ``param.grad`` is a view of it)."""

    @staticmethod
    def forward(ctx, anchor, closure, prescale, text_feats, image_feats, cls_feats, raw_cls_feats):
        ctx.closure, ctx.prescale = closure, float(prescale)
        return text_feats.clone(), image_feats.clone(), cls_feats.clone(), raw_cls_feats.clone()

    @staticmethod
    def backward(ctx, g_txt, g_img, g_cls, g_raw):
        sc = (lambda g: None if g is None else (g * ctx.prescale if ctx.prescale != 1.0 else g))
        ctx.closure(sc(g_txt), sc(g_img), sc(g_cls), sc(g_raw))
        return None, None, None, None, None, None, None


class _SlotClosure:
    """The backward closure of one differentiable ``infer`` call together with its stash slot: the slot becomes free again when the
    backward has run or when autograd drops the graph (the closure is garbage-collected with the autograd node)."""

    def __init__(self, fn, busy: set, slot: int):
        self.fn, self.busy, self.slot = fn, busy, slot

    def __call__(self, *grads):
        try:
            self.fn(*grads)
        finally:
            self.busy.discard(self.slot)

    def __del__(self):
        self.busy.discard(self.slot)


class ViLTransformerSS(nn.Module):
    MAX_PENDING_INFER = 4

    def __init__(self, config, device="cuda:0", compute_dtype="bf16", exact=False, pgd_dtype=None):
        super().__init__()
        self.hparams = types.SimpleNamespace(config=config)
        self.config = config
        self.engine = Engine(config, device, compute_dtype, exact, pgd_dtype)
        eng = self.engine
        if config.get("ln_fold") is False:
            eng.fold = {}
        self.current_tasks = []
        self.logged = {}
        self._infer_busy = set()               # stash slots of differentiable infer() calls whose backward is pending
        # query parameters = views into the flat fp32 arena; .grad = views into the gradient arena
        for name, off, shape in eng.specs:
            if name.startswith("itm_score") and config["loss_names"].get("itm", 0) <= 0:
                continue
            if name.startswith("moco_head") and config["loss_names"].get("moco", 0) <= 0:
                continue
            p = nn.Parameter(eng.view(eng.q32, off, shape))
            p.grad = eng.view(eng.g32, off, shape)
            _attach(self, name, p)
        self.init_weights()
        if config["loss_names"].get("moco", 0) > 0:
            self.multimodal = config.get("Multimodal", True)
            self.per_step_bs = config["num_gpus"] * config["num_nodes"] * config["per_gpu_batchsize"]
            for name, off, shape in eng.specs:
                if name.split(".")[0] in EMA_GROUPS:
                    p = nn.Parameter(eng.view(eng.k32, off, shape), requires_grad=False)
                    _attach(self, "k_" + name, p)
            self.shadow_momentum_encoder()
            self.momentum = config["momentum"]
            self.temperature = config["temperature"]
            self.text_view = config["text_view"]
            self.augmentation = config["augmentation"]
            self.image_view = config["image_view"]
            self.num_negative = config["num_negative"]
            self.register_buffer("proj_queue", torch.randn(128, self.num_negative, device=eng.device))
            self.register_buffer("proj_queue_ptr", torch.zeros(1, dtype=torch.long, device=eng.device))
            eng.queue = self.proj_queue
            self._queue_ptr_host = 0
            if self.image_view and not self.augmentation:
                self.pgd_attacker = PGDAttack_moco(config)
            if self.text_view and not self.augmentation:
                self.greedy_attacker = GreedyAttack_moco(config)
        if config["loss_names"].get("barlowtwins", 0) > 0:                 # vilt_module.py:109-131
            self.multimodal = config.get("Multimodal", True)
            self.per_step_bs = config["num_gpus"] * config["num_nodes"] * config["per_gpu_batchsize"]
            self.text_view = config["text_view"]
            self.image_view = config["image_view"]
            self.augmentation = config["augmentation"]
            self.adv_lr = config["adv_lr"]
            self.loss_weight = 0.001
            H = eng.bt
            o = 0
            for i, (key, n) in enumerate((("projector.1", H.H1), ("projector.4", H.H2), ("norm", H.H3))):
                node = self                                               # nn.BatchNorm1d buffers under the reference's names
                for part in ("barlowtwins_head." + key).split("."):
                    if part not in node._modules:
                        node.add_module(part, _Node())
                    node = node._modules[part]
                node.register_buffer("running_mean", eng.bt_running[o:o + n])
                node.register_buffer("running_var", eng.bt_running[o + n:o + 2 * n])
                node.register_buffer("num_batches_tracked", eng.bt_tracked[i])
                o += 2 * n
            if self.text_view and not self.augmentation:
                self.greedy_attacker = GreedyAttack_barlowtwins(config)
            if self.image_view and not self.augmentation:
                self.pgd_attacker = PGDAttack_bartlowtwins(config)
        self.grad_anchor = torch.zeros((), device=eng.device, requires_grad=True)
        self.sync_grads = True                     # False on the early micro-steps of gradient accumulation (DDP no_sync)
        self.step_sync = dist_utils.StepGradSync(algo=config.get("grad_allreduce_algo", "ring"), compress=config.get("grad_allreduce_dtype"))
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._after_load())
        # downstream checkpoint (vilt_module.py:134-160, test_only twin :252-268): loaded AFTER the momentum copies were
        # shadowed, strict=False, like the reference - so k_* keys absent from the file keep their pre-load values
        if config.get("load_path", "") not in ("", None):
            self._load_checkpoint(config["load_path"])

    def _load_checkpoint(self, path):
        # weights_only=True: executes nothing from the file (the reference uses a full unpickle, vilt_module.py:138)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        state_dict = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        if self.hparams.config["loss_names"].get("itm", 0) > 0:          # ITM head from the 200k checkpoint (:152-159)
            for cand in ("models_weight/vilt_200k_mlm_itm.ckpt", "../models_weight/vilt_200k_mlm_itm.ckpt"):
                if os.path.isfile(cand):
                    c2 = torch.load(cand, map_location="cpu", weights_only=True)["state_dict"]
                    state_dict["itm_score.fc.weight"] = c2["itm_score.fc.weight"]
                    state_dict["itm_score.fc.bias"] = c2["itm_score.fc.bias"]
                    break
        mine = self.state_dict()
        # a tensor of another shape (position table at another resolution, text positions at another max_text_len) is an ERROR like
        # in the reference, whose load_state_dict(strict=False) raises on size mismatches (vilt_module.py:159-160) - not a silent skip
        wrong = [(k, tuple(v.shape), tuple(mine[k].shape)) for k, v in state_dict.items() if k in mine and tuple(mine[k].shape) != tuple(v.shape)]
        if wrong:
            raise RuntimeError("Error(s) in loading state_dict: " + "; ".join(f"size mismatch for {k}: checkpoint {a}, model {b}" for k, a, b in wrong))
        res = self.load_state_dict({k: v for k, v in state_dict.items() if k in mine}, strict=False)
        unexpected = sorted(k for k in state_dict if k not in mine)
        self.load_report = {"missing": sorted(res.missing_keys), "unexpected": unexpected}
        if res.missing_keys or unexpected:
            import warnings
            warnings.warn(f"load_path={path}: {len(res.missing_keys)} model tensors not in the checkpoint (kept: "
                          f"{', '.join(sorted(res.missing_keys)[:6])}{' ...' if len(res.missing_keys) > 6 else ''}); "
                          f"{len(unexpected)} checkpoint tensors unused ({', '.join(unexpected[:6])}{' ...' if len(unexpected) > 6 else ''})")
        # LayerNorm fold after a checkpoint load.  Round 3 switched it off here: the folded GEMM ate bf16(x) of the RAW residual stream,
        # whose rounding grows with the row's common-mode offset (a trained checkpoint's stream is not zero-mean).  Since round 4 the
        # producers store bf16(x - c) with c = the row mean of the row's previous LayerNorm (LN(x) = LN(x - c) exactly; gemm.h
        # ln_center), so the operand's rounding follows the row's spread whatever its offset (tests/test_kernels_gpu.py
        # test_layernorm_fold_precision_on_offset_rows: fold within 2x of the separate LayerNorm at 10 and 50 sigma) and the fold STAYS
        # ON.  config["ln_fold"] = False switches it off (separate LayerNorm kernels in every pass).
        self.load_report["ln_fold"] = "on (shift-robust form)" if self.engine.fold else "off"
        return res

    # ---- initialisation (objectives.init_weights :1505-1516, ViT _init_weights :512-519) ----
    @torch.no_grad()
    def init_weights(self):
        for name, p in self.named_parameters():
            if name.startswith("k_"):
                continue
            leaf = name.split(".")[-1]
            is_ln = any(t in name for t in ("LayerNorm", "norm1", "norm2", "transformer.norm", "projector.1"))
            if name.startswith("barlowtwins_head."):
                # never passed through init_weights in the reference (vilt_module.py:115): nn.Linear / nn.BatchNorm1d defaults
                if p.dim() == 2:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                else:
                    p.fill_(1.0 if leaf == "weight" else 0.0)
            elif is_ln:
                p.fill_(1.0 if leaf == "weight" else 0.0)
            elif name.startswith("transformer.patch_embed.proj"):
                # the reference leaves the patch projection at nn.Conv2d's default init (vision_transformer.py:397-403;
                # VisionTransformer._init_weights :512-519 only touches nn.Linear / nn.LayerNorm)
                fan_in = p.shape[1] * p.shape[2] * p.shape[3] if p.dim() == 4 else 3 * config_patch(self) ** 2
                if leaf == "weight":
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                else:
                    bound = 1.0 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound)
            elif leaf == "bias":
                p.zero_()
            elif name.startswith("transformer."):
                nn.init.trunc_normal_(p, std=0.02)
            else:
                p.normal_(mean=0.0, std=0.02)
        self.engine.lp_stale = True

    @torch.no_grad()
    def shadow_momentum_encoder(self):
        """_shadow_layer (vilt_module.py:270-273): copy q -> k."""
        e = self.engine
        e.k32.copy_(e.q32[: e.layout.ema_end])
        e.lp_stale = True

    def _after_load(self):
        self.engine.lp_stale = True
        if hasattr(self, "proj_queue_ptr"):
            self._queue_ptr_host = None

    # ---- Lightning shims -------------------------------------------------------------------
    @property
    def device(self):
        return self.engine.device

    def log(self, name, value):
        self.logged[name] = value

    @property
    def queue_ptr(self) -> int:
        if self._queue_ptr_host is None:
            self._queue_ptr_host = int(self.proj_queue_ptr)
        return self._queue_ptr_host

    @queue_ptr.setter
    def queue_ptr(self, v: int):
        self._queue_ptr_host = int(v)
        self.proj_queue_ptr.fill_(int(v))

    def after_backward(self, overlap: bool = False):
        """DDP replacement (run.py:96).  Called by every deferred-backward closure right after its HIP backward has been
        ENQUEUED.  The arena is reduced once per step, after the LAST closure (dist_utils.StepGradSync): with a single
        closure (the MoCo image-view step) as per-layer all-reduces gated on the backward's gradient-ready events
        (overlap=True), otherwise as one blocking pass over the arena."""
        e = self.engine
        factory = None
        if overlap and e.g32.is_cuda and os.environ.get("RMCL_NO_GRAD_OVERLAP", "0") != "1":
            lay = e.layout

            def gate(layer, stream):
                check(lib.rmcl_grad_ready_wait(int(layer), C.c_void_p(stream.cuda_stream)), "grad_ready_wait")

            def factory():
                buckets = dist_utils.grad_buckets(int(lay.layer0), int(lay.layer_stride), int(self.hparams.config["num_layers"]),
                                                  int(e.g32.numel()))
                return dist_utils.GradSync(e.g32, buckets, e.comm_stream, gate, prescaled=True,
                                           compress=self.hparams.config.get("grad_allreduce_dtype"),
                                           algo=self.hparams.config.get("grad_allreduce_algo", "ring"))

        self.step_sync.closure_done(e.g32, enabled=self.sync_grads, overlap=factory)

    def grad_prescale(self) -> float:
        """Registers one deferred backward with the step's gradient reducer and returns 1 / world_size: the mean over
        ranks is applied to the loss gradient of EVERY closure (objectives._DeferredBackward), also on micro-steps
        whose reduction is skipped (sync_grads=False: gradient accumulation)."""
        return self.step_sync.register()

    def wait_grad_sync(self):
        """Make the current stream wait for the overlapped gradient all-reduces (the optimizer calls this)."""
        self.step_sync.wait()

    def zero_grad(self, set_to_none: bool = False):
        self.engine.zero_grads()

    # ---- inference API ---------------------------------------------------------------------
    def _infer(self, batch, key, mask_text, mask_image, image_token_type_idx, image_embeds, image_masks):
        if mask_text or mask_image:
            raise NotImplementedError("MLM/MPP masking is outside the RMCL hot path")
        if image_embeds is not None or image_masks is not None:
            raise NotImplementedError("image_embeds shortcut is not on the RMCL hot path")
        if image_token_type_idx != 1:
            raise NotImplementedError("image_token_type_idx != 1 (NLVR2) is outside the RMCL hot path")
        eng = self.engine
        eng.dropout_on = self.training and eng.drop_p > 0
        text_ids, text_masks = batch["text_ids"], batch["text_masks"]
        # the reference's infer is an ordinary differentiable forward (vilt_module.py:275-351): with autograd on, the query pass keeps
        # the FULL stash and the returned features carry a backward into the gradient arena (the momentum pass never does: k_* get no
        # gradients, vilt_module.py:270-273)
        need_grad = (not key) and torch.is_grad_enabled()
        # every differentiable call whose backward is still pending keeps its OWN FULL stash (a second infer() of the same batch
        # size must not overwrite the activations the first one's backward will read); the slot is returned by the backward, or
        # when autograd drops the graph
        slot = self._infer_slot_take() if need_grad else None
        pb = eng.bind_batch(text_ids, text_masks, batch["image"][0], tag=f"infer{slot}" if need_grad else "moco")
        op = eng.make_operand(pb)
        eng.encoder_forward(pb, key=key, mode=L.MODE_FULL if need_grad else L.MODE_INFER, patchesT=op)
        eng.heads_forward(pb, key=key, want_q=False)
        d = pb.d
        N = d.L + 1 + d.P
        x = pb.xn.view(pb.B, N, d.D)
        if pb.geom is None:
            g = self.config["image_size"] // self.config["patch_size"]
            ii, jj = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
            patch_index = torch.stack([ii, jj], dim=-1).reshape(1, g * g, 2).expand(pb.B, -1, -1)
            grid_hw = (g, g)
        else:                                                  # zero-padded batch: (row, col) of every selected slot
            sel = pb.geom.sel[:, : pb.geom.n].to(torch.int64)
            patch_index = torch.stack([sel // pb.geom.gw, sel % pb.geom.gw], dim=-1)
            grid_hw = (pb.geom.gh, pb.geom.gw)
        feats = (x[:, : d.L].clone(), x[:, d.L:].clone(), pb.cls.clone(), x[:, 0].clone())
        if need_grad:
            def backward(g_txt, g_img, g_cls, g_raw, pb=pb, op=op, N=N):
                dxn = torch.zeros(pb.B, N, d.D, device=eng.device)
                if g_txt is not None:
                    dxn[:, : d.L] += g_txt.to(torch.float32)
                if g_img is not None:
                    dxn[:, d.L:] += g_img.to(torch.float32)
                if g_raw is not None:
                    dxn[:, 0] += g_raw.to(torch.float32)
                dcls = torch.zeros(pb.B, d.D, device=eng.device) if g_cls is None else g_cls.to(torch.float32).contiguous()
                eng.heads_backward(pb, None, dcls, with_grads=True)          # pooler: tanh(W x[:, 0] + b) (heads.py:10-20)
                dxn[:, 0] += pb.dcls
                eng.encoder_backward(pb, L.MODE_FULL, op, dxn.view(pb.B * N, d.D), cls_only=False, dpatches=None)
                self.after_backward()

            # the step belongs to whoever opened it (forward() / training_step); a stand-alone infer() opens one only when no
            # deferred backward of an earlier call is still outstanding - it never wipes their registrations
            if self.step_sync.open <= 0:
                self.step_sync.begin_step()
            feats = _InferFeatures.apply(self.grad_anchor, _SlotClosure(backward, self._infer_busy, slot), self.grad_prescale(), *feats)
        ret = {
            "text_feats": feats[0],
            "image_feats": feats[1],
            "cls_feats": feats[2],
            "raw_cls_feats": feats[3],
            "image_masks": pb.co_mask[:, d.L:].to(torch.int64),
            "text_ids": text_ids,
            "text_masks": text_masks,
            "patch_index": (patch_index, grid_hw),
        }
        if not key:
            ret["image_labels"] = None
            ret["text_labels"] = batch.get("text_labels")
        return ret

    def _infer_slot_take(self) -> int:
        for slot in range(self.MAX_PENDING_INFER):
            if slot not in self._infer_busy:
                self._infer_busy.add(slot)
                return slot
        raise RuntimeError(f"infer(): {self.MAX_PENDING_INFER} differentiable calls are still waiting for their backward (each keeps a full "
                           "activation stash); back-propagate or drop their outputs first, or call infer() under torch.no_grad() for inference")

    def infer(self, batch, mask_text=False, mask_image=False, image_token_type_idx=1, image_embeds=None, image_masks=None):
        """vilt_module.py:275-351.  Differentiable like the reference's when autograd is on (``text_feats`` / ``image_feats`` /
        ``cls_feats`` / ``raw_cls_feats`` back-propagate into every query parameter's ``.grad``); ``torch.no_grad()`` gives the
        stash-free inference pass.  The ``image_embeds`` shortcut (pre-computed visual_embed output, used by the reference's
        downstream tasks only) is not built."""
        return self._infer(batch, False, mask_text, mask_image, image_token_type_idx, image_embeds, image_masks)

    @torch.no_grad()
    def infer_k(self, batch, mask_text=False, mask_image=False, image_token_type_idx=1, image_embeds=None, image_masks=None):
        return self._infer(batch, True, mask_text, mask_image, image_token_type_idx, image_embeds, image_masks)

    def forward(self, batch):
        self.step_sync.begin_step()
        self.engine.dropout_on = self.training and self.engine.drop_p > 0
        ret = dict()
        if len(self.current_tasks) == 0:
            ret.update(self.infer(batch))
            return ret
        if "itm" in self.current_tasks:
            ret.update(objectives.compute_itm_wpa(self, batch))
        if "moco" in self.current_tasks:
            ret.update(objectives.compute_moco_contrastive(self, batch))
        if "barlowtwins" in self.current_tasks:
            ret.update(objectives.compute_barlowtwins_contrastive(self, batch))
        unsupported = [t for t in self.current_tasks if t not in ("itm", "moco", "barlowtwins")]
        if unsupported:
            raise NotImplementedError(f"tasks {unsupported} are outside the RMCL hot path (SURVEY 8)")
        return ret

    def training_step(self, batch, batch_idx):
        vilt_utils.set_task(self)
        output = self(batch)
        total_loss = sum([v for k, v in output.items() if "loss" in k])
        return total_loss

    def training_epoch_end(self, outs=None):
        vilt_utils.epoch_wrapup(self)

    def validation_step(self, batch, batch_idx):
        vilt_utils.set_task(self)
        with torch.no_grad():
            return self(batch)

    def validation_epoch_end(self, outs=None):
        vilt_utils.epoch_wrapup(self)

    def test_epoch_end(self, outs=None):
        vilt_utils.epoch_wrapup(self)

    def test_step(self, batch, batch_idx):
        return self.validation_step(batch, batch_idx)

    def configure_optimizers(self):
        return vilt_utils.set_schedule(self)
