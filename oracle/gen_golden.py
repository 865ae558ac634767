"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own arithmetic.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference, which
does not exist on the GPU box).  Nothing here is imported by the product.

How the reference is imported: its files are imported unmodified from
/root/reference.  Third-party packages that are absent from this image and contribute no
forward arithmetic on this path get minimal stand-ins in ``sys.modules`` (timm: identity
DropPath, to_2tuple, trunc_normal_ -> torch.nn.init.trunc_normal_, registry decorator;
torchvision / pytorch_lightning / sacred / nltk: import-time names only).  ``vilt`` and
``vilt.modules`` are registered as namespace packages so ``vilt/modules/__init__.py`` (which
drags in the Lightning module, nltk and a weight download) is never executed.
``ViLTransformerSS`` itself cannot be built offline (vilt_module.py:78-80 downloads weights),
so a small holder module owns the reference's sub-modules and exposes the attributes that
``objectives.compute_moco_contrastive`` / ``compute_itm_wpa`` read.  ``infer`` / ``infer_k`` are
served by the reference's ``PGDAttack.infer`` (attack/pgd_attack_vilt.py:29-106, a twin of
vilt_module.py:275-351) pointed at the q-modules, resp. k-modules + q-pooler (:405).

Weights: ``oracle.rmcl_oracle.init_params(cfg, seed)`` loaded into the reference modules via
their state-dict names, so a fixture is reproducible from (cfg, seed) without the reference.

Usage:  python oracle/gen_golden.py [moco itm moco2 cleanitm txtatk sched ragged pipeline dataset]   (writes tests/golden/*.npz, ~2 min)
"""
from __future__ import annotations

import os
import sys
import types
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import rmcl_oracle as O  # noqa: E402


def _install_standins():
    import transformers.models.bert.modeling_bert  # noqa: F401  (must precede the stand-ins)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def no_fetch(*a, **k):
        raise RuntimeError("no weight download offline")

    mod("timm")
    mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    mod("timm.models")
    mod("timm.models.helpers", load_pretrained=no_fetch)
    mod("timm.models.layers", StdConv2dSame=nn.Conv2d, DropPath=DropPath, to_2tuple=to_2tuple,
        trunc_normal_=torch.nn.init.trunc_normal_)
    mod("timm.models.resnet", resnet26d=None, resnet50d=None)
    mod("timm.models.resnetv2", ResNetV2=None)
    mod("timm.models.registry", register_model=lambda f: f)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=lambda x: x, Normalize=lambda **kw: ("normalize", kw), ToTensor=lambda: "to_tensor")
    pl = mod("pytorch_lightning", LightningModule=nn.Module)
    pl.metrics = mod("pytorch_lightning.metrics", Metric=object)
    mod("TSNE_vizualisation", TSNE_projection=None)
    # attack/greedy_attack_vilt.py:2,7-8 imports nltk at module top; its stop-word / wordnet loaders are never called
    nl = mod("nltk")
    nl.corpus = mod("nltk.corpus", stopwords=None, wordnet=None)
    for pkg in ("vilt", "vilt.modules"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m
    sys.path.insert(0, REF)


_install_standins()
import vilt.modules.vision_transformer as vit  # noqa: E402  (reference, unmodified)
import vilt.modules.heads as heads  # noqa: E402
import vilt.modules.objectives as objectives  # noqa: E402
from attack.pgd_attack_vilt import PGDAttack, PGDAttack_moco, PGDAttack_bartlowtwins  # noqa: E402
from attack.greedy_attack_vilt import GreedyAttack_moco, GreedyAttack_barlowtwins  # noqa: E402
from transformers.models.bert.modeling_bert import BertConfig, BertEmbeddings  # noqa: E402


class Holder(nn.Module):
    """Owns the reference sub-modules; mirrors the attributes of ViLTransformerSS read on the path."""

    def __init__(self, cfg):
        super().__init__()
        self.hparams = types.SimpleNamespace(config=dict(cfg, vit="vit_base_patch32_384"))
        bc = BertConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                        num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"],
                        intermediate_size=cfg["hidden_size"] * cfg["mlp_ratio"],
                        max_position_embeddings=cfg["max_text_len"],
                        hidden_dropout_prob=cfg["drop_rate"], attention_probs_dropout_prob=cfg["drop_rate"])

        def mk_vit():
            return vit.VisionTransformer(img_size=cfg["image_size"], patch_size=cfg["patch_size"],
                                         embed_dim=cfg["hidden_size"], depth=cfg["num_layers"],
                                         num_heads=cfg["num_heads"], config=cfg)

        D = cfg["hidden_size"]
        self.text_embeddings = BertEmbeddings(bc)
        self.token_type_embeddings = nn.Embedding(2, D)
        self.transformer = mk_vit()
        self.pooler = heads.Pooler(D)
        self.itm_score = heads.ITMHead(D)
        self.k_text_embeddings = BertEmbeddings(bc)
        self.k_token_type_embeddings = nn.Embedding(2, D)
        self.k_transformer = mk_vit()
        self.moco_head = heads.MOCOHead(D, D, cfg["proj_dim"])
        self.k_moco_head = heads.MOCOHead(D, D, cfg["proj_dim"])
        self.momentum = cfg["momentum"]
        self.temperature = cfg["temperature"]
        self.text_view, self.image_view, self.augmentation = cfg["text_view"], cfg["image_view"], cfg["augmentation"]
        self.num_negative = cfg["num_negative"]
        self.per_step_bs = cfg["num_gpus"] * cfg["num_nodes"] * cfg["per_gpu_batchsize"]
        self.cosine = nn.CosineSimilarity(dim=1, eps=1e-6)
        self.register_buffer("proj_queue", torch.zeros(cfg["proj_dim"], self.num_negative))
        self.register_buffer("proj_queue_ptr", torch.zeros(1, dtype=torch.long))
        self.pgd_attacker = PGDAttack_moco(cfg)
        self.logged = {}
        self._q = PGDAttack(cfg, "q")
        self._k = PGDAttack(cfg, "k")
        for name in ("train", "val"):
            for met in ("moco_loss", "itm_loss", "itm_wpa_loss"):
                setattr(self, f"{name}_{met}", lambda x: x)
            setattr(self, f"{name}_itm_accuracy", lambda lg, lb: (lg.argmax(-1) == lb).float().mean())

    @property
    def device(self):
        return torch.device("cpu")

    def log(self, name, value):
        self.logged[name] = float(value)

    def infer(self, batch, **kw):
        a = self._q
        a.text_embeddings, a.token_type_embeddings = self.text_embeddings, self.token_type_embeddings
        a.transformer, a.pooler = self.transformer, self.pooler
        return a.infer(batch, **kw)

    def infer_k(self, batch, **kw):
        a = self._k
        a.text_embeddings, a.token_type_embeddings = self.k_text_embeddings, self.k_token_type_embeddings
        a.transformer, a.pooler = self.k_transformer, self.pooler
        return a.infer(batch, **kw)

    def load_oracle_params(self, p):
        sd = self.state_dict()
        for n, t in p.items():
            assert n in sd, n
            assert sd[n].shape == t.shape, (n, sd[n].shape, t.shape)
            sd[n].copy_(t)
        for n, prm in self.named_parameters():
            if n.startswith("k_"):
                prm.requires_grad = False


NUDGE = 0.002         # step size of the SGD nudge between the two steps of the moco2 fixtures


def tensor_digest(t: torch.Tensor) -> np.ndarray:
    """[sum, l2, abs-max, first 8 values] of a tensor, as float64."""
    f = t.detach().double().flatten()
    head = torch.zeros(8, dtype=torch.float64)
    head[: min(8, f.numel())] = f[:8]
    return torch.cat([torch.stack([f.sum(), f.norm(), f.abs().max()]), head]).numpy()


RAGGED_SIZES = [(384, 352), (320, 384), (384, 384), (224, 288)]     # zero-padded to 384 x 384; 132 / 120 / 144 / 63 valid patches


def run_moco(tag, cfg, B, seed_w, seed_b, ragged, sizes=None):
    torch.manual_seed(1234)
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    cfg = dict(cfg, per_gpu_batchsize=B)
    p = O.init_params(cfg, seed_w)
    h = Holder(cfg)
    h.load_oracle_params({n: t for n, t in p.items()})
    h.proj_queue.copy_(O.init_queue(cfg, 0))
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged, sizes=sizes)
    out = {}

    # (1) plain infer, both encoders
    with torch.no_grad():
        r = h.infer(deepcopy(batch))
        out["cls_feats"] = r["cls_feats"].numpy()
        out["raw_cls_feats"] = r["raw_cls_feats"].numpy()
        out["text_feats"] = r["text_feats"].numpy()
        # image_feats come back in a random patch order (vision_transformer.py:633-636): un-permute
        pi = r["patch_index"][0]                               # [B, n, 2] (row, col)
        G = cfg["image_size"] // cfg["patch_size"]
        flat = pi[..., 0] * G + pi[..., 1]
        img_f = r["image_feats"]
        if sizes is None:
            dense = torch.zeros_like(img_f)
            dense[:, 0] = img_f[:, 0]
            for b in range(B):
                dense[b, 1 + flat[b]] = img_f[b, 1:]
            out["image_feats"] = dense.numpy()
        else:
            # zero-padded batch: the reference keeps the valid patches in row-major order and pads with randomly drawn
            # NON-valid patches, which are all identical tokens - no un-permutation needed; the draw itself is recorded
            # (a sample with exactly n valid patches comes back as a random PERMUTATION of them - multinomial without
            # replacement over all of them, :633-636: its valid tokens are put back in row-major order here)
            img_f = img_f.clone()
            msk = r["image_masks"]
            for b in range(B):
                nv = int(msk[b, 1:].sum())
                order = torch.argsort(flat[b, :nv])
                img_f[b, 1:1 + nv] = img_f[b, 1:1 + nv][order]
                flat[b, :nv] = flat[b, :nv][order]
            out["image_feats"] = img_f.numpy()
            out["patch_index_flat"] = flat.numpy()
            out["image_masks"] = r["image_masks"].numpy()

    # (2) PGD alone (K steps and 1 step), momentum copies == query weights here, k from infer_k
    with torch.no_grad():
        rk = h.infer_k(deepcopy(batch))
        k0 = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1)
    out["pgd_k_input"] = k0.numpy()
    for K in (1, cfg["adv_steps_img"]):
        att = PGDAttack_moco(dict(cfg, adv_steps_img=K))
        d = att.pgd_attack(h, deepcopy(batch), k_modality=k0)
        out[f"pgd_delta_K{K}_sub"] = d[:, :, ::8, ::8].contiguous().numpy()
        out[f"pgd_delta_K{K}_digest"] = tensor_digest(d)
        out[f"pgd_delta_K{K}_persample_l2"] = d.flatten(1).norm(dim=1).numpy()
        out[f"pgd_delta_K{K}_patch00"] = d[:, :, :32, :32].contiguous().numpy()

    # (3) the full step: objectives.compute_moco_contrastive + backward
    h.zero_grad()
    ret = objectives.compute_moco_contrastive(h, deepcopy(batch))
    loss = sum(v for kk, v in ret.items() if "loss" in kk)     # vilt_module.py:475
    loss.backward()
    out["moco_loss"] = np.float64(loss.item())
    for kk, v in ret.items():
        if kk != "moco_loss":
            out["ret_" + kk] = np.float64(float(v))
    for kk, v in h.logged.items():
        out["log_" + kk.replace("/", "__")] = np.float64(v)
    out["queue_ptr_after"] = np.int64(int(h.proj_queue_ptr))
    out["queue_head_after"] = h.proj_queue[:, : 2 * B].numpy().copy()
    names, gd, kd = [], [], []
    for n, prm in h.named_parameters():
        if n.startswith("k_"):
            kd.append(tensor_digest(prm))
            names.append(n)
    out["ema_names"] = np.array(names)
    out["ema_digest"] = np.stack(kd)
    gnames = []
    for n, prm in h.named_parameters():
        if not n.startswith("k_") and prm.grad is not None:
            gnames.append(n)
            gd.append(tensor_digest(prm.grad))
    out["grad_names"] = np.array(gnames)
    out["grad_digest"] = np.stack(gd)
    out["grad_pooler_w"] = h.pooler.dense.weight.grad[:8, :64].numpy().copy()
    out["grad_qkv0_w"] = h.transformer.blocks[0].attn.qkv.weight.grad[:8, :64].numpy().copy()
    out["grad_patch_w"] = h.transformer.patch_embed.proj.weight.grad[:4, :, :4, :8].numpy().copy()
    out["grad_pos_embed"] = h.transformer.pos_embed.grad[0, :4, :64].numpy().copy()
    we = h.text_embeddings.word_embeddings.weight.grad
    ids = batch["text_ids"]
    out["grad_word_rows"] = we[ids[0, :4]][:, :64].numpy().copy()
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"], cfg["num_negative"], cfg["adv_steps_img"]])
    if sizes is not None:
        out["sizes"] = np.array(sizes)
    path = os.path.join(ROOT, "tests", "golden", f"moco_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "moco_loss", out["moco_loss"], "bytes", os.path.getsize(path))


def run_itm(tag, cfg, B, seed_w, seed_b, ragged):
    torch.manual_seed(4321)
    cfg = dict(cfg, per_gpu_batchsize=B)
    p = O.init_params(cfg, seed_w)
    h = Holder(cfg)
    h.load_oracle_params(p)
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged)
    h.zero_grad()
    ret = objectives.compute_itm_wpa(h, batch)
    loss = sum(v for kk, v in ret.items() if "loss" in kk)
    loss.backward()
    out = {"itm_loss": np.float64(ret["itm_loss"].item()), "itm_wpa_loss": np.float64(ret["itm_wpa_loss"].item()),
           "itm_logits": ret["itm_logits"].detach().numpy(), "itm_labels": ret["itm_labels"].numpy()}
    gnames, gd = [], []
    for n, prm in h.named_parameters():
        if not n.startswith("k_") and prm.grad is not None:
            gnames.append(n)
            gd.append(tensor_digest(prm.grad))
    out["grad_names"] = np.array(gnames)
    out["grad_digest"] = np.stack(gd)
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"]])
    path = os.path.join(ROOT, "tests", "golden", f"itm_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "itm", out["itm_loss"], out["itm_wpa_loss"], "bytes", os.path.getsize(path))


def _grad_digests(h):
    gnames, gd = [], []
    for n, prm in h.named_parameters():
        if not n.startswith("k_") and prm.grad is not None:
            gnames.append(n)
            gd.append(tensor_digest(prm.grad))
    return np.array(gnames), np.stack(gd)


def _ema_digests(h):
    names, kd = [], []
    for n, prm in h.named_parameters():
        if n.startswith("k_"):
            names.append(n)
            kd.append(tensor_digest(prm))
    return np.array(names), np.stack(kd)


def run_moco_two_step(tag, cfg, B, seed_w, seed_k, seed_b, ragged, ptr0, nudge):
    """Two consecutive reference steps from a NON-degenerate state: k_* != q (own seed), queue pointer != 0,
    momentum < 1, and an SGD-style nudge q <- q - nudge * grad between the steps, so that the EMA, the enqueue
    offset, the key encoder's own weights and the 'k modules + query pooler' quirk (vilt_module.py:405) all
    carry information in the fixture."""
    torch.manual_seed(1234)
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    cfg = dict(cfg, per_gpu_batchsize=B)
    p = O.init_params(cfg, seed_w, k_seed=seed_k)
    h = Holder(cfg)
    h.load_oracle_params(p)
    h.proj_queue.copy_(O.init_queue(cfg, 0))
    h.proj_queue_ptr[0] = ptr0
    h.train()
    out = {}
    b0 = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged)
    with torch.no_grad():
        rk = h.infer_k(deepcopy(b0))                        # k modules + QUERY pooler, before any EMA
        out["init_k_cls_feats"] = rk["cls_feats"].numpy()
        out["init_k_raw_cls_feats"] = rk["raw_cls_feats"].numpy()
        out["init_k_proj"] = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1).numpy()
        rq = h.infer(deepcopy(b0))
        out["init_q_cls_feats"] = rq["cls_feats"].numpy()
    for s in range(2):
        batch = O.synthetic_batch(cfg, B, seed_b + s, ragged_text=ragged)
        h.zero_grad()
        h.logged = {}
        ret = objectives.compute_moco_contrastive(h, deepcopy(batch))
        loss = sum(v for kk, v in ret.items() if "loss" in kk)
        loss.backward()
        out[f"s{s}_moco_loss"] = np.float64(loss.item())
        out[f"s{s}_ptr_after"] = np.int64(int(h.proj_queue_ptr))
        with torch.no_grad():                               # the key of this step: k weights are final after the EMA
            rk = h.infer_k(deepcopy(batch))
            out[f"s{s}_k"] = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1).numpy()
        out["ema_names"], out[f"s{s}_ema_digest"] = _ema_digests(h)
        out["grad_names"], out[f"s{s}_grad_digest"] = _grad_digests(h)
        out[f"s{s}_k_qkv0_w"] = h.k_transformer.blocks[0].attn.qkv.weight[:8, :64].detach().numpy().copy()
        out[f"s{s}_delta_log"] = np.float64(h.logged["moco_attack/train/delta"])
        for kk, v in ret.items():
            if kk != "moco_loss":
                out[f"s{s}_ret_" + kk] = np.float64(float(v))
        if s == 0:
            with torch.no_grad():
                for n, prm in h.named_parameters():
                    if not n.startswith("k_") and prm.grad is not None:
                        prm.data -= nudge * prm.grad
    out["queue_block_after"] = h.proj_queue[:, ptr0 - B: min(ptr0 + 3 * B, cfg["num_negative"])].numpy().copy()
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"], cfg["num_negative"], cfg["adv_steps_img"]])
    out["meta2"] = np.array([seed_k, ptr0, cfg["momentum"], nudge], dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", f"moco2_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "two-step losses", out["s0_moco_loss"], out["s1_moco_loss"], "bytes", os.path.getsize(path))


def run_clean_itm(tag, cfg, B, seed_w, seed_k, seed_b, ragged):
    """BASELINE configs[1] 'clean ITM + contrastive' from the reference's pieces: the clean logits exactly as
    objectives.py:262-275 forms them (EMA, infer_k, infer, moco_head, normalize, einsum, /T) + CrossEntropyLoss
    with label 0 like :333,351, plus the unmodified compute_itm_wpa; total = sum of the three losses."""
    torch.manual_seed(777)
    cfg = dict(cfg, per_gpu_batchsize=B)
    p = O.init_params(cfg, seed_w, k_seed=seed_k)
    h = Holder(cfg)
    h.load_oracle_params(p)
    h.proj_queue.copy_(O.init_queue(cfg, 0))
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged)
    h.zero_grad()
    ret_itm = objectives.compute_itm_wpa(h, deepcopy(batch))
    with torch.no_grad():                                    # objectives.py:219-224,257-260
        for q_l, k_l in ((h.text_embeddings, h.k_text_embeddings), (h.token_type_embeddings, h.k_token_type_embeddings),
                         (h.transformer, h.k_transformer), (h.moco_head, h.k_moco_head)):
            for pq, pk in zip(q_l.parameters(), k_l.parameters()):
                pk.data = pk.data * h.momentum + pq.data * (1.0 - h.momentum)
        rk = h.infer_k(deepcopy(batch), mask_text=False, mask_image=False)
        k = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1)
    inf = h.infer(deepcopy(batch), mask_text=False, mask_image=False)
    q_original = nn.functional.normalize(h.moco_head(inf["cls_feats"]), dim=1)
    neg_k = h.proj_queue.clone().detach()
    l_pos = torch.einsum('nc,nc->n', [q_original, k]).unsqueeze(-1)
    l_neg = torch.einsum('nc,ck->nk', [q_original, neg_k])
    logits = torch.cat([l_pos, l_neg], dim=1)
    logits = logits / h.temperature
    labels = torch.zeros(logits.shape[0], dtype=torch.long)
    clean_loss = nn.CrossEntropyLoss()(logits.float(), labels)
    total = ret_itm["itm_loss"] + ret_itm["itm_wpa_loss"] + clean_loss
    total.backward()
    out = {"itm_loss": np.float64(ret_itm["itm_loss"].item()), "itm_wpa_loss": np.float64(ret_itm["itm_wpa_loss"].item()),
           "clean_loss": np.float64(clean_loss.item()), "total_loss": np.float64(total.item()),
           "itm_labels": ret_itm["itm_labels"].numpy(), "itm_logits": ret_itm["itm_logits"].detach().numpy(),
           "k": k.numpy(), "q_original": q_original.detach().numpy(),
           "logits_head": logits[:, :64].detach().numpy(), "prediction_original": logits.argmax(-1).numpy()}
    out["grad_names"], out["grad_digest"] = _grad_digests(h)
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"], cfg["num_negative"], seed_k])
    path = os.path.join(ROOT, "tests", "golden", f"cleanitm_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "clean+itm", out["clean_loss"], out["itm_loss"], out["itm_wpa_loss"], "bytes", os.path.getsize(path))


def run_text_attack(tag, cfg, B, seed_w, seed_k, seed_b, ragged, n_cand):
    """Tensor side of the greedy text attack from the reference's own GreedyAttack_moco.get_grad / split_forward
    (attack/greedy_attack_vilt.py:406-492).  The object is made with __new__ (its __init__ loads a tokenizer by
    name and nltk stop words - unavailable offline and never needed by these two methods); build_mini_vilt
    (:391-397) deep-copies the holder's modules like the reference does."""
    torch.manual_seed(99)
    cfg = dict(cfg, per_gpu_batchsize=B)
    p = O.init_params(cfg, seed_w, k_seed=seed_k)
    h = Holder(cfg)
    h.load_oracle_params(p)
    h.proj_queue.copy_(O.init_queue(cfg, 0))
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged)
    with torch.no_grad():
        rk = h.infer_k(deepcopy(batch))
        k = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1)
    g = GreedyAttack_moco.__new__(GreedyAttack_moco)
    g.max_image_len = cfg["max_image_len"]
    g.criterion = nn.CrossEntropyLoss()                     # greedy_attack_vilt.py:501 (without .cuda)
    g.build_mini_vilt(h)
    ids, masks = batch["text_ids"], batch["text_masks"]
    loss, grads, q = g.get_grad(ids.clone(), masks.clone(), batch["text"], deepcopy(batch), torch.device("cpu"), k)
    out = {"k": k.numpy(), "loss": np.float64(loss.item()), "q": q.detach().numpy(),
           "saliency_l1": np.abs(grads).sum(-1), "grads_sub": grads[:, :, ::16].copy(),
           "grads_digest": tensor_digest(torch.from_numpy(grads))}
    # candidates: the same synthetic replacement rule the build's default candidate_fn uses, at the top-saliency
    # position of every sample (positions 1..sep-1)
    fn = O.synthetic_candidates(cfg.get("seed", 0), n_cand, cfg["vocab_size"])
    rows, all_num, pos = [], [], []
    sal = torch.from_numpy(out["saliency_l1"])
    for b in range(B):
        sep = int((ids[b] == 102).nonzero()[0])
        t = int(torch.argsort(sal[b, 1:sep], descending=True, stable=True)[0]) + 1
        pos.append(t)
        for c in fn(0, b, t, ids[b]):
            r = ids[b].clone()
            r[t] = c
            rows.append(r)
        all_num.append(n_cand)
    cids = torch.stack(rows)
    own = torch.arange(B).repeat_interleave(n_cand)
    cb = {"text_ids": cids, "text_masks": masks[own], "text_labels": batch["text_labels"][own], "image": [batch["image"][0][own]],
          "text": ["synthetic"] * len(rows)}
    all_loss = g.split_forward(cb, all_num, q.detach().clone(), k)
    out["cand_pos"] = np.array(pos)
    out["cand_ids"] = cids.numpy()
    out["cand_loss"] = np.array([[float(x) for x in cl] for cl, _ in all_loss])       # batch-mean CE with row i replaced
    out["cand_best_idx"] = np.array([j for _, j in all_loss])
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"], cfg["num_negative"], seed_k, n_cand])
    path = os.path.join(ROOT, "tests", "golden", f"txtatk_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "text attack loss", out["loss"], "best idx", out["cand_best_idx"], "bytes", os.path.getsize(path))


TOY_GROUPS = [
    "dog puppy hound canine pooch dogs", "cat kitten feline kitty tabby cats", "man guy gentleman fellow male men",
    "woman lady female gal madam women", "child kid toddler youngster infant children", "house home cottage dwelling cabin houses",
    "street road avenue lane boulevard streets", "car automobile vehicle sedan auto cars", "boat ship vessel yacht canoe boats",
    "field meadow pasture lawn prairie fields", "ball sphere orb globe football balls", "table desk counter bench stand tables",
    "big large huge giant enormous bigger", "small little tiny petite miniature smaller", "red crimson scarlet ruby maroon reddish",
    "green emerald olive lime jade greenish", "run jog sprint dash race running", "walk stroll hike march wander walking",
    "sit rest perch lounge squat sitting", "eat dine chew munch devour eating", "jump leap hop bound spring jumping",
    "hold grasp grip clutch carry holding", "look gaze stare glance peer looking", "play frolic romp sport game playing",
]
TOY_FUNCTION = "a an the on in at with of and is are near by to from under over two three some his her its".split()
TOY_STOP = "near two three some".split()                      # stands in for nltk's English stop words (not in the built-in list)
TOY_PIECES = "##s ##ing ##ed ##er ##ly".split()


def write_toy_resources():
    """Small WordPiece vocabulary (special ids where bert-base-uncased has them), synonym groups with counter-fitted-style
    vectors, stop words: the offline stand-ins for the resources the reference loads by name
    (greedy_attack_vilt.py:51,53,66-67).  Original data of this repo, written next to the fixtures."""
    gold = os.path.join(ROOT, "tests", "golden")
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"[unused{i}]" for i in range(99, 995)]
    assert len(toks) == 1000
    words = [w for grp in TOY_GROUPS for w in grp.split()[:5]]               # the 6th word of a group is an inflection built from pieces
    toks += TOY_FUNCTION + words + TOY_PIECES
    with open(os.path.join(gold, "toy_vocab.txt"), "w") as f:
        f.write("\n".join(toks) + "\n")
    with open(os.path.join(gold, "toy_stopwords.txt"), "w") as f:
        f.write("\n".join(TOY_STOP) + "\n")
    rng = np.random.RandomState(7)
    lines = []
    for gi, grp in enumerate(TOY_GROUPS):
        base = rng.randn(24)
        for wi, w in enumerate(grp.split()):
            v = base + (0.35 + 0.25 * wi) * rng.randn(24)                    # later words of a group drift out of the 0.5 threshold
            lines.append(w + " " + " ".join(f"{x:.5f}" for x in v))
    lines.append(lines[3])                                                   # one duplicated line: row numbering vs word numbering (:86-88)
    for w in ("the", "near", "is"):
        lines.append(w + " " + " ".join(f"{x:.5f}" for x in rng.randn(24)))
    with open(os.path.join(gold, "toy_counter_fitted.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return gold


TOY_SENTENCES = [
    "a big dog and a small cat run on the green field near the house",
    "the man and the woman walk on the street with a child holding a red ball",
    "two dogs jump over a table in the house",
    "a lady is sitting in a boat looking at the kid playing with some puppy near a car",
    "the kitten eat under the desk",
    "three men hold the large football on a lawn by the road and look at the ship",
]


def run_text_attack_words(tag, cfg, B, seed_w, seed_k, seed_b, max_loops, n_cand):
    """The WHOLE greedy text attack (host + tensor side) from the reference's own GreedyAttack_moco.adv_attack_samples
    (attack/greedy_attack_vilt.py:494-599) on toy resources.  The object is made with __new__ and given the attributes
    __init__ (:48-74) would set; init_matrix (:76-111) is the reference's.  Only adaptation: the installed transformers
    5.x tokenizer has no `_convert_token_to_id` (the reference calls it at :285; transformers 4.2.1 had it) - a subclass
    forwards it to convert_tokens_to_ids.  Run with PYTHONHASHSEED=0: the reference keeps candidates in Python sets."""
    import tempfile
    from transformers import BertTokenizer
    from torch.nn import CosineSimilarity

    class Tok(BertTokenizer):
        def _convert_token_to_id(self, token):
            return self.convert_tokens_to_ids(token)

    gold = write_toy_resources()
    with open(os.path.join(gold, "toy_vocab.txt")) as f:
        vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
    tok = Tok(vocab=vocab, do_lower_case=True)
    torch.manual_seed(99)
    cfg = dict(cfg, per_gpu_batchsize=B, max_loops=max_loops, n_candidates=n_cand)
    p = O.init_params(cfg, seed_w, k_seed=seed_k)
    h = Holder(cfg)
    h.load_oracle_params(p)
    h.proj_queue.copy_(O.init_queue(cfg, 0))
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b)
    sentences = TOY_SENTENCES[:B]
    enc = tok(sentences, truncation=True, padding="max_length", max_length=cfg["max_text_len"], return_special_tokens_mask=True)
    batch["text"] = list(sentences)
    batch["text_ids"] = torch.tensor(enc["input_ids"])
    batch["text_masks"] = torch.tensor(enc["attention_mask"])
    with torch.no_grad():
        rk = h.infer_k(deepcopy(batch))
        k = nn.functional.normalize(h.k_moco_head(rk["cls_feats"]), dim=1)

    g = GreedyAttack_moco.__new__(GreedyAttack_moco)
    g.pl_module, g.contrastive_framework = None, "moco"
    g.stopwords = set(TOY_STOP)
    g.cosine_similarity = CosineSimilarity(dim=1, eps=1e-6)
    g.tokenizer = tok
    g.device, g.words_to_sub_words = None, None
    g.max_length, g.n_candidates, g.max_loops, g.sim_thred = cfg["max_text_len"], n_cand, max_loops, 0.5
    g.word2id = tok.get_vocab()
    g.id2word = {v: kk for kk, v in g.word2id.items()}
    g.cos_sim = g.sim_word2id = g.sim_id2word = g.cos_sim_dict = None
    g.synonym = "cos_sim"
    g.max_image_len = cfg["max_image_len"]
    g.moco_head = None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:                       # init_matrix saves its cosine matrix into the cwd (:97)
        os.chdir(tmp)
        try:
            g.init_matrix(os.path.join(gold, "toy_counter_fitted.txt"), os.path.join(tmp, "absent.npy"))
        finally:
            os.chdir(cwd)

    trace = {"replace_idx": [], "all_new_text": [], "all_num": [], "best_idx": [], "cand_loss": [], "grads": [], "ids": []}
    cwi, cns, spf, gg = g.compute_word_importance, g.construct_new_samples, g.split_forward, g.get_grad

    def w_gg(*a, **kw):
        out = gg(*a, **kw)
        trace["grads"].append(np.array(out[1], dtype=np.float32))
        trace["ids"].append(a[0].numpy().copy())
        return out

    def w_cwi(**kw):
        out = cwi(**kw)
        trace["replace_idx"].append([-1 if x is None else int(x) for x in out[0]])
        return out

    def w_cns(**kw):
        out = cns(**kw)
        trace["all_new_text"].append(list(out[0]))
        trace["all_num"].append(list(out[1]))
        return out

    def w_spf(*a, **kw):
        out = spf(*a, **kw)
        trace["best_idx"].append([int(j) for _, j in out])
        trace["cand_loss"].append([float(x) for cl, _ in out for x in cl])
        return out

    g.get_grad, g.compute_word_importance, g.construct_new_samples, g.split_forward = w_gg, w_cwi, w_cns, w_spf
    ids_in, masks_in = batch["text_ids"].numpy().copy(), batch["text_masks"].numpy().copy()     # (get_grad rebinds the batch's entries)
    res = g.adv_attack_samples(h, batch, k)
    words = sorted(g.sim_word2id, key=g.sim_word2id.get)
    out = {
        "k": k.numpy(), "text_in": np.array(sentences), "text_ids_in": ids_in, "text_masks_in": masks_in,
        "text_out": np.array(res["text"]), "text_ids_out": res["txt_input_ids"].numpy(), "text_masks_out": res["text_masks"].numpy(),
        "num_changes": np.float64(res["num_changes"]), "change_rate": np.float64(res["change_rate"]),
        "problem": np.array(bool(res["Problem"])), "changes_verification": np.array(res["changes_verification"]),
        "replace_idx": np.array(trace["replace_idx"]), "best_idx": np.array(trace["best_idx"]),
        "grads_loop0": trace["grads"][0], "ids_loops": np.stack(trace["ids"]),
        "syn_words": np.array(words),
        # the reference's candidate sets in ITS iteration order (a set: depends on the hash seed), '|'-joined per word
        "syn_cands": np.array(["|".join(g.cos_sim_dict[g.sim_word2id[w]]) for w in words]),
        "meta": np.array([B, seed_w, seed_b, cfg["num_layers"], cfg["num_negative"], seed_k, n_cand, max_loops]),
    }
    for li in range(max_loops):
        out[f"new_text_{li}"] = np.array(trace["all_new_text"][li])
        out[f"all_num_{li}"] = np.array(trace["all_num"][li])
        out[f"cand_loss_{li}"] = np.array(trace["cand_loss"][li])
    path = os.path.join(ROOT, "tests", "golden", f"txtatk_words_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "word attack", res["text"], "changes", res["num_changes"], res["changes_verification"], "replace", trace["replace_idx"],
          "best", trace["best_idx"], "bytes", os.path.getsize(path))


def run_barlow(tag, cfg, B, seed_w, seed_h, seed_b, ragged, dims):
    """One Barlow-Twins image-view step from the reference's own compute_barlowtwins_contrastive (objectives.py:449-602),
    PGDAttack_bartlowtwins (pgd_attack_vilt.py:178-236) and heads.BarlowTwinsHead (heads.py:88-107), built with widths
    `dims` (vilt_module.py:115 hard-codes 8192: the class itself takes them as arguments), then backward.  The clean
    projection k, the attacked projection, the PGD delta (what pgd_attack returns) and BatchNorm's running statistics are
    recorded through forward hooks / a wrapper around the attacker - nothing in the reference is edited."""
    torch.manual_seed(777)
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    cfg = dict(cfg, per_gpu_batchsize=B, barlowtwins_dims=tuple(dims), image_view=True, text_view=False)
    p = O.init_params(cfg, seed_w)
    hp = O.bt_init_params(cfg, seed_h)
    h = Holder(cfg)
    h.barlowtwins_head = heads.BarlowTwinsHead(cfg["hidden_size"], [dims[0], dims[1]], dims[2])
    h.adv_lr = cfg["adv_lr"]
    h.pgd_attacker = PGDAttack_bartlowtwins(cfg)
    for name in ("train", "val"):
        for met in ("barlowtwins_loss", "barlowtwins_loss_invariance_img", "barlowtwins_loss_redundancy_img"):
            setattr(h, f"{name}_{met}", lambda x: x)
    h.load_oracle_params(dict(p, **hp))
    h.train()
    batch = O.synthetic_batch(cfg, B, seed_b, ragged_text=ragged)
    seen, deltas = [], []
    h.barlowtwins_head.register_forward_hook(lambda mod, inp, out_: seen.append((mod is h.barlowtwins_head, out_.detach().clone())))
    attack = h.pgd_attacker.pgd_attack

    def recording_attack(*a, **kw):
        d = attack(*a, **kw)
        deltas.append(d.detach().clone())
        return d

    h.pgd_attacker.pgd_attack = recording_attack
    h.zero_grad()
    ret = objectives.compute_barlowtwins_contrastive(h, deepcopy(batch))
    # vilt_module.py:475 sums EVERY returned value whose key contains "loss": barlowtwins_loss AND the two logged components
    # barlowtwins_loss_invariance_img / _redundancy_img, which are live graph tensors - the step optimises 2x the loss
    loss = sum(v for kk, v in ret.items() if "loss" in kk)
    loss.backward()
    own = [t for is_own, t in seen if is_own]                                # the module's own head: clean k, then the attacked view
    assert len(own) == 2 and len(deltas) == 1
    out = {"barlowtwins_loss": np.float64(ret["barlowtwins_loss"].item()), "total_loss": np.float64(loss.item()), "k": own[0].numpy(), "q_image": own[1].numpy(),
           "delta_sub": deltas[0][:, :, ::8, ::8].contiguous().numpy(), "delta_digest": tensor_digest(deltas[0]),
           "delta_patch00": deltas[0][:, :, :32, :32].contiguous().numpy()}
    for kk, v in ret.items():
        if kk != "barlowtwins_loss":
            out["ret_" + kk] = np.float64(float(v))
    for kk, v in h.logged.items():
        out["log_" + kk.replace("/", "__")] = np.float64(v)
    gnames, gd = [], []
    for n, prm in h.named_parameters():
        if not n.startswith("k_") and prm.grad is not None:
            gnames.append(n)
            gd.append(tensor_digest(prm.grad))
    out["grad_names"] = np.array(gnames)
    out["grad_digest"] = np.stack(gd)
    out["grad_bt_w1"] = h.barlowtwins_head.projector[0].weight.grad[:8, :64].numpy().copy()
    out["grad_bt_w3"] = h.barlowtwins_head.projector[6].weight.grad[:8, :64].numpy().copy()
    out["grad_bt_g2"] = h.barlowtwins_head.projector[4].weight.grad[:64].numpy().copy()
    out["grad_pooler_w"] = h.pooler.dense.weight.grad[:8, :64].numpy().copy()
    out["grad_qkv0_w"] = h.transformer.blocks[0].attn.qkv.weight.grad[:8, :64].numpy().copy()
    for n, buf in h.barlowtwins_head.named_buffers():
        out["buf_" + n.replace(".", "__")] = buf.detach().double().numpy() if buf.dtype != torch.int64 else np.int64(int(buf))
    out["meta"] = np.array([B, seed_w, seed_b, int(ragged), cfg["num_layers"], cfg["adv_steps_img"], seed_h, dims[0], dims[1], dims[2]])
    path = os.path.join(ROOT, "tests", "golden", f"barlow_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "barlowtwins_loss", out["barlowtwins_loss"], {k_: float(v) for k_, v in ret.items() if k_ != "barlowtwins_loss"},
          "bytes", os.path.getsize(path))


def run_barlow_views(tag, cfg, B, seed_w, seed_h, seed_b, dims, max_loops, n_cand):
    """The three-view Barlow-Twins step (text / image / both) from the reference's compute_barlowtwins_contrastive with its
    own GreedyAttack_barlowtwins (greedy_attack_vilt.py:602-830) on the toy linguistic resources and PGDAttack_bartlowtwins,
    then the backward of training_step's loss sum.  The attacker object is made with __new__ (its __init__ downloads a
    tokenizer / loads nltk) exactly like run_text_attack_words.  PYTHONHASHSEED=0 (candidate sets)."""
    import tempfile
    from transformers import BertTokenizer
    from torch.nn import CosineSimilarity
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)

    class Tok(BertTokenizer):
        def _convert_token_to_id(self, token):
            return self.convert_tokens_to_ids(token)

    gold = write_toy_resources()
    with open(os.path.join(gold, "toy_vocab.txt")) as f:
        vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
    tok = Tok(vocab=vocab, do_lower_case=True)
    torch.manual_seed(555)
    cfg = dict(cfg, per_gpu_batchsize=B, barlowtwins_dims=tuple(dims), image_view=True, text_view=True, max_loops=max_loops, n_candidates=n_cand)
    p = O.init_params(cfg, seed_w)
    hp = O.bt_init_params(cfg, seed_h)
    h = Holder(cfg)
    h.barlowtwins_head = heads.BarlowTwinsHead(cfg["hidden_size"], [dims[0], dims[1]], dims[2])
    h.adv_lr = cfg["adv_lr"]
    h.pgd_attacker = PGDAttack_bartlowtwins(cfg)
    for name in ("train", "val"):
        for met in ("barlowtwins_loss",) + tuple(f"barlowtwins_loss_{a}_{b}" for a in ("invariance", "redundancy") for b in ("img", "text", "both")):
            setattr(h, f"{name}_{met}", lambda x: x)
    h.load_oracle_params(dict(p, **hp))
    h.train()
    g = GreedyAttack_barlowtwins.__new__(GreedyAttack_barlowtwins)
    g.pl_module, g.contrastive_framework = None, "barlowtwins"
    g.stopwords = set(TOY_STOP)
    g.cosine_similarity = CosineSimilarity(dim=1, eps=1e-6)
    g.tokenizer = tok
    g.device, g.words_to_sub_words = None, None
    g.max_length, g.n_candidates, g.max_loops, g.sim_thred = cfg["max_text_len"], n_cand, max_loops, 0.5
    g.word2id = tok.get_vocab()
    g.id2word = {v: kk for kk, v in g.word2id.items()}
    g.cos_sim = g.sim_word2id = g.sim_id2word = g.cos_sim_dict = None
    g.synonym = "cos_sim"
    g.max_image_len = cfg["max_image_len"]
    g.barlowtwins_head = None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            g.init_matrix(os.path.join(gold, "toy_counter_fitted.txt"), os.path.join(tmp, "absent.npy"))
        finally:
            os.chdir(cwd)
    h.greedy_attacker = g
    batch = O.synthetic_batch(cfg, B, seed_b)
    sentences = TOY_SENTENCES[:B]
    enc = tok(sentences, truncation=True, padding="max_length", max_length=cfg["max_text_len"], return_special_tokens_mask=True)
    batch["text"] = list(sentences)
    batch["text_ids"] = torch.tensor(enc["input_ids"])
    batch["text_masks"] = torch.tensor(enc["attention_mask"])
    # keys the reference's Barlow-Twins text attack copies per candidate (:752-760); their values never reach the arithmetic
    batch["cap_index"] = batch["replica"] = batch["img_index"] = batch["raw_index"] = list(range(B))
    batch["text_ids_mlm"] = batch["text_ids"].clone()
    batch["text_labels_mlm"] = batch["text_labels"].clone()
    ids_in, masks_in = batch["text_ids"].numpy().copy(), batch["text_masks"].numpy().copy()
    trace = {"replace_idx": [], "best_idx": [], "all_new_text": []}
    cwi, spf, cns = g.compute_word_importance, g.split_forward, g.construct_new_samples

    def w_cwi(**kw):
        o_ = cwi(**kw)
        trace["replace_idx"].append([-1 if x is None else int(x) for x in o_[0]])
        return o_

    def w_spf(*a, **kw):
        o_ = spf(*a, **kw)
        trace["best_idx"].append([int(j) for _, j in o_])
        return o_

    def w_cns(**kw):
        o_ = cns(**kw)
        trace["all_new_text"].append(list(o_[0]))
        return o_

    g.compute_word_importance, g.split_forward, g.construct_new_samples = w_cwi, w_spf, w_cns
    geo = objectives.compute_geometric
    attacked = {}

    def rec_geo(pl_module, b_, name, k_modality=None):
        out_ = geo(pl_module, b_, name, k_modality=k_modality)
        attacked["text"], attacked["ids"], attacked["masks"] = list(out_["text"]), out_["text_ids"].clone(), out_["text_masks"].clone()
        return out_

    objectives.compute_geometric = rec_geo
    try:
        h.zero_grad()
        ret = objectives.compute_barlowtwins_contrastive(h, deepcopy(batch))
    finally:
        objectives.compute_geometric = geo
    loss = sum(v for kk, v in ret.items() if "loss" in kk)                        # vilt_module.py:475
    loss.backward()
    words = sorted(g.sim_word2id, key=g.sim_word2id.get)
    out = {"total_loss": np.float64(loss.item()), "text_in": np.array(sentences), "text_ids_in": ids_in, "text_masks_in": masks_in,
           "text_out": np.array(attacked["text"]), "text_ids_out": attacked["ids"].numpy(), "text_masks_out": attacked["masks"].numpy(),
           "replace_idx": np.array(trace["replace_idx"]), "best_idx": np.array(trace["best_idx"]),
           "syn_words": np.array(words), "syn_cands": np.array(["|".join(g.cos_sim_dict[g.sim_word2id[w]]) for w in words])}
    for li, t in enumerate(trace["all_new_text"]):
        out[f"new_text_{li}"] = np.array(t)
    for kk, v in ret.items():
        out["ret_" + kk] = np.float64(float(v))
    for kk, v in h.logged.items():
        out["log_" + kk.replace("/", "__")] = np.float64(v)
    gnames, gd = [], []
    for n, prm in h.named_parameters():
        if not n.startswith("k_") and prm.grad is not None:
            gnames.append(n)
            gd.append(tensor_digest(prm.grad))
    out["grad_names"] = np.array(gnames)
    out["grad_digest"] = np.stack(gd)
    for n, buf in h.barlowtwins_head.named_buffers():
        out["buf_" + n.replace(".", "__")] = buf.detach().double().numpy() if buf.dtype != torch.int64 else np.int64(int(buf))
    out["meta"] = np.array([B, seed_w, seed_b, cfg["num_layers"], cfg["adv_steps_img"], seed_h, dims[0], dims[1], dims[2], max_loops, n_cand])
    path = os.path.join(ROOT, "tests", "golden", f"barlow3_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "three views total", out["total_loss"], attacked["text"], "replace", trace["replace_idx"], "best", trace["best_idx"],
          "bytes", os.path.getsize(path))


def run_pipeline():
    """Input-pipeline pieces (row f3) from the reference's own MinMaxResize (vilt/transforms/utils.py:5-26) and
    BaseDataset.collate (vilt/datasets/base_dataset.py:167-245; an unbound call - the method never touches self)."""
    import importlib.util
    from PIL import Image
    spec = importlib.util.spec_from_file_location("ref_transforms_utils", os.path.join(REF, "vilt", "transforms", "utils.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    rng = np.random.RandomState(5)
    sizes = [(640, 480), (480, 640), (500, 375), (333, 500), (1024, 200), (200, 1024), (384, 384), (50, 60), (799, 801), (2000, 1500)]
    sizes += [(int(rng.randint(200, 1500)), int(rng.randint(200, 1500))) for _ in range(40)]
    out = {"sizes_in": np.array(sizes)}
    for shorter, longer in ((384, 640), (800, 1333), (224, 373)):
        r = tu.MinMaxResize(shorter=shorter, longer=longer)
        res = []
        for w, h in sizes:
            img = Image.new("RGB", (w, h))
            res.append(r(img).size)                             # (new_w, new_h)
        out[f"sizes_out_{shorter}_{longer}"] = np.array(res)
    # pixels: one synthetic image through MinMaxResize(384, 640) -> ToTensor -> Normalize(.5, .5)
    src = (rng.rand(300, 451, 3) * 255).astype(np.uint8)
    img = tu.MinMaxResize(shorter=384, longer=640)(Image.fromarray(src))
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255.0)
    t = (t - 0.5) / 0.5                                         # transforms.Normalize(mean .5, std .5) (pixelbert.py:9-17)
    out["pix_src"] = src
    out["pix_out_shape"] = np.array(t.shape)
    out["pix_out_digest"] = tensor_digest(t)
    out["pix_out_sub"] = t[:, ::16, ::16].numpy()
    # collate
    sys.modules["vilt.transforms"] = types.ModuleType("vilt.transforms")
    sys.modules["vilt.transforms"].keys_to_transforms = lambda keys, size=224: []
    spec = importlib.util.spec_from_file_location("ref_base_dataset", os.path.join(REF, "vilt", "datasets", "base_dataset.py"))
    bd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bd)
    g = torch.Generator().manual_seed(9)
    shapes = [(3, 384, 352), (3, 320, 384), (3, 224, 288)]
    lens = [7, 40, 13]
    batch = []
    for (c, h, w), n in zip(shapes, lens):
        ids = torch.randint(1000, 30000, (n,), generator=g).tolist()
        batch.append({"image": [torch.rand(c, h, w, generator=g) * 2 - 1], "false_image_0": [torch.rand(c, h, w, generator=g) * 2 - 1],
                      "text": ("caption %d" % n, {"input_ids": ids, "attention_mask": [1] * n}), "img_index": n, "cap_index": 0, "raw_index": n})

    def stub_collator(encodings):                               # stands in for DataCollatorForLanguageModeling (no masking)
        ids = torch.zeros(len(encodings), 40, dtype=torch.int64)
        for i, e in enumerate(encodings):
            ids[i, : len(e["input_ids"])] = torch.tensor(e["input_ids"])
        return {"input_ids": ids, "labels": torch.full_like(ids, -100)}

    d = bd.BaseDataset.collate(None, batch, stub_collator)
    out["collate_image"] = d["image"][0].numpy()[:, :, ::8, ::8]
    out["collate_image_digest"] = tensor_digest(d["image"][0])
    out["collate_false_image_digest"] = tensor_digest(d["false_image_0"][0])
    out["collate_text_ids"] = d["text_ids"].numpy()
    out["collate_text_masks"] = d["text_masks"].numpy()
    out["collate_text_labels"] = d["text_labels"].numpy()
    out["collate_keys"] = np.array(sorted(d.keys()))
    path = os.path.join(ROOT, "tests", "golden", "pipeline.npz")
    np.savez_compressed(path, **out)
    print("pipeline fixture", os.path.getsize(path), "bytes;", out["sizes_out_384_640"][:4].tolist())


def run_dataset():
    """The reference's own BaseDataset (vilt/datasets/base_dataset.py:11-165) on a toy arrow shard written by this script
    (tests/golden/toy_shard.arrow: 6 PNG images of different sizes, 1-3 captions each out of the toy vocabulary): index_mapper,
    corpus, and get_suite (image through the reference's MinMaxResize + ToTensor/Normalize arithmetic, tokenised text, replica
    flag, false image / false text draws under random.seed) -> tests/golden/dataset.npz."""
    import importlib.util
    import io
    import random
    from PIL import Image
    import pyarrow as pa
    sys.path.insert(0, ROOT)
    import rmcl_pkg  # noqa: F401
    from rmcl_amd.attack import word_substitution as WS
    from rmcl_amd.vilt.datasets import write_arrow_table
    gold = os.path.join(ROOT, "tests", "golden")
    rng = np.random.RandomState(7)
    sizes = [(96, 64), (64, 96), (128, 80), (50, 70), (200, 120), (64, 64)]                       # (w, h)
    images = []
    for w, h in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        a = (120 + 70 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 7.0)[..., None] + rng.normal(0, 15, (h, w, 3))).clip(0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, "PNG")
        images.append(buf.getvalue())
    captions = [["a dog near the house", "two cat on the street"], ["the man by a car"], ["a child in the home", "a child in the home", "her kitten under the cabin"],
                ["some woman at the road"], ["his puppy over the lane", "the lady with a automobile"], ["three kid from the cottage"]]
    shard = os.path.join(gold, "toy_shard.arrow")
    write_arrow_table(shard, images, captions)
    # the reference module with stand-ins for what it imports at the top (torchvision; its own vilt.transforms -> torchvision)
    spec = importlib.util.spec_from_file_location("ref_transforms_utils2", os.path.join(REF, "vilt", "transforms", "utils.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)

    def ref_transform(size):
        r = tu.MinMaxResize(shorter=size, longer=int((1333 / 800) * size))
        def f(img):
            t = torch.from_numpy(np.asarray(r(img)).copy()).permute(2, 0, 1).float().div(255.0)      # ToTensor
            return (t - 0.5) / 0.5                                                                 # Normalize(.5, .5)
        return f
    sys.modules["vilt.transforms"] = types.ModuleType("vilt.transforms")
    sys.modules["vilt.transforms"].keys_to_transforms = lambda keys, size=224: [ref_transform(size) for _ in keys]
    spec = importlib.util.spec_from_file_location("ref_base_dataset2", os.path.join(REF, "vilt", "datasets", "base_dataset.py"))
    bd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bd)
    _concat = pa.concat_tables                                  # (pyarrow 25 spells promote=True promote_options="default")
    pa.concat_tables = lambda tables, promote=False, **k: _concat(tables, promote_options="default" if promote else "none", **k)
    tok = WS.load_tokenizer(os.path.join(gold, "toy_vocab.txt"))
    out = {}
    for tag, kw in (("dup", dict(remove_duplicate=False)), ("imgonly", dict(remove_duplicate=False, image_only=True)), ("max4", dict(remove_duplicate=False, max_num=4))):
        ds = bd.BaseDataset(gold, ["pixelbert"], 96, ["toy_shard"], text_column_name="caption", draw_false_image=1, draw_false_text=1, **kw)
        ds.tokenizer = tok
        out[f"{tag}_len"] = np.array(len(ds))
        out[f"{tag}_index_mapper"] = np.array([[ds.index_mapper[j][0], -1 if ds.index_mapper[j][1] is None else ds.index_mapper[j][1]] for j in range(len(ds))])
        if tag != "dup":
            continue
        out["corpus"] = np.array(ds.corpus)
        random.seed(123)
        for j in (0, 1, 4, 7):
            r = ds.get_suite(j)
            out[f"s{j}_meta"] = np.array([r["img_index"], r["cap_index"], r["raw_index"], int(r["replica"])])
            out[f"s{j}_text"] = np.array(r["text"][0])
            out[f"s{j}_ids"] = np.array(r["text"][1]["input_ids"])
            out[f"s{j}_mask"] = np.array(r["text"][1]["attention_mask"])
            out[f"s{j}_image"] = r["image"][0].numpy()
            out[f"s{j}_false_image_digest"] = tensor_digest(r["false_image_0"][0])
            out[f"s{j}_false_text"] = np.array(r["false_text_0"][0])
            out[f"s{j}_false_ids"] = np.array(r["false_text_0"][1]["input_ids"])
    ds = bd.BaseDataset(gold, ["pixelbert"], 96, ["toy_shard"], text_column_name="caption")        # default: captions de-duplicated per image
    out["dedup_len"] = np.array(len(ds))
    out["dedup_counts"] = np.array([len(t) for t in ds.all_texts])
    pa.concat_tables = _concat
    path = os.path.join(gold, "dataset.npz")
    np.savez_compressed(path, **out)
    print("dataset fixture", os.path.getsize(path), "bytes; shard", os.path.getsize(shard), "bytes; len", int(out["dup_len"]), int(out["dedup_len"]))


def run_schedules():
    """LR curves from transformers.optimization (the functions vilt_utils.py:404-432 calls; importable here).
    HF AdamW itself (vilt_utils.py:395-398, transformers==4.2.1) no longer exists in the installed transformers:
    the optimizer stays parity-unpinned (restated from the 4.2.1 source), only the schedules are pinned."""
    from transformers.optimization import get_polynomial_decay_schedule_with_warmup, get_cosine_schedule_with_warmup
    out = {}
    for name, (base, warm, total, end_lr, power) in {"poly1": (1e-4, 10, 100, 0.0, 1.0), "poly2": (3e-4, 25, 250, 1e-6, 2.0),
                                                      "poly_nowarm": (1e-4, 0, 40, 0.0, 1.0)}.items():
        w = nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=base)
        sch = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total, lr_end=end_lr, power=power)
        lrs = []
        for _ in range(total + 20):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[name] = np.array(lrs)
        out[name + "_args"] = np.array([base, warm, total, end_lr, power])
    for name, (base, warm, total) in {"cos1": (1e-4, 10, 100), "cos2": (2e-4, 0, 64)}.items():
        w = nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=base)
        sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
        lrs = []
        for _ in range(total + 20):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[name] = np.array(lrs)
        out[name + "_args"] = np.array([base, warm, total])
    path = os.path.join(ROOT, "tests", "golden", "schedules.npz")
    np.savez_compressed(path, **out)
    print("schedules", {k: v.shape for k, v in out.items() if not k.endswith("_args")})


if __name__ == "__main__":
    torch.set_num_threads(8)
    small = O.default_config(num_layers=2, num_negative=1024)
    full = O.default_config()
    only = set(sys.argv[1:])
    want = lambda n: not only or n in only
    if want("moco"):
        run_moco("L2_B4_ragged", small, 4, 11, 21, True)
        run_moco("L12_B2", full, 2, 12, 22, False)
    if want("ragged"):
        run_moco("L2_B4_raggedimg", small, 4, 11, 21, True, sizes=RAGGED_SIZES)
        # no full-size image in the batch: n = 132 selected patches (N = 173 tokens != 185), position table still 12 x 12
        run_moco("L2_B3_raggedimg2", small, 3, 11, 23, True, sizes=[(384, 352), (320, 384), (224, 288)])
    if want("itm"):
        run_itm("L2_B4_ragged", small, 4, 11, 21, True)
        run_itm("L12_B2", full, 2, 12, 22, False)
    if want("moco2"):
        run_moco_two_step("L2_B4_ragged", dict(small, adv_steps_img=2, momentum=0.9), 4, 11, 31, 21, True, 12, NUDGE)
        run_moco_two_step("L12_B2", dict(full, momentum=0.95), 2, 12, 32, 22, False, 65536 - 4, NUDGE)     # second enqueue wraps the pointer to 0
    if want("cleanitm"):
        run_clean_itm("L2_B4_ragged", dict(small, momentum=0.9), 4, 11, 31, 21, True)
        run_clean_itm("L12_B2", dict(full, momentum=0.95), 2, 12, 32, 22, False)
    if want("dataset"):
        run_dataset()
    if want("txtatk"):
        run_text_attack("L2_B4_ragged", small, 4, 11, 31, 21, True, 5)
    if want("txtwords"):
        if os.environ.get("PYTHONHASHSEED") != "0":
            sys.exit("txtwords: run with PYTHONHASHSEED=0 (the reference iterates Python sets of candidate words)")
        run_text_attack_words("L2_B4", small, 4, 11, 31, 21, 4, 5)
    if want("barlow"):
        run_barlow("L2_B4_ragged", dict(small, adv_steps_img=2), 4, 11, 41, 21, True, (512, 384, 256))
        run_barlow("L2_B4_wide", dict(small, adv_steps_img=1), 4, 13, 42, 23, False, (8192, 8192, 8192))   # the reference's widths
    if want("barlow3"):
        if os.environ.get("PYTHONHASHSEED") != "0":
            sys.exit("barlow3: run with PYTHONHASHSEED=0 (the reference iterates Python sets of candidate words)")
        run_barlow_views("L2_B4", dict(small, adv_steps_img=2), 4, 11, 41, 21, (512, 384, 256), 3, 5)
    if want("sched"):
        run_schedules()
    if want("pipeline"):
        run_pipeline()
