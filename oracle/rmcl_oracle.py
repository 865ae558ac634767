"""CPU oracle for the RMCL hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``rmcl_amd``) never imports anything under ``oracle/`` and fails
loudly when the HIP library is missing.

It is a restatement, in plain PyTorch CPU tensor math (float32, or float64 when
``dtype=torch.float64`` is passed), of the reference's algorithm for the path named by
BASELINE.json.  Every function cites the reference file:line it follows (paths relative
to the reference checkout).  Parity of this oracle with the reference itself is pinned
by ``tests/golden/*.npz`` which ``oracle/gen_golden.py`` produced by importing the
reference's own modules in the build container (see DESIGN.md "Oracle").

Parameters live in a flat ``dict[str, Tensor]`` keyed by the reference's state-dict
names, e.g. ``transformer.blocks.3.attn.qkv.weight`` and the momentum
copies ``k_transformer...``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# configuration defaults (vilt/config.py:24-116 and task_moco :128-164)
# --------------------------------------------------------------------------------------

def default_config(**over) -> dict:
    cfg = dict(
        vocab_size=30522, hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4,
        max_text_len=40, drop_rate=0.0, image_size=384, patch_size=32, max_image_len=200,
        num_negative=65536, momentum=0.999, temperature=0.07,
        text_view=False, image_view=True, augmentation=False,
        adv_steps_img=3, adv_lr_img=0.05, adv_max_norm_img=0.005,
        num_gpus=1, num_nodes=1, per_gpu_batchsize=64, proj_dim=128, n_candidates=5, max_loops=10, seed=0,
        loss_names={"moco": 1, "itm": 0}, adv_lr=0.0051,
    )
    cfg.update(over)
    return cfg


# groups that receive the momentum (EMA) update, in the reference's call order
# (vilt/modules/objectives.py:257-260).  The pooler is NOT copied (vilt_module.py:405).
EMA_GROUPS = ("text_embeddings", "token_type_embeddings", "transformer", "moco_head")


def param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """Query-side parameter names and shapes in the reference's registration order
    (vilt_module.py:26-85; vision_transformer.py:470-506; heads.py:10-20,129-143,173-180)."""
    D = cfg["hidden_size"]
    P = cfg["patch_size"]
    G = cfg["image_size"] // P
    Hm = D * cfg["mlp_ratio"]
    s: List[Tuple[str, Tuple[int, ...]]] = [
        ("text_embeddings.word_embeddings.weight", (cfg["vocab_size"], D)),
        ("text_embeddings.position_embeddings.weight", (cfg["max_text_len"], D)),
        ("text_embeddings.token_type_embeddings.weight", (2, D)),
        ("text_embeddings.LayerNorm.weight", (D,)),
        ("text_embeddings.LayerNorm.bias", (D,)),
        ("token_type_embeddings.weight", (2, D)),
        ("transformer.cls_token", (1, 1, D)),
        ("transformer.pos_embed", (1, G * G + 1, D)),
        ("transformer.patch_embed.proj.weight", (D, 3, P, P)),
        ("transformer.patch_embed.proj.bias", (D,)),
    ]
    for i in range(cfg["num_layers"]):
        b = f"transformer.blocks.{i}."
        s += [
            (b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)),
            (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.qkv.bias", (3 * D,)),
            (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
            (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)),
            (b + "mlp.fc1.weight", (Hm, D)), (b + "mlp.fc1.bias", (Hm,)),
            (b + "mlp.fc2.weight", (D, Hm)), (b + "mlp.fc2.bias", (D,)),
        ]
    s += [
        ("transformer.norm.weight", (D,)), ("transformer.norm.bias", (D,)),
        ("moco_head.projector.0.weight", (D, D)), ("moco_head.projector.0.bias", (D,)),
        ("moco_head.projector.1.weight", (D,)), ("moco_head.projector.1.bias", (D,)),
        ("moco_head.projector.3.weight", (cfg["proj_dim"], D)),
        ("pooler.dense.weight", (D, D)), ("pooler.dense.bias", (D,)),
        ("itm_score.fc.weight", (2, D)), ("itm_score.fc.bias", (2,)),
    ]
    return s


def init_params(cfg: dict, seed: int, dtype=torch.float32, k_seed: Optional[int] = None, k_scale: float = 0.01) -> Params:
    """Seeded, machine-independent synthetic weights (CPU generator).

    NOT the reference's initialiser (objectives.py:1505-1516 zeroes every bias and sets
    LayerNorm to (1,0), which would hide bias/affine bugs): weights ~ N(0, 0.02) like the
    reference, but biases ~ N(0, 0.02) and LayerNorm weights ~ 1 + N(0, 0.05).  The golden
    generator loads exactly these tensors into the reference modules, so fixtures are
    reproducible from (cfg, seed) alone.  Momentum copies start equal to the query
    weights (vilt_module.py:270-273) unless ``k_seed`` is given: then k_* = q + k_scale * N(0,1) from
    its own generator - the NORMAL case in the reference, whose k_* modules are shadowed before the
    strict=False checkpoint load (vilt_module.py:75-85 vs :135-138) and then trail q by the EMA."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    p: Params = {}
    for name, shape in param_shapes(cfg):
        t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith(
            "norm.weight") or name.endswith("LayerNorm.weight") or name.endswith("projector.1.weight"):
            t = 1.0 + 2.5 * t
        p[name] = t.to(dtype)
    gk = None
    if k_seed is not None:
        gk = torch.Generator(device="cpu")
        gk.manual_seed(k_seed)
    for name in list(p.keys()):
        if name.split(".")[0] in EMA_GROUPS:
            kt = p[name].clone()
            if gk is not None:
                kt = kt + (k_scale * torch.randn(kt.shape, generator=gk, dtype=torch.float32)).to(dtype)
            p["k_" + name] = kt
    return p


def init_queue(cfg: dict, seed: int = 0, dtype=torch.float32) -> Tensor:
    """Un-normalised randn queue (vilt_module.py:92-94; the normalise line is commented out)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(cfg["proj_dim"], cfg["num_negative"], generator=g, dtype=torch.float32).to(dtype)


def synthetic_batch(cfg: dict, B: int, seed: int, ragged_text: bool = False, dtype=torch.float32, sizes=None) -> dict:
    """Synthetic batch in the layout of BaseDataset.collate (vilt/datasets/base_dataset.py:167-245):
    image = list of views, view 0 = [B,3,H,W] in [-1,1] (transforms/utils.py:48-50)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    S = cfg["image_size"]
    L = cfg["max_text_len"]
    img = (torch.rand(B, 3, S, S, generator=g, dtype=torch.float32) * 2 - 1).to(dtype)
    ids = torch.randint(1000, cfg["vocab_size"], (B, L), generator=g, dtype=torch.int64)
    masks = torch.ones(B, L, dtype=torch.int64)
    if ragged_text:
        lens = torch.randint(8, L + 1, (B,), generator=g)
        for b in range(B):
            n = int(lens[b])
            masks[b, n:] = 0
            ids[b, n:] = 0
            ids[b, n - 1] = 102
    else:
        ids[:, -1] = 102
    ids[:, 0] = 101
    if sizes is not None:
        # zero-padded batch of smaller images (BaseDataset.collate pads bottom/right with zeros, base_dataset.py:192-206;
        # MinMaxResize makes every side a multiple of 32, transforms/utils.py:5-26): sample b keeps img[:, :h, :w]
        for b, (h, w) in enumerate(sizes):
            img[b, :, h:, :] = 0
            img[b, :, :, w:] = 0
        hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
        img = img[:, :, :hm, :wm].contiguous()
    false_img = torch.roll(img, shifts=1, dims=0)
    return {
        "image": [img], "false_image_0": [false_img],
        "text": ["synthetic"] * B,
        "text_ids": ids, "text_masks": masks,
        "text_labels": torch.full((B, L), -100, dtype=torch.int64),
    }


# --------------------------------------------------------------------------------------
# model ops
# --------------------------------------------------------------------------------------

def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() exact form (vision_transformer.py:268,275)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def text_embed(p: Params, pre: str, ids: Tensor, word_embeds: Optional[Tensor] = None) -> Tensor:
    """HF BertEmbeddings (vilt_module.py:26-38,293): LN_{1e-12}(word[id] + type[0] + pos[0:L]).  word_embeddings is
    nn.Embedding(vocab, hidden, padding_idx=pad_token_id = 0) (transformers 4.2.1 modeling_bert.py BertEmbeddings.__init__): the [PAD]
    row never receives a gradient, whatever the loss reads."""
    L = ids.shape[1]
    if word_embeds is None:
        w = p[pre + "text_embeddings.word_embeddings.weight"]
        we = torch.nn.functional.embedding(ids, w, padding_idx=0)
    else:
        we = word_embeds
    e = (we
         + p[pre + "text_embeddings.token_type_embeddings.weight"][0]
         + p[pre + "text_embeddings.position_embeddings.weight"][:L][None])
    return layer_norm(e, p[pre + "text_embeddings.LayerNorm.weight"],
                      p[pre + "text_embeddings.LayerNorm.bias"], 1e-12)


def patchify(img: Tensor, P: int) -> Tensor:
    """[B,3,H,W] -> [B, (H/P)*(W/P), 3*P*P], K ordered (c, ky, kx), patches row-major over
    the grid: the GEMM view of Conv2d(3,D,P,P) (vision_transformer.py:397-409,589)."""
    B, C, H, W = img.shape
    x = img.reshape(B, C, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // P) * (W // P), C * P * P)


def patch_mask(img: Tensor, P: int) -> Tensor:
    """Pixel mask -> patch mask (vision_transformer.py:564-565): (sum_c img != 0) sampled by
    nearest-neighbour resize, i.e. at the top-left pixel of every patch."""
    m = (img.sum(dim=1) != 0)
    return m[:, ::P, ::P].reshape(img.shape[0], -1).to(torch.int64)


def visual_embed_dense(p: Params, pre: str, img: Tensor, cfg: dict, drop_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Dense fixed-order visual_embed (vision_transformer.py:559-677) for full-size images
    (every patch valid, G*G <= max_image_len).  The reference permutes patch order with a
    CPU multinomial (:633-636); the encoder is permutation-equivariant so cls/logits are
    unchanged up to rounding."""
    P = cfg["patch_size"]
    W = p[pre + "transformer.patch_embed.proj.weight"]
    D = W.shape[0]
    x = patchify(img, P) @ W.reshape(D, -1).t() + p[pre + "transformer.patch_embed.proj.bias"]
    B = img.shape[0]
    cls = p[pre + "transformer.cls_token"].expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1) + p[pre + "transformer.pos_embed"]
    if drop_mask is not None:                              # pos_drop (vision_transformer.py:667)
        x = x * drop_mask
    m = torch.cat([torch.ones(B, 1, dtype=torch.int64), patch_mask(img, P)], dim=1)
    return x, m


def ragged_geometry(img: Tensor, P: int):
    """Per-sample valid-patch grid of a zero-padded batch (vision_transformer.py:563-567): pixel mask (sum_c != 0) sampled
    by nearest-neighbour resize = the top-left pixel of every patch; x_h / x_w = valid rows in column 0 / columns in row 0."""
    m = (img.sum(dim=1) != 0)[:, ::P, ::P].long()                 # [B, Gh, Gw]
    return m, m[:, :, 0].sum(dim=1), m[:, 0, :].sum(dim=1)


def ragged_select(mask: Tensor, max_image_len: int, select: Optional[Tensor] = None) -> Tuple[Tensor, int]:
    """Patch selection of visual_embed (vision_transformer.py:605-651): per sample the valid patches in row-major order, padded
    to n = min(max valid count, max_image_len) with non-valid patches.  The reference draws the pads (and, when a sample has
    MORE than n valid patches, the kept subset) with torch.multinomial; all non-valid patches of a zero-padded batch are
    identical (conv(0) + bias, zero position embedding, masked as keys), so the pad choice is immaterial: this restatement
    pads with the first non-valid patch.  A sample with more than n valid patches needs the reference's draw: pass it as
    ``select`` [B, n] (flat patch indices), e.g. captured from the reference's returned patch_index."""
    B = mask.shape[0]
    flat = mask.flatten(1)
    counts = flat.sum(1)
    n = int(counts.max())
    if isinstance(max_image_len, int) and max_image_len > 0:
        n = min(n, max_image_len)
    if select is not None:
        assert select.shape == (B, n)
        return select, n
    out = torch.zeros(B, n, dtype=torch.int64)
    for b in range(B):
        v = flat[b].nonzero().flatten()
        if v.numel() > n:
            raise ValueError("sample has more valid patches than max_image_len: pass the reference's selection")
        nv = (1 - flat[b]).nonzero().flatten()
        out[b, : v.numel()] = v
        if v.numel() < n:
            out[b, v.numel():] = nv[0]
    return out, n


def resize_pos_embed(pos: Tensor, G0: int, h: int, w: int) -> Tensor:
    """Bilinear, align_corners=True resize of the [G0*G0, D] spatial position table to [h, w, D] (vision_transformer.py:570-583)."""
    D = pos.shape[-1]
    sp = pos.t().reshape(1, D, G0, G0)
    r = F.interpolate(sp, size=(h, w), mode="bilinear", align_corners=True)
    return r[0].permute(1, 2, 0)


def visual_embed(p: Params, pre: str, img: Tensor, cfg: dict, select: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """VisionTransformer.visual_embed (vision_transformer.py:559-677) for any zero-padded batch: patch projection, per-sample
    position-embedding resize, valid-patch selection / padding, cls token.  Returns (x [B, 1+n, D], mask [B, 1+n], select)."""
    P = cfg["patch_size"]
    G0 = cfg["image_size"] // P
    W = p[pre + "transformer.patch_embed.proj.weight"]
    D = W.shape[0]
    B = img.shape[0]
    Gw = img.shape[3] // P
    m, xh, xw = ragged_geometry(img, P)
    sel, n = ragged_select(m, cfg.get("max_image_len", -1), select)
    xp = patchify(img, P) @ W.reshape(D, -1).t() + p[pre + "transformer.patch_embed.proj.bias"]      # [B, Gh*Gw, D]
    pos_tab = p[pre + "transformer.pos_embed"][0, 1:]
    rows = []
    masks = []
    for b in range(B):
        h, w = int(xh[b]), int(xw[b])
        pe = torch.zeros(img.shape[2] // P, Gw, D, dtype=xp.dtype)
        pe[:h, :w] = resize_pos_embed(pos_tab, G0, h, w)
        pe = pe.reshape(-1, D)
        rows.append(xp[b, sel[b]] + pe[sel[b]])
        masks.append(m[b].flatten()[sel[b]])
    x = torch.stack(rows)
    cls = (p[pre + "transformer.cls_token"][0] + p[pre + "transformer.pos_embed"][0, :1]).expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1)
    mask = torch.cat([torch.ones(B, 1, dtype=torch.int64), torch.stack(masks)], dim=1)
    return x, mask, sel



def attention(p: Params, b: str, x: Tensor, mask: Tensor, H: int) -> Tensor:
    """Attention.forward (vision_transformer.py:309-332): qkv features ordered (which, head, d)."""
    B, N, C = x.shape
    d = C // H
    qkv = x @ p[b + "attn.qkv.weight"].t() + p[b + "attn.qkv.bias"]
    qkv = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return o @ p[b + "attn.proj.weight"].t() + p[b + "attn.proj.bias"]


def block(p: Params, b: str, x: Tensor, mask: Tensor, H: int, drop: Optional[dict] = None) -> Tensor:
    """Block.forward (vision_transformer.py:371-375), drop_path=0, eps 1e-6 (:466).  ``drop`` (optional):
    explicit dropout scale masks {"proj", "hidden", "fc2"} for proj_drop (:331) and the two Mlp drops (:282,:284)."""
    d = drop or {}
    a = attention(p, b, layer_norm(x, p[b + "norm1.weight"], p[b + "norm1.bias"], 1e-6), mask, H)
    x = x + (a * d["proj"] if "proj" in d else a)
    h = layer_norm(x, p[b + "norm2.weight"], p[b + "norm2.bias"], 1e-6)
    h = gelu_erf(h @ p[b + "mlp.fc1.weight"].t() + p[b + "mlp.fc1.bias"])
    if "hidden" in d:
        h = h * d["hidden"]
    o = h @ p[b + "mlp.fc2.weight"].t() + p[b + "mlp.fc2.bias"]
    return x + (o * d["fc2"] if "fc2" in d else o)


def infer(p: Params, cfg: dict, ids: Tensor, text_masks: Tensor, img: Tensor, key: bool = False,
          image_token_type_idx: int = 1, word_embeds: Optional[Tensor] = None, drop: Optional[dict] = None,
          select: Optional[Tensor] = None) -> dict:
    """ViLTransformerSS.infer (vilt_module.py:275-351) / infer_k (:353-418).
    ``key=True`` uses the k_* momentum copies but the *query* pooler (:405)."""
    pre = "k_" if key else ""
    drop = drop or {}
    te = text_embed(p, pre, ids, word_embeds)
    if "text" in drop:                                     # BertEmbeddings.dropout
        te = te * drop["text"]
    te = te + p[pre + "token_type_embeddings.weight"][0]
    S = cfg["image_size"]
    if img.shape[2] == S and img.shape[3] == S and select is None and bool((patch_mask(img, cfg["patch_size"]) == 1).all()):
        ie, im = visual_embed_dense(p, pre, img, cfg, drop.get("image"))
    else:                                                   # zero-padded batch of smaller images
        assert "image" not in drop
        ie, im, _ = visual_embed(p, pre, img, cfg, select)
    ie = ie + p[pre + "token_type_embeddings.weight"][image_token_type_idx]
    x = torch.cat([te, ie], dim=1)
    m = torch.cat([text_masks, im], dim=1)
    for i in range(cfg["num_layers"]):
        x = block(p, f"{pre}transformer.blocks.{i}.", x, m, cfg["num_heads"], drop.get(i))
    x = layer_norm(x, p[pre + "transformer.norm.weight"], p[pre + "transformer.norm.bias"], 1e-6)
    L = ids.shape[1]
    cls = torch.tanh(x[:, 0] @ p["pooler.dense.weight"].t() + p["pooler.dense.bias"])  # heads.py:16-20
    return {"text_feats": x[:, :L], "image_feats": x[:, L:], "cls_feats": cls, "raw_cls_feats": x[:, 0],
            "image_masks": im, "text_masks": text_masks, "text_ids": ids}


def moco_head(p: Params, pre: str, cls: Tensor) -> Tensor:
    """MOCOHead (heads.py:129-143): Linear -> LN(1e-5) -> ReLU -> Linear(no bias)."""
    h = cls @ p[pre + "moco_head.projector.0.weight"].t() + p[pre + "moco_head.projector.0.bias"]
    h = layer_norm(h, p[pre + "moco_head.projector.1.weight"], p[pre + "moco_head.projector.1.bias"], 1e-5)
    return torch.relu(h) @ p[pre + "moco_head.projector.3.weight"].t()


def l2_normalize(x: Tensor) -> Tensor:
    """F.normalize(dim=1), eps 1e-12 (objectives.py:265,269,326)."""
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)


def infonce_logits(q: Tensor, k: Tensor, queue: Tensor, T: float) -> Tensor:
    """objectives.py:328-331: cat([q.k, q @ queue], 1) / T."""
    return torch.cat([(q * k).sum(1, keepdim=True), q @ queue], dim=1) / T


def infonce_loss(logits: Tensor) -> Tensor:
    """CrossEntropyLoss(logits.float(), zeros) mean over batch (objectives.py:333,351)."""
    return (torch.logsumexp(logits, dim=1) - logits[:, 0]).mean()


def queue_metrics(q: Tensor, k: Tensor, queue: Tensor) -> dict:
    """objectives.py:337-349 without the per-query Python loop (same values)."""
    qt = queue.t()                                    # [Kq, P]
    q2 = (q * q).sum(1, keepdim=True)
    c2 = (qt * qt).sum(1)[None]
    dots = q @ queue
    dist = torch.sqrt((q2 + c2 - 2 * dots).clamp_min(0))
    cos = dots / (q.norm(dim=1, keepdim=True) * qt.norm(dim=1)[None]).clamp_min(1e-6)
    return {
        "pos_dist": (q - k).norm(dim=1).mean(),
        "pos_cosine": F.cosine_similarity(q, k, dim=1, eps=1e-6).mean(),
        "pos_dot": (q * k).sum(1).mean(),
        "neg_dist": dist.mean(), "neg_cosine": cos.mean(), "neg_dot": dots.mean(),
    }


def encode_q(p: Params, cfg: dict, ids, masks, img, drop: Optional[dict] = None) -> Tuple[Tensor, dict]:
    out = infer(p, cfg, ids, masks, img, drop=drop)
    return l2_normalize(moco_head(p, "", out["cls_feats"])), out


def pgd_attack(p: Params, cfg: dict, batch: dict, k: Tensor, queue: Tensor,
               return_steps: bool = False, drops: Optional[list] = None):
    """PGDAttack_moco.pgd_attack (attack/pgd_attack_vilt.py:130-175): K steps of
    delta <- clamp(delta + lr * g / max(|g|_inf per sample, 1e-8), +-eps), g = d(CE/K)/d(delta).
    ``drops``: explicit dropout masks of each step's forward (the deep-copied encoder stays in train mode there)."""
    K, lr, eps = cfg["adv_steps_img"], cfg["adv_lr_img"], cfg["adv_max_norm_img"]
    img0 = batch["image"][0]
    delta = torch.zeros_like(img0)
    steps = []
    for step in range(K):
        d = delta.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            q, _ = encode_q(p, cfg, batch["text_ids"], batch["text_masks"], img0 + d, drop=drops[step] if drops else None)
            loss = infonce_loss(infonce_logits(q, k, queue, cfg["temperature"])) / float(K)
            (g,) = torch.autograd.grad(loss, d)
        den = g.abs().flatten(1).max(dim=1).values.clamp_min(1e-8).view(-1, 1, 1, 1)
        delta = delta + lr * g / den
        if eps > 0:
            delta = delta.clamp(-eps, eps)
        delta = delta.detach()
        steps.append(delta.clone())
    return (delta, steps) if return_steps else delta


def infonce_ce_rows(q: Tensor, k: Tensor, queue: Tensor, T: float) -> Tensor:
    """per-row CE (label 0) of the InfoNCE logits"""
    lg = infonce_logits(q, k, queue, T)
    return torch.logsumexp(lg, dim=1) - lg[:, 0]


def text_saliency(p: Params, cfg: dict, ids, masks, img, k, queue):
    """GreedyAttack_moco.get_grad (attack/greedy_attack_vilt.py:406-452): gradient of the batch-mean InfoNCE
    loss wrt the OUTPUT of word_embeddings [B,L,D] (what the backward hook captures), plus q."""
    we = p["text_embeddings.word_embeddings.weight"].detach()[ids].clone().requires_grad_(True)
    pd = {n: t.detach() for n, t in p.items()}
    with torch.enable_grad():
        out = infer(pd, cfg, ids, masks, img, word_embeds=we)
        q = l2_normalize(moco_head(pd, "", out["cls_feats"]))
        loss = infonce_loss(infonce_logits(q, k, queue, cfg["temperature"]))
        (g,) = torch.autograd.grad(loss, we)
    return g, q.detach()


def synthetic_candidates(seed: int, n: int, vocab: int):
    """Deterministic stand-in for the synonym tables (counter-fitted vectors / wordnet are not available
    offline): n pseudo-random replacement token ids for (loop, sample, position)."""
    def fn(loop: int, b: int, t: int, ids_row) -> List[int]:
        g = torch.Generator().manual_seed(seed * 1000003 + loop * 10007 + b * 101 + t)
        return torch.randint(1000, vocab, (n,), generator=g).tolist()
    return fn


def greedy_text_attack(p: Params, cfg: dict, batch: dict, k: Tensor, queue: Tensor, candidate_fn, max_loops: int,
                       sep_id: int = 102):
    """Token-level restatement of GreedyAttack_moco.adv_attack_samples (greedy_attack_vilt.py:494-599) with
    word := token (the tokenizer / stop-word / synonym resources are unavailable offline).  Per loop and sample:
    pick the unused position with the largest L1 saliency (:221-228,:280-308) subject to the 20 % change budget,
    build one sentence per candidate (:312-356), keep the candidate with the largest loss if it beats the current
    loss AND its index is > 0 (the reference's `selected_idx > 0`, :571)."""
    ids = batch["text_ids"].clone()
    masks = batch["text_masks"]
    img = batch["image"][0]
    Bn, Lt = ids.shape
    T = cfg["temperature"]
    orig = ids.clone()
    history = [set() for _ in range(Bn)]
    changes = [0] * Bn
    pd = {n: t.detach() for n, t in p.items()}
    for loop in range(max_loops):
        g, q = text_saliency(p, cfg, ids, masks, img, k, queue)
        sal = g.abs().sum(-1)                                        # [B, L]
        ce0 = infonce_ce_rows(q, k, queue, T)
        cand_ids, owner, pos_of = [], [], []
        for b in range(Bn):
            sep = int((ids[b] == sep_id).nonzero()[0])
            max_len = int(sep * 0.2)
            order = torch.argsort(sal[b, 1:sep], descending=True, stable=True) + 1     # words = tokens 1..sep-1
            chosen = None
            for t in order.tolist():
                if t in history[b] or changes[b] >= min(max_len, max_loops):
                    continue
                chosen = t
                break
            if chosen is None:
                cand_ids.append(ids[b].clone()); owner.append(b); pos_of.append(None)
                continue
            history[b].add(chosen)
            for c in candidate_fn(loop, b, chosen, ids[b]):
                row = ids[b].clone(); row[chosen] = c
                cand_ids.append(row); owner.append(b); pos_of.append(chosen)
        cids = torch.stack(cand_ids)
        own = torch.tensor(owner)
        with torch.no_grad():
            out = infer(pd, cfg, cids, masks[own], img[own])
            qc = l2_normalize(moco_head(pd, "", out["cls_feats"]))
            cec = infonce_ce_rows(qc, k[own], queue, T)
        # split_forward (:454-492) scores candidate j of sample b by the BATCH-MEAN loss with row b replaced, against the
        # original batch-mean loss.  Reference quirk (found while pinning the fixture): `t_save = ori_z[i]` (:475) is a
        # view, so the "restore" at :489 is a no-op and row i keeps its LAST candidate while samples i+1.. are scored:
        # loss_bj = mean(ce0) + sum_{r<b} (ce_{r,last} - ce0_r)/B + (ce_bj - ce0_b)/B.
        ori_loss = float(ce0.mean())
        drift = 0.0
        for b in range(Bn):
            idx = (own == b).nonzero().flatten().tolist()
            best, best_j = ori_loss, -1
            for j, r in enumerate(idx):
                lj = ori_loss + drift + (float(cec[r]) - float(ce0[b])) / Bn
                if lj > best:
                    best, best_j = lj, j
            drift += (float(cec[idx[-1]]) - float(ce0[b])) / Bn
            if pos_of[idx[0]] is None:
                continue
            if best_j > 0:
                changes[b] += 1
                ids[b] = cids[idx[best_j]]
    nchg = [(orig[b] != ids[b]).sum().item() for b in range(Bn)]
    nwords = [int((orig[b] == sep_id).nonzero()[0]) - 1 for b in range(Bn)]
    return {"txt_input_ids": ids, "text_masks": masks, "num_changes": sum(nchg) / Bn,
            "change_rate": sum(c / max(n, 1) for c, n in zip(nchg, nwords)) / Bn, "changes_verification": changes}


def ema_update(p: Params, m: float) -> None:
    """_momentum_update_key_layer (objectives.py:219-224,257-260; on .data, i.e. outside autograd)."""
    with torch.no_grad():
        for name in list(p.keys()):
            if name.startswith("k_"):
                p[name] = p[name] * m + p[name[2:]] * (1.0 - m)


def enqueue(queue: Tensor, ptr: int, keys_all: Tensor, per_step_bs: int) -> int:
    """_dequeue_and_enqueue (objectives.py:238-248); keys_all = rank-major concat of every rank's keys."""
    n = keys_all.shape[0]
    if n != per_step_bs:
        return ptr
    queue[:, ptr:ptr + n] = keys_all.t()
    return (ptr + n) % queue.shape[1]


def compute_moco_contrastive(p: Params, cfg: dict, batch: dict, queue: Tensor, ptr: int,
                             training: bool = True, gathered_keys=None, drops: Optional[dict] = None) -> dict:
    """objectives.compute_moco_contrastive (objectives.py:217-447), image view.

    ``drops`` (image view only): explicit dropout masks per encoder pass - {"key", "clean", "pgd": [one per step], "img"}, each an
    ``infer(drop=...)`` dict - for the training-realistic configuration (drop_rate 0.1: dropout is live in every train-mode forward
    incl. the key encoder and the PGD copies; torch's RNG stream cannot be matched, so the caller hands over the masks).

    Mutates ``p`` (EMA of k_*) and ``queue``.  Returns the loss (with autograd graph onto the
    query params that have requires_grad), logits, delta, metrics and the new queue pointer."""
    if not (cfg["image_view"] or cfg["text_view"] or cfg.get("clean_view", False)):
        raise ZeroDivisionError("loss / loss_num with both views off (objectives.py:250-251,397)")
    ema_update(p, cfg["momentum"])
    ids, masks, img = batch["text_ids"], batch["text_masks"], batch["image"][0]
    with torch.no_grad():
        out_k = infer(p, cfg, ids, masks, img, key=True, drop=(drops or {}).get("key"))
        k = l2_normalize(moco_head(p, "k_", out_k["cls_feats"]))
    q0, _ = encode_q(p, cfg, ids, masks, img, drop=(drops or {}).get("clean"))
    T = cfg["temperature"]
    neg = queue.clone().detach()
    logits0 = infonce_logits(q0, k, neg, T)
    pred0 = logits0.argmax(-1)
    ret = {"k": k, "q_original": q0.detach(), "logits_original": logits0.detach()}
    loss, n = 0.0, 0
    if cfg.get("clean_view", False):
        # BASELINE configs[1] "clean ITM + contrastive" (SURVEY 8d Config 2): CE on the clean logits the reference
        # forms at objectives.py:267-275 (the reference itself never turns them into a loss: quirk 3)
        loss, n = loss + infonce_loss(logits0), n + 1
    t_ids = t_masks = None
    if cfg["text_view"]:                                                   # objectives.py:277-317
        fn = cfg.get("candidate_fn") or synthetic_candidates(cfg.get("seed", 0), cfg["n_candidates"], cfg["vocab_size"])
        att = greedy_text_attack(p, cfg, batch, k, neg, fn, cfg["max_loops"])
        t_ids, t_masks = att["txt_input_ids"], att["text_masks"]
        qt, _ = encode_q(p, cfg, t_ids, t_masks, img)
        lt = infonce_logits(qt, k, neg, T)
        loss, n = loss + infonce_loss(lt), n + 1
        ret.update({"attacked_text_ids": t_ids, "num_changes": att["num_changes"], "change_rate": att["change_rate"],
                    "geom_success_rate": (lt.argmax(-1) != pred0).float().mean()})
    if cfg["image_view"]:
        pd = {kk: (v.detach() if torch.is_tensor(v) else v) for kk, v in p.items()}
        delta, steps = pgd_attack(pd, cfg, batch, k, neg, return_steps=True, drops=(drops or {}).get("pgd"))
        # Reference quirk: pgd_attack overwrites the (deep-copied) batch image in place with
        # img_init + delta_{K-1} on its last iteration (pgd_attack_vilt.py:144) and compute_pgd then
        # adds the returned delta_K on top (objectives.py:176), so the attacked view is
        # img + delta_{K-1} + delta_K (delta_0 = 0), i.e. up to 2*eps away from the clean image.
        prev = steps[-2] if len(steps) >= 2 else torch.zeros_like(delta)
        attacked = img + prev + delta
        qa, _ = encode_q(p, cfg, ids, masks, attacked, drop=(drops or {}).get("img"))
        la = infonce_logits(qa, k, neg, T)
        li = infonce_loss(la)
        loss, n = loss + li, n + 1
        m = queue_metrics(qa.detach(), k, neg)
        ret.update({f"{a}_attacked_img": b for a, b in m.items()})
        ret.update({"delta": delta, "attacked_image": attacked, "q_img_attack": qa.detach(), "logits_img_attack": la.detach(),
                    "pgd_success_rate": (la.argmax(-1) != pred0).float().mean(),
                    "delta_range": delta.norm(dim=1).mean()})          # objectives.py:184
    if cfg["image_view"] and cfg["text_view"]:                             # objectives.py:356-392
        qb, _ = encode_q(p, cfg, t_ids, t_masks, ret["attacked_image"])
        lb = infonce_logits(qb, k, neg, T)
        loss, n = loss + infonce_loss(lb), n + 1
    if training:
        keys_all = k if gathered_keys is None else gathered_keys
        ptr = enqueue(queue, ptr, keys_all, cfg["num_gpus"] * cfg["num_nodes"] * cfg["per_gpu_batchsize"])
    ret["moco_loss"] = loss / n
    ret["ptr"] = ptr
    return ret


# --------------------------------------------------------------------------------------
# Barlow-Twins variant
# --------------------------------------------------------------------------------------
BT_PREFIX = "barlowtwins_head."


def bt_dims(cfg: dict) -> Tuple[int, int, int]:
    """Widths of BarlowTwinsHead: the reference hard-codes [8192, 8192], 8192 (vilt_module.py:115)."""
    return tuple(cfg.get("barlowtwins_dims", (8192, 8192, 8192)))


def bt_param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """state-dict names of heads.BarlowTwinsHead (heads.py:88-107): projector = Sequential(Linear, BatchNorm1d, ReLU,
    Linear, BatchNorm1d, ReLU, Linear), all linears without bias; norm = BatchNorm1d(affine=False) has buffers only."""
    D = cfg["hidden_size"]
    H1, H2, H3 = bt_dims(cfg)
    n = BT_PREFIX + "projector."
    return [(n + "0.weight", (H1, D)), (n + "1.weight", (H1,)), (n + "1.bias", (H1,)),
            (n + "3.weight", (H2, H1)), (n + "4.weight", (H2,)), (n + "4.bias", (H2,)), (n + "6.weight", (H3, H2))]


def bt_init_params(cfg: dict, seed: int) -> Params:
    """Seeded head weights: linears ~ U(+-1/sqrt(fan_in)) (nn.Linear's default range - the reference never applies
    init_weights to this head), BatchNorm gamma ~ 1 + N(0, 0.05), beta ~ N(0, 0.02) (not the (1, 0) default, which would
    hide affine bugs)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    p: Params = {}
    for name, shape in bt_param_shapes(cfg):
        if len(shape) == 2:
            p[name] = (torch.rand(shape, generator=g) * 2 - 1) / float(shape[1]) ** 0.5
        elif name.endswith("weight"):
            p[name] = 1.0 + 0.05 * torch.randn(shape, generator=g)
        else:
            p[name] = 0.02 * torch.randn(shape, generator=g)
    return p


def bt_running_init(cfg: dict) -> Params:
    H1, H2, H3 = bt_dims(cfg)
    out: Params = {}
    for key, n in (("projector.1", H1), ("projector.4", H2), ("norm", H3)):
        out[BT_PREFIX + key + ".running_mean"] = torch.zeros(n)
        out[BT_PREFIX + key + ".running_var"] = torch.ones(n)
        out[BT_PREFIX + key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return out


def batch_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], running: Optional[Params], key: str, training: bool,
               eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """nn.BatchNorm1d: batch statistics (biased variance) in training, running estimates in eval; the running variance is
    updated with the UNBIASED batch variance."""
    if training:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        if running is not None:
            n = x.shape[0]
            with torch.no_grad():
                running[key + ".running_mean"].mul_(1 - momentum).add_(momentum * mean)
                running[key + ".running_var"].mul_(1 - momentum).add_(momentum * var * (n / (n - 1) if n > 1 else 1.0))
                running[key + ".num_batches_tracked"] += 1
    else:
        mean, var = running[key + ".running_mean"], running[key + ".running_var"]
    y = (x - mean) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def barlowtwins_head(p: Params, cls: Tensor, running: Optional[Params] = None, training: bool = True) -> Tensor:
    """BarlowTwinsHead.forward (heads.py:104-107)."""
    n = BT_PREFIX + "projector."
    h = cls @ p[n + "0.weight"].t()
    h = torch.relu(batch_norm(h, p[n + "1.weight"], p[n + "1.bias"], running, n + "1", training))
    h = h @ p[n + "3.weight"].t()
    h = torch.relu(batch_norm(h, p[n + "4.weight"], p[n + "4.bias"], running, n + "4", training))
    h = h @ p[n + "6.weight"].t()
    return batch_norm(h, None, None, running, BT_PREFIX + "norm", training)


def barlow_loss(zq: Tensor, zk: Tensor, denom: float, lam: float, gathered_c=None) -> Tuple[Tensor, Tensor, Tensor]:
    """c = zq^T zk / denom; on_diag = sum (c_ii - 1)^2, off_diag = sum_{i != j} c_ij^2 (objectives.py:478-484).
    ``gathered_c``: callable standing in for the all-reduce of c over ranks (:480)."""
    c = zq.t() @ zk / denom
    if gathered_c is not None:
        c = gathered_c(c)
    d = torch.diagonal(c)
    on = ((d - 1) ** 2).sum()
    off = (c ** 2).sum() - (d ** 2).sum()
    return on + lam * off, on, off


def bt_pgd_attack(p: Params, cfg: dict, batch: dict, k: Tensor, return_steps: bool = False, running: Optional[Params] = None,
                  training: bool = True):
    """PGDAttack_bartlowtwins.pgd_attack (attack/pgd_attack_vilt.py:198-236): like the MoCo PGD with the loss
    (on_diag + adv_lr * off_diag) / K on c = q^T k / B (LOCAL batch, no all-reduce); the head is a deep copy that keeps the
    module's mode: batch statistics in training (its running estimates are a private copy), the running estimates in eval."""
    K, lr, eps = cfg["adv_steps_img"], cfg["adv_lr_img"], cfg["adv_max_norm_img"]
    img0 = batch["image"][0]
    delta = torch.zeros_like(img0)
    steps = []
    for _ in range(K):
        d = delta.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            out = infer(p, cfg, batch["text_ids"], batch["text_masks"], img0 + d)
            q = barlowtwins_head(p, out["cls_feats"], None if training else running, training)
            loss = barlow_loss(q, k, float(q.shape[0]), cfg["adv_lr"])[0] / float(K)
            (g,) = torch.autograd.grad(loss, d)
        den = g.abs().flatten(1).max(dim=1).values.clamp_min(1e-8).view(-1, 1, 1, 1)
        delta = delta + lr * g / den
        if eps > 0:
            delta = delta.clamp(-eps, eps)
        delta = delta.detach()
        steps.append(delta.clone())
    return (delta, steps) if return_steps else delta


def compute_barlowtwins_contrastive(p: Params, cfg: dict, batch: dict, running: Optional[Params] = None, training: bool = True,
                                    gathered_c=None) -> dict:
    """objectives.compute_barlowtwins_contrastive (objectives.py:449-602), image view (the text view differs from MoCo's only
    in the loss its greedy attack maximises, greedy_attack_vilt.py:602-700).  ONE encoder and ONE head (no momentum copies):
    k = head(infer(clean)) under no_grad, q = head(infer(attacked)); BatchNorm runs in the module's mode, so in training
    both calls use their own batch statistics and update the running estimates."""
    if not cfg["image_view"]:
        raise NotImplementedError("oracle: Barlow-Twins image view only")
    ids, masks, img = batch["text_ids"], batch["text_masks"], batch["image"][0]
    per_step_bs = cfg["num_gpus"] * cfg["num_nodes"] * cfg["per_gpu_batchsize"]
    with torch.no_grad():
        k = barlowtwins_head(p, infer(p, cfg, ids, masks, img)["cls_feats"], running, training)
    pd = {kk: (v.detach() if torch.is_tensor(v) else v) for kk, v in p.items()}
    delta, steps = bt_pgd_attack(pd, cfg, batch, k, return_steps=True, running=running, training=training)
    prev = steps[-2] if len(steps) >= 2 else torch.zeros_like(delta)
    attacked = img + prev + delta                                          # same in-place quirk as the MoCo path
    out = infer(p, cfg, ids, masks, attacked)
    q = barlowtwins_head(p, out["cls_feats"], running, training)
    loss, on, off = barlow_loss(q, k, float(per_step_bs), cfg["adv_lr"], gathered_c)
    cos = torch.nn.functional.cosine_similarity(q, k, dim=1, eps=1e-6)
    return {"barlowtwins_loss": loss, "barlowtwins_loss_invariance_img": on, "barlowtwins_loss_redundancy_img": cfg["adv_lr"] * off,
            "k": k, "q_image": q.detach(), "delta": delta, "attacked_image": attacked,
            "pos_dist_attacked_img": (q - k).norm(dim=1).mean().detach(), "pos_cosine_attacked_img": cos.mean().detach(),
            "pos_dot_attacked_img": (q * k).sum(1).mean().detach(), "delta_range": delta.norm(dim=1).mean()}


# --------------------------------------------------------------------------------------
# ITM + word-patch alignment (BASELINE configs 1-2)
# --------------------------------------------------------------------------------------

def ipot(C, x_len, x_pad, y_len, y_pad, joint_pad, beta: float, iteration: int, k: int) -> Tensor:
    """IPOT (objectives.py:46-76). C [B,M,N]; returns T [B,N,M]."""
    b, m, n = C.shape
    sigma = torch.ones(b, m, dtype=C.dtype) / x_len[:, None]
    T = torch.ones(b, n, m, dtype=C.dtype)
    A = torch.exp(-C.transpose(1, 2) / beta)
    sigma = sigma.masked_fill(x_pad, 0)
    jp = joint_pad.transpose(1, 2)
    T = T.masked_fill(jp, 0)
    A = A.masked_fill(jp, 0)
    xl, yl = x_len[:, None, None], y_len[:, None, None]
    xm = (x_pad.to(C.dtype) * 1e4)[:, None]
    ym = (y_pad.to(C.dtype) * 1e4)[:, None]
    for _ in range(iteration):
        Q = A * T
        sigma = sigma.view(b, m, 1)
        for _ in range(k):
            delta = 1 / (yl * Q.matmul(sigma).view(b, 1, n) + ym)
            sigma = 1 / (xl * delta.matmul(Q) + xm)
        T = delta.view(b, n, 1) * Q * sigma
    return T.masked_fill(jp, 0)


def compute_itm_wpa(p: Params, cfg: dict, batch: dict, itm_labels: Tensor) -> dict:
    """compute_itm_wpa (objectives.py:714-787) with the random 50/50 labels supplied by the caller
    (the reference draws them with randperm :715-720)."""
    imgs = torch.stack([batch["image"][0][i] if itm_labels[i] == 1 else batch["false_image_0"][0][i]
                        for i in range(itm_labels.shape[0])])
    out = infer(p, cfg, batch["text_ids"], batch["text_masks"], imgs)
    txt_emb, img_emb = out["text_feats"], out["image_feats"]
    txt_mask, img_mask = out["text_masks"].bool().clone(), out["image_masks"].bool().clone()
    for i, n in enumerate(txt_mask.sum(dim=1)):
        txt_mask[i, n - 1] = False
    txt_mask[:, 0] = False
    img_mask[:, 0] = False
    txt_pad, img_pad = ~txt_mask, ~img_mask
    xn = txt_emb / txt_emb.norm(dim=-1, keepdim=True).clamp_min(1e-5)       # objectives.py:24-34
    yn = img_emb / img_emb.norm(dim=-1, keepdim=True).clamp_min(1e-5)
    cost = 1 - xn @ yn.transpose(1, 2)
    joint_pad = txt_pad[:, :, None] | img_pad[:, None, :]
    cost = cost.masked_fill(joint_pad, 0)
    txt_len = (txt_pad.shape[1] - txt_pad.sum(1)).to(cost.dtype)
    img_len = (img_pad.shape[1] - img_pad.sum(1)).to(cost.dtype)
    T = ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, 0.5, 50, 1)
    distance = torch.einsum("bmn,bnm->b", cost, T.detach())                  # trace(cost @ T) :37-43,761
    pos, neg = distance[itm_labels == 1], distance[itm_labels == 0]
    ot_loss = (pos.sum() - neg.sum()) / (pos.shape[0] + neg.shape[0])
    logits = out["cls_feats"] @ p["itm_score.fc.weight"].t() + p["itm_score.fc.bias"]
    lab = itm_labels.long()
    itm_loss = (torch.logsumexp(logits, 1) - logits.gather(1, lab[:, None])[:, 0]).mean()
    return {"itm_loss": itm_loss, "itm_wpa_loss": 0.1 * ot_loss, "itm_logits": logits,
            "itm_labels": itm_labels, "ot_T": T, "ot_cost": cost.detach(), "cls_feats": out["cls_feats"]}


# --------------------------------------------------------------------------------------
# optimiser ("next" row f1): HF AdamW as used by vilt_utils.set_schedule (:331-437)
# --------------------------------------------------------------------------------------

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight",
            "norm1.bias", "norm1.weight", "norm2.bias", "norm2.weight")
HEAD_NAMES = ("vqa_classifier", "nlvr2_classifier", "moco_head", "barlowtwinshead")


def param_group(name: str) -> Tuple[bool, bool]:
    """(decay?, head?) group membership by substring match (vilt_utils.py:335-393)."""
    return (not any(nd in name for nd in NO_DECAY)), any(h in name for h in HEAD_NAMES)


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.98, eps: float = 1e-8) -> None:
    """transformers.AdamW (pinned 4.2.1, requirements.txt:2) single-tensor step: bias-corrected
    Adam update, then decoupled decay p -= lr*wd*p applied AFTER the update."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
    if wd > 0:
        p.add_(p, alpha=-lr * wd)


def poly_lr(step: int, base_lr: float, warmup: int, total: int, end_lr: float, power: float) -> float:
    """get_polynomial_decay_schedule_with_warmup (HF optimization.py) as called at vilt_utils.py:423-430."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    if step > total:
        return end_lr
    rem = 1 - (step - warmup) / (total - warmup)
    return (base_lr - end_lr) * rem ** power + end_lr
