"""-m gpu: the RMCL hot path (ViLTransformerSS mirror over librmcl_hip.so) against
  (1) the golden vectors generated from the reference's own code (tests/golden/*.npz), and
  (2) the CPU oracle on fresh seeded inputs.
Tolerances: fp32 path -> north_star's 1e-3 on loss/logits (tighter where fp32 allows);
bf16 path -> documented drift bounds (bf16 has 8 significant bits; logits ~ +-40)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rmcl_pkg  # noqa: F401,E402
from oracle import rmcl_oracle as O  # noqa: E402
from rmcl_amd.vilt.config import task_moco  # noqa: E402
from rmcl_amd.vilt.modules import ViLTransformerSS  # noqa: E402
from rmcl_amd.attack.pgd_attack_vilt import PGDAttack_moco  # noqa: E402
from tests.golden_util import cfg_from_meta, digest, load  # noqa: E402

DEV = "cuda:0"


def build_module(ocfg, seed_w, dtype="f32", itm=0):
    cfg = task_moco(num_layers=ocfg["num_layers"], num_negative=ocfg["num_negative"], adv_steps_img=ocfg["adv_steps_img"],
                    per_gpu_batchsize=ocfg["per_gpu_batchsize"], drop_rate=0.0, image_view=True, text_view=False,
                    num_gpus=1, num_nodes=1)
    cfg["loss_names"]["itm"] = itm
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype=dtype)
    p = O.init_params(ocfg, seed_w)
    missing, unexpected = m.load_state_dict({n: t.to(DEV) for n, t in p.items()}, strict=False)
    assert set(missing) <= {"proj_queue", "proj_queue_ptr"}, missing
    m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
    m.train()
    return m, p


def dev_batch(batch):
    return {k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v))
            for k, v in batch.items()}


@pytest.fixture(scope="module", params=["L2_B4_ragged", "L12_B2", "L2_B4_raggedimg", "L2_B3_raggedimg2"])
def case(request):
    """L2_B4_raggedimg: zero-padded batch of 384x352 / 320x384 / 384x384 / 224x288 images (reference: per-sample position-
    embedding resize + valid-patch selection / padding, vision_transformer.py:564-651) - the on-device ragged visual_embed."""
    g = load(f"moco_{request.param}.npz")
    ocfg, B, sw, sb, ragged = cfg_from_meta(O, g["meta"])
    m, p = build_module(ocfg, sw, "f32")
    sizes = [tuple(int(v) for v in r) for r in g["sizes"]] if "sizes" in g.files else None
    batch = O.synthetic_batch(ocfg, B, sb, ragged_text=ragged, sizes=sizes)
    return g, ocfg, m, p, batch


def test_infer_matches_reference_golden(case):
    g, ocfg, m, p, batch = case
    with torch.no_grad():                                    # (the stash-free inference pass; with autograd on infer is differentiable)
        r = m.infer(dev_batch(batch))
    np.testing.assert_allclose(r["cls_feats"].cpu().numpy(), g["cls_feats"], atol=1e-4)
    np.testing.assert_allclose(r["raw_cls_feats"].cpu().numpy(), g["raw_cls_feats"], atol=2e-4)
    np.testing.assert_allclose(r["text_feats"].cpu().numpy(), g["text_feats"], atol=2e-4)
    np.testing.assert_allclose(r["image_feats"].cpu().numpy(), g["image_feats"], atol=2e-4)
    if "image_masks" in g.files:                             # zero-padded batch: masks and the selected (row, col) of every valid slot
        np.testing.assert_array_equal(r["image_masks"].cpu().numpy(), g["image_masks"])
        pi, (gh, gw) = r["patch_index"]
        flat = (pi[..., 0] * gw + pi[..., 1]).cpu().numpy()
        valid = g["image_masks"][:, 1:] == 1
        np.testing.assert_array_equal(flat[valid], g["patch_index_flat"][valid])
    else:
        assert r["image_masks"].shape == (batch["text_ids"].shape[0], 145) and bool((r["image_masks"] == 1).all())
    rk = m.infer_k(dev_batch(batch))                         # momentum copies == query weights at init
    np.testing.assert_allclose(rk["cls_feats"].cpu().numpy(), g["cls_feats"], atol=1e-4)


def test_pgd_attack_matches_reference_golden(case):
    g, ocfg, m, p, batch = case
    k = torch.from_numpy(g["pgd_k_input"]).to(DEV)
    for K in (1, ocfg["adv_steps_img"]):
        att = PGDAttack_moco(dict(m.config, adv_steps_img=K))
        b = dev_batch(batch)
        img0 = b["image"][0].clone()
        delta = att.pgd_attack(m, b, k_modality=k)
        assert delta.shape == img0.shape
        np.testing.assert_allclose(delta[:, :, ::8, ::8].cpu().numpy(), g[f"pgd_delta_K{K}_sub"], atol=2e-5)
        np.testing.assert_allclose(delta[:, :, :32, :32].cpu().numpy(), g[f"pgd_delta_K{K}_patch00"], atol=2e-5)
        np.testing.assert_allclose(delta.flatten(1).norm(dim=1).cpu().numpy(), g[f"pgd_delta_K{K}_persample_l2"], rtol=2e-3)
        assert float(delta.abs().max()) <= ocfg["adv_max_norm_img"] + 1e-9
        if K == 1:                                            # batch image left at img + delta_{K-1} = img
            assert torch.equal(b["image"][0], img0)


def test_training_step_matches_reference_golden(case):
    g, ocfg, m, p, batch = case
    m.zero_grad()
    m.queue_ptr = 0
    m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
    m.shadow_momentum_encoder()
    loss = m.training_step(dev_batch(batch), 0)
    assert abs(float(loss) - float(g["moco_loss"])) < 1e-3                       # north_star tolerance
    loss.backward()
    B = batch["text_ids"].shape[0]
    assert m.queue_ptr == int(g["queue_ptr_after"]) and int(m.proj_queue_ptr) == int(g["queue_ptr_after"])
    np.testing.assert_allclose(m.proj_queue[:, : 2 * B].cpu().numpy(), g["queue_head_after"], atol=1e-4)
    lg = m.logged
    for a in ("dist", "cosine", "dot"):
        tag = {"dist": "L2", "cosine": "Cosine", "dot": "Dot"}[a]
        assert abs(float(lg[f"moco_dist_train_{tag}/Pos_attacked_img"]) - float(g[f"ret_pos_{a}_attacked_img"])) < 1e-3
        assert abs(float(lg[f"moco_dist_train_{tag}/Neg_attacked_img"]) - float(g[f"ret_neg_{a}_attacked_img"])) < 1e-3
    assert abs(float(lg["moco_attack/train/delta"]) - float(g["log_moco_attack__train__delta"])) < 1e-6
    assert abs(float(lg["moco_attack/PGD_success_rate"]) - float(g["log_moco_attack__PGD_success_rate"])) < 1e-6
    sd = m.state_dict()
    for name, dg in zip(g["ema_names"], g["ema_digest"]):
        np.testing.assert_allclose(digest(sd[str(name)]), dg, rtol=2e-5, atol=1e-5, err_msg=str(name))
    params = dict(m.named_parameters())
    for name, dg in zip(g["grad_names"], g["grad_digest"]):
        if str(name).startswith("itm_score"):
            continue
        mine = digest(params[str(name)].grad)
        assert abs(mine[1] - dg[1]) <= 5e-3 * max(dg[1], 1e-6) + 1e-7, (name, mine[1], dg[1])
        np.testing.assert_allclose(mine[3:], dg[3:], atol=5e-3 * dg[2] + 1e-7, err_msg=str(name))
    np.testing.assert_allclose(params["pooler.dense.weight"].grad[:8, :64].cpu().numpy(), g["grad_pooler_w"],
                               atol=5e-3 * np.abs(g["grad_pooler_w"]).max())
    np.testing.assert_allclose(params["transformer.blocks.0.attn.qkv.weight"].grad[:8, :64].cpu().numpy(), g["grad_qkv0_w"],
                               atol=5e-3 * np.abs(g["grad_qkv0_w"]).max())
    np.testing.assert_allclose(params["transformer.patch_embed.proj.weight"].grad[:4, :, :4, :8].cpu().numpy(), g["grad_patch_w"],
                               atol=5e-3 * np.abs(g["grad_patch_w"]).max())
    np.testing.assert_allclose(params["transformer.pos_embed"].grad[0, :4, :64].cpu().numpy(), g["grad_pos_embed"],
                               atol=5e-3 * np.abs(g["grad_pos_embed"]).max())
    ids = batch["text_ids"]
    we = params["text_embeddings.word_embeddings.weight"].grad
    np.testing.assert_allclose(we[ids[0, :4].to(DEV)][:, :64].cpu().numpy(), g["grad_word_rows"],
                               atol=5e-3 * np.abs(g["grad_word_rows"]).max() + 1e-9)


def test_bf16_path_tracks_fp32_golden():
    """bf16 storage / fp32 accumulate path: bounded drift against the reference numbers."""
    g = load("moco_L2_B4_ragged.npz")
    ocfg, B, sw, sb, ragged = cfg_from_meta(O, g["meta"])
    m, p = build_module(ocfg, sw, "bf16")
    batch = O.synthetic_batch(ocfg, B, sb, ragged_text=ragged)
    with torch.no_grad():                                    # (the stash-free inference pass; with autograd on infer is differentiable)
        r = m.infer(dev_batch(batch))
    assert float((r["cls_feats"].cpu() - torch.from_numpy(g["cls_feats"])).abs().max()) < 3e-2
    loss = m.training_step(dev_batch(batch), 0)
    assert abs(float(loss) - float(g["moco_loss"])) < 0.5          # logits ~ +-40 in bf16 inputs: 8-bit mantissa
    loss.backward()
    params = dict(m.named_parameters())
    gq = params["transformer.blocks.0.attn.qkv.weight"].grad
    ref = g["grad_digest"][list(g["grad_names"]).index("transformer.blocks.0.attn.qkv.weight")]
    assert abs(float(gq.double().norm()) - ref[1]) < 0.1 * ref[1]


def test_both_views_off_raises_like_reference():
    ocfg = O.default_config(num_layers=1, num_negative=1024, per_gpu_batchsize=2)
    cfg = task_moco(num_layers=1, num_negative=1024, per_gpu_batchsize=2, drop_rate=0.0)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="f32")
    with pytest.raises(ZeroDivisionError):
        m.training_step(dev_batch(O.synthetic_batch(ocfg, 2, 1)), 0)


def test_full_size_properties_bs64():
    """BASELINE size (B=64, queue 65536, K=3, bf16): properties that hold at any size."""
    ocfg = O.default_config(per_gpu_batchsize=64)
    m, p = build_module(ocfg, 7, "bf16")
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 5))
    q0 = m.proj_queue.clone()
    loss = m.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss) and 20.0 < float(loss) < 80.0
    pb = m.engine.bufs(64)
    assert float(pb.delta.abs().max()) <= 0.005 + 1e-9 and float(pb.delta.abs().max()) > 0.0049       # eps-ball, saturated
    assert float((pb.q.norm(dim=1) - 1).abs().max()) < 1e-5 and float((pb.k.norm(dim=1) - 1).abs().max()) < 1e-5
    assert m.queue_ptr == 64 and torch.equal(m.proj_queue[:, 64:], q0[:, 64:])                      # only the block moved
    assert torch.allclose(m.proj_queue[:, :64].t(), pb.k, atol=0)                                    # enqueue == keys^T
    gn = float(m.engine.g32.norm())
    assert np.isfinite(gn) and gn > 0
    # EMA idempotence property: with q == k the momentum update is a fixed point
    k_before = m.engine.k32.clone()
    m.engine.ema(0.999)
    assert float((m.engine.k32 - k_before).abs().max()) < 1e-6


@pytest.mark.parametrize("tag", ["L2_B4_ragged", "L12_B2"])
def test_itm_wpa_matches_reference_golden(tag):
    """BASELINE configs[0]/[1] objective: ITM head + word-patch alignment (IPOT) against the reference."""
    g = load(f"itm_{tag}.npz")
    ocfg, B, sw, sb, ragged = cfg_from_meta(O, g["meta"], kind="itm")
    cfg = task_moco(num_layers=ocfg["num_layers"], num_negative=1024, per_gpu_batchsize=B, drop_rate=0.0, num_gpus=1, num_nodes=1)
    cfg["loss_names"] = dict(cfg["loss_names"], moco=0, itm=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="f32")
    p = O.init_params(ocfg, sw)
    m.load_state_dict({n: t.to(DEV) for n, t in p.items() if not n.startswith("k_") and not n.startswith("moco_head")}, strict=False)
    m.train()
    m.itm_labels_override = torch.from_numpy(g["itm_labels"])
    batch = dev_batch(O.synthetic_batch(ocfg, B, sb, ragged_text=ragged))
    m.zero_grad()
    loss = m.training_step(batch, 0)                                   # itm_loss + itm_wpa_loss (vilt_module.py:475)
    assert abs(float(loss) - (float(g["itm_loss"]) + float(g["itm_wpa_loss"]))) < 1e-4
    np.testing.assert_allclose(m.logged["itm/train/loss"].item(), float(g["itm_loss"]), atol=1e-4)
    np.testing.assert_allclose(m.logged["itm/train/wpa_loss"].item(), float(g["itm_wpa_loss"]), atol=2e-5)
    loss.backward()
    params = dict(m.named_parameters())
    for name, dg in zip(g["grad_names"], g["grad_digest"]):
        if str(name) not in params:
            continue
        mine = digest(params[str(name)].grad)
        assert abs(mine[1] - dg[1]) <= 5e-3 * max(dg[1], 1e-6) + 1e-7, (name, mine[1], dg[1])
        np.testing.assert_allclose(mine[3:], dg[3:], atol=5e-3 * dg[2] + 1e-7, err_msg=str(name))


def test_itm_wpa_full_size_bs64_kernels_agree():
    """BASELINE configs[1] size (clean ITM + word-patch alignment, bs=64, bf16): the step on the 192x192 kernels against the
    128x128 kernels (tune cfg 2), same labels and inputs; plus size-independent properties of the objective."""
    from rmcl_amd._lib import lib
    B = 64
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=B)
    cfg = task_moco(num_layers=2, num_negative=1024, per_gpu_batchsize=B, drop_rate=0.0, num_gpus=1, num_nodes=1)
    cfg["loss_names"] = dict(cfg["loss_names"], moco=0, itm=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
    p = O.init_params(ocfg, 3)
    m.load_state_dict({n: t.to(DEV) for n, t in p.items() if not n.startswith("k_") and not n.startswith("moco_head")}, strict=False)
    m.train()
    g = torch.Generator().manual_seed(11)
    m.itm_labels_override = (torch.rand(B, generator=g) < 0.5).long()
    batch = dev_batch(O.synthetic_batch(ocfg, B, 13, ragged_text=True))
    out = []
    try:
        for cfgk in (-1, 2):
            lib.rmcl_tune_set(0, cfgk)
            m.zero_grad()
            loss = m.training_step(batch, 0)
            loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss), m.logged["itm/train/loss"].item(), m.logged["itm/train/wpa_loss"].item(), m.engine.g32.clone()))
    finally:
        lib.rmcl_tune_set(0, -1)
    (l0, i0, w0, g0), (l1, i1, w1, g1) = out
    assert np.isfinite(l0) and 0.3 < i0 < 1.5                                   # 2-class CE of an untrained head ~ ln 2
    assert abs(l0 - (i0 + w0)) < 1e-4                                           # total = itm + wpa (vilt_module.py:475)
    assert abs(i0 - i1) < 2e-2 and abs(w0 - w1) < 2e-2 * max(1.0, abs(w1))
    assert float((g0 - g1).norm() / g1.norm()) < 3e-2


def test_side_stream_weight_gradients_are_race_free():
    """The weight-gradient GEMMs run on a second stream behind events; the split-K slab reduce is ordered,
    so weight-matrix gradients must be BITWISE equal with and without the side stream."""
    import ctypes as C
    from rmcl_amd._lib import lib
    ocfg = O.default_config(num_layers=3, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1)
    m, p = build_module(ocfg, 5, "bf16")
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9))
    grads = []
    for side in (True, False):
        lib.rmcl_set_side_stream(C.c_void_p(m.engine.dw_stream.cuda_stream if side else 0))
        m.zero_grad()
        m.queue_ptr = 0
        m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
        m.shadow_momentum_encoder()
        loss = m.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        grads.append(m.engine.g32.clone())
    lib.rmcl_set_side_stream(C.c_void_p(m.engine.dw_stream.cuda_stream))
    lay = m.engine.layout
    for l in range(3):
        base = lay.layer0 + l * lay.layer_stride
        for off, n in ((lay.qkv_w, 3 * 768 * 768), (lay.proj_w, 768 * 768), (lay.fc1_w, 3072 * 768), (lay.fc2_w, 3072 * 768)):
            a, b = grads[0][base + off: base + off + n], grads[1][base + off: base + off + n]
            assert torch.equal(a, b), (l, off)
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-4 * float(grads[1].abs().max())    # bias grads use float atomics


def test_dropout_step_kernels_agree_bs64():
    """Training-realistic configuration (drop_rate 0.1, reference default config.py:57) at B = 64: the dropout epilogues of
    the 192x192 kernels against those of the 128x128 kernels.  The masks are a pure function of (pass seed, layer, site,
    element), so with the pass counter rewound both runs draw identical masks and must agree like the no-dropout runs."""
    from rmcl_amd._lib import lib
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1)
    cfg = task_moco(num_layers=2, num_negative=1024, adv_steps_img=1, per_gpu_batchsize=64, drop_rate=0.1, image_view=True,
                    text_view=False, num_gpus=1, num_nodes=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
    m.load_state_dict({n: t.to(DEV) for n, t in O.init_params(ocfg, 5).items()}, strict=False)
    m.train()
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9))
    grads, losses = [], []
    try:
        for cfgk in (-1, 2):
            lib.rmcl_tune_set(0, cfgk)
            m.engine.pass_counter = 0
            m.zero_grad()
            m.queue_ptr = 0
            m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
            m.shadow_momentum_encoder()
            loss = m.training_step(batch, 0)
            loss.backward()
            torch.cuda.synchronize()
            grads.append(m.engine.g32.clone())
            losses.append(float(loss))
    finally:
        lib.rmcl_tune_set(0, -1)
    assert abs(losses[0] - losses[1]) < 2e-2 * abs(losses[1])
    lay = m.engine.layout
    for l in range(2):
        base = lay.layer0 + l * lay.layer_stride
        for off, n in ((lay.qkv_w, 3 * 768 * 768), (lay.proj_w, 768 * 768), (lay.fc1_w, 3072 * 768), (lay.fc2_w, 3072 * 768)):
            a, b = grads[0][base + off: base + off + n], grads[1][base + off: base + off + n]
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert rel < 3e-2, (l, off, rel)


def test_weight_gradient_kernels_agree_bs64():
    """B = 64 (tokens = 11840): the 192x192 ping-pong kernels (activation GEMMs + [K][M]x[K][N] split-K weight gradients)
    against the 128x128 kernels (tune cfg 2) on the same step - same bf16 operands, fp32 accumulation, so only the
    summation order and the bf16 rounding points of intermediate activations differ."""
    from rmcl_amd._lib import lib
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1)
    m, p = build_module(ocfg, 5, "bf16")
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9))
    grads, losses = [], []
    try:
        for cfg in (-1, 2):
            lib.rmcl_tune_set(0, cfg)
            m.zero_grad()
            m.queue_ptr = 0
            m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
            m.shadow_momentum_encoder()
            loss = m.training_step(batch, 0)
            loss.backward()
            torch.cuda.synchronize()
            grads.append(m.engine.g32.clone())
            losses.append(float(loss))
    finally:
        lib.rmcl_tune_set(0, -1)
    assert abs(losses[0] - losses[1]) < 2e-2 * abs(losses[1])
    lay = m.engine.layout
    for l in range(2):
        base = lay.layer0 + l * lay.layer_stride
        for off, n in ((lay.qkv_w, 3 * 768 * 768), (lay.proj_w, 768 * 768), (lay.fc1_w, 3072 * 768), (lay.fc2_w, 3072 * 768)):
            a, b = grads[0][base + off: base + off + n], grads[1][base + off: base + off + n]
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert rel < 3e-2, (l, off, rel)


def test_overlapped_gradient_sync_is_exact_on_one_rank():
    """The N > 1 gradient path (per-layer all-reduces on the communication stream, gated on the backward's
    gradient-ready events) run over a 1-rank RCCL group: averaging over one rank is the identity, so the gradients must
    equal the unsynchronised run (weight matrices bit for bit) - a missing event / stream dependency shows up as a torn
    bucket."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        ocfg = O.default_config(num_layers=3, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1)
        m, p = build_module(ocfg, 5, "bf16")
        (opt,), _ = m.configure_optimizers()
        batch = dev_batch(O.synthetic_batch(ocfg, 64, 9))
        q0 = m.engine.q32.clone()
        outs = []
        for sync, algo in ((False, "ring"), (True, "ring"), (True, "direct")):
            m.sync_grads = sync
            m.hparams.config["grad_allreduce_algo"] = algo     # "direct": every per-layer bucket through DirectReduce
            m.engine.q32.copy_(q0)
            m.engine.lp_stale = True
            opt.m.zero_(); opt.v.zero_(); opt.t = 0
            m.zero_grad()
            m.queue_ptr = 0
            m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
            m.shadow_momentum_encoder()
            loss = m.training_step(batch, 0)
            loss.backward()
            assert (m.step_sync.handle is not None) == sync           # single closure -> the overlapped per-layer path
            opt.step()
            torch.cuda.synchronize()
            outs.append((m.engine.g32.clone(), m.engine.q32.clone()))
        lay = m.engine.layout
        for l in range(3):                                     # weight-matrix gradients: ordered slab reduce -> bitwise
            base = lay.layer0 + l * lay.layer_stride
            for off, n in ((lay.qkv_w, 3 * 768 * 768), (lay.proj_w, 768 * 768), (lay.fc1_w, 3072 * 768), (lay.fc2_w, 3072 * 768)):
                for other in (1, 2):
                    assert torch.equal(outs[0][0][base + off: base + off + n], outs[other][0][base + off: base + off + n]), (l, off, other)
        # bias / LayerNorm gradients use float atomics (summation order varies run to run)
        for other in (1, 2):
            assert float((outs[0][0] - outs[other][0]).abs().max()) <= 1e-4 * float(outs[other][0].abs().max())
            assert float((outs[0][1] - outs[other][1]).abs().max()) <= 2.1 * float(opt.param_groups[0]["lr"]) * 10
    finally:
        m.sync_grads = True
        if created:
            dist.destroy_process_group()


def test_text_attack_and_three_view_step_match_oracle():
    """BASELINE configs[4] tensor side (text view + image view + both view).  The reference's linguistic
    resources are unavailable offline, so there is no reference golden for this row ("parity unpinned");
    the HIP path is checked against the oracle's restatement with the same synthetic candidate generator."""
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=3, text_view=True, image_view=True,
                            adv_steps_img=2, max_loops=2, n_candidates=5, seed=0)
    cfg = task_moco(num_layers=2, num_negative=1024, per_gpu_batchsize=3, drop_rate=0.0, image_view=True, text_view=True,
                    adv_steps_img=2, max_loops=2, n_candidates=5, seed=0, num_gpus=1, num_nodes=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="f32")
    p = O.init_params(ocfg, 3)
    m.load_state_dict({n: t.to(DEV) for n, t in p.items()}, strict=False)
    queue = O.init_queue(ocfg, 0)
    m.proj_queue.copy_(queue.to(DEV))
    m.train()
    batch = O.synthetic_batch(ocfg, 3, 4, ragged_text=True)
    # the attack alone: identical token substitutions
    with torch.no_grad():
        k = O.l2_normalize(O.moco_head(p, "k_", O.infer(p, ocfg, batch["text_ids"], batch["text_masks"], batch["image"][0], key=True)["cls_feats"]))
    ref_att = O.greedy_text_attack(p, ocfg, batch, k, queue, O.synthetic_candidates(0, 5, ocfg["vocab_size"]), 2)
    att = m.greedy_attacker.adv_attack_samples(m, dev_batch(batch), k.to(DEV))
    assert torch.equal(att["txt_input_ids"].cpu(), ref_att["txt_input_ids"])
    assert att["changes_verification"] == ref_att["changes_verification"]
    assert abs(att["num_changes"] - ref_att["num_changes"]) < 1e-9 and abs(att["change_rate"] - ref_att["change_rate"]) < 1e-9
    # the full three-view step
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    ref = O.compute_moco_contrastive(p, ocfg, batch, queue, 0, training=True)
    ref["moco_loss"].backward()
    m.zero_grad()
    loss = m.training_step(dev_batch(batch), 0)
    assert abs(float(loss) - float(ref["moco_loss"])) < 1e-3
    loss.backward()
    params = dict(m.named_parameters())
    for name in ("transformer.blocks.0.attn.qkv.weight", "text_embeddings.word_embeddings.weight", "pooler.dense.weight",
                 "transformer.patch_embed.proj.weight", "moco_head.projector.3.weight"):
        a, b = params[name].grad.cpu(), p[name].grad
        assert float((a - b).abs().max()) <= 5e-3 * float(b.abs().max()) + 1e-8, name


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dropout_forward_backward_match_oracle_with_same_masks(dtype):
    """Dropout (reference default drop_rate=0.1; BertEmbeddings dropout, pos_drop, proj_drop, both Mlp drops).
    torch's RNG stream cannot be reproduced, so parity is checked the other way round: the HIP path's counter-based
    masks are materialised (rmcl_dropout_mask_apply) and fed to the oracle as explicit masks; forward AND backward
    must then agree like the deterministic path does."""
    import ctypes as C
    from rmcl_amd import _lib as L
    from rmcl_amd._lib import lib, check, P, I64, F
    from rmcl_amd.runtime import stream_ptr
    pdrop, Bn = 0.25, 3
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=Bn)
    cfg = task_moco(num_layers=2, num_negative=1024, per_gpu_batchsize=Bn, drop_rate=pdrop, image_view=True, num_gpus=1, num_nodes=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype=dtype)
    p = O.init_params(ocfg, 3)
    m.load_state_dict({n: t.to(DEV) for n, t in p.items()}, strict=False)
    queue = O.init_queue(ocfg, 0)
    m.proj_queue.copy_(queue.to(DEV))
    batch = O.synthetic_batch(ocfg, Bn, 4, ragged_text=True)
    eng = m.engine
    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
    op = eng.make_operand(pb, out=pb.patchesT_full)
    k = torch.nn.functional.normalize(torch.randn(Bn, 128, generator=torch.Generator().manual_seed(1)), dim=1)
    pb.k.copy_(k.to(DEV))
    # eval mode: dropout off -> identical to the deterministic forward
    eng.dropout_on = False
    eng.encoder_forward(pb, key=False, mode=L.MODE_INFER, patchesT=op)
    xn_eval = pb.xn.clone()
    eng.dropout_on = True
    eng.zero_grads()
    eng.encoder_forward(pb, key=False, mode=L.MODE_FULL, patchesT=op)
    seed, pp = pb.drop[L.MODE_FULL]
    assert pp == pdrop and not torch.equal(pb.xn, xn_eval)
    eng.heads_forward(pb, key=False)
    eng.infonce(pb, 1.0 / Bn, want_dq=True)
    loss = float(pb.loss_sum)
    eng.heads_backward(pb, pb.dq, None, with_grads=True)
    dpat = torch.empty_like(pb.patchesT_full)
    eng.encoder_backward(pb, L.MODE_FULL, op, pb.dcls, cls_only=True, dpatches=dpat)
    torch.cuda.synchronize()

    def mask(shape, layer, site):
        x = torch.ones(shape, device=DEV)
        check(lib.rmcl_dropout_mask_apply(P(x), I64(x.numel()), C.c_uint32(seed), layer, site, F(pdrop), stream_ptr()))
        return x.cpu()
    N, D = 185, 768
    drop = {"text": mask((Bn, 40, D), 0, 3), "image": mask((Bn, 145, D), 0, 4)}
    for l in range(2):
        drop[l] = {"proj": mask((Bn, N, D), l, 0), "hidden": mask((Bn, N, 4 * D), l, 1), "fc2": mask((Bn, N, D), l, 2)}
    keep = float((drop[0]["hidden"] != 0).float().mean())
    assert abs(keep - (1 - pdrop)) < 5e-3 and abs(float(drop[0]["hidden"].max()) - 1 / (1 - pdrop)) < 1e-6
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    img = batch["image"][0].clone().requires_grad_(True)
    out = O.infer(p, ocfg, batch["text_ids"], batch["text_masks"], img, drop=drop)
    q = O.l2_normalize(O.moco_head(p, "", out["cls_feats"]))
    ref = O.infonce_loss(O.infonce_logits(q, k, queue, ocfg["temperature"]))
    ref.backward()
    tol_l, tol_g = (1e-3, 5e-3) if dtype == "f32" else (0.5, 0.2)
    assert abs(loss - float(ref)) < tol_l
    params = dict(m.named_parameters())
    for name in ("transformer.blocks.1.mlp.fc2.weight", "transformer.blocks.0.mlp.fc1.weight", "transformer.blocks.0.attn.proj.bias",
                 "transformer.blocks.0.attn.qkv.weight", "transformer.pos_embed", "transformer.cls_token", "token_type_embeddings.weight",
                 "text_embeddings.LayerNorm.weight", "text_embeddings.position_embeddings.weight", "transformer.patch_embed.proj.weight"):
        a, b = params[name].grad.cpu(), p[name].grad
        assert float((a - b).abs().max()) <= tol_g * float(b.abs().max()) + 1e-8, name
    g_img = O.patchify(img.grad, 32).reshape(Bn * 144, 3072)
    assert float((dpat.float().cpu() - g_img).abs().max()) <= tol_g * float(g_img.abs().max()) + 1e-9


@pytest.mark.parametrize("drop_rate", [0.0, 0.1])
def test_short_training_run_reduces_loss_and_keeps_state_consistent(drop_rate):
    """End-to-end: 12 optimizer steps (bf16, fused AdamW, EMA, enqueue) on a fixed synthetic batch: the loss falls,
    everything stays finite, the momentum encoder trails the query encoder, the queue pointer wraps as in the reference.
    drop_rate 0.1 = the reference's recipe (config.py:57): dropout in every train-mode pass, half-batch lanes (B = 32), cls-only tail and
    LayerNorm fold under dropout, the clean query forward on the key stream."""
    ocfg = O.default_config(num_layers=2, num_negative=256, per_gpu_batchsize=32, adv_steps_img=1)
    cfg = task_moco(num_layers=2, num_negative=256, per_gpu_batchsize=32, adv_steps_img=1, drop_rate=drop_rate, image_view=True,
                    num_gpus=1, num_nodes=1, learning_rate=2e-4, warmup_steps=2, max_steps=100, dense_images=True)
    torch.manual_seed(0)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
    m.train()
    (opt,), (sched,) = m.configure_optimizers()
    batch = dev_batch(O.synthetic_batch(ocfg, 32, 3))
    losses = []
    for i in range(12):
        loss = m.training_step(batch, i)
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        opt.zero_grad()
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0] - 0.5, losses
    if drop_rate > 0:
        assert getattr(m.engine.bufs(32), "_lanes", None) is not None and m.engine.fold, "lanes / fold expected under dropout"
    assert m.queue_ptr == (12 * 32) % 256                                   # 256 % 32 == 0: wraps cleanly
    e = m.engine
    diff = float((e.q32[: e.layout.ema_end] - e.k32).abs().max())
    assert 0 < diff < 0.1                                                   # k trails q (m = 0.999), never equal after updates
    assert torch.equal(e.q_lp.float()[:1000], e.q32[:1000].to(torch.bfloat16).float())   # bf16 shadow refreshed by AdamW


def test_grouped_weight_gradients_match_per_gemm_path_bs64():
    """B = 64: the per-layer weight-gradient launch (four dW + four bias gradients + the LayerNorm dgamma/dbeta finish in ONE
    kernel, no split-K slabs) against the per-GEMM path (split-K slabs + ordered reduce + column-sum kernels) on the same step.
    Same bf16 operands, fp32 accumulation: only the summation order differs.  EVERY gradient of the arena is compared."""
    from rmcl_amd._lib import lib
    ocfg = O.default_config(num_layers=3, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1)
    m, p = build_module(ocfg, 5, "bf16")
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9, ragged_text=True))
    grads, losses = [], []
    try:
        for grouped in (1, 0, 1):
            lib.rmcl_tune_set(3, grouped)
            m.zero_grad()
            m.queue_ptr = 0
            m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
            m.shadow_momentum_encoder()
            loss = m.training_step(batch, 0)
            loss.backward()
            torch.cuda.synchronize()
            grads.append(m.engine.g32.clone())
            losses.append(float(loss))
    finally:
        lib.rmcl_tune_set(3, 1)
    assert max(losses) - min(losses) < 1e-4          # (the batch mean is 64 float atomics: last-bit differences run to run)
    params = dict(m.named_parameters())
    lay = m.engine.layout
    worst = 0.0
    for name, prm in params.items():
        if name.startswith("k_") or name.startswith("itm_score"):
            continue
        off = prm.grad.data_ptr() - m.engine.g32.data_ptr()
        assert off % 4 == 0
        a = grads[0][off // 4: off // 4 + prm.numel()]
        b = grads[1][off // 4: off // 4 + prm.numel()]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        assert rel < 2e-4, (name, rel)
    # the grouped launch accumulates every weight tile in one workgroup in k order: bitwise reproducible run to run
    for l in range(3):
        base = lay.layer0 + l * lay.layer_stride
        for off, n in ((lay.qkv_w, 3 * 768 * 768), (lay.proj_w, 768 * 768), (lay.fc1_w, 3072 * 768), (lay.fc2_w, 3072 * 768)):
            assert torch.equal(grads[0][base + off: base + off + n], grads[2][base + off: base + off + n]), (l, off)


def test_two_closure_step_reduces_once_on_one_rank():
    """itm + clean-InfoNCE in one training_step = TWO deferred backwards into one arena, under a 1-rank RCCL group: the arena
    must be reduced exactly once, after the second closure (blocking path, no per-layer overlap), and - averaging over one
    rank being the identity - equal the unsynchronised run."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        B = 8
        cfg = task_moco(num_layers=2, num_negative=1024, per_gpu_batchsize=B, drop_rate=0.0, image_view=False, text_view=False,
                        clean_view=True, num_gpus=1, num_nodes=1)
        cfg["loss_names"]["itm"] = 1
        ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=B)
        m = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
        m.load_state_dict({n: t.to(DEV) for n, t in O.init_params(ocfg, 5, k_seed=6).items()}, strict=False)
        m.train()
        m.itm_labels_override = (torch.arange(B) % 2)
        batch = dev_batch(O.synthetic_batch(ocfg, B, 9, ragged_text=True))
        outs = []
        k0, q0 = m.engine.k32.clone(), m.proj_queue.clone()
        for sync, algo in ((False, "ring"), (True, "ring"), (True, "direct")):
            m.sync_grads = sync
            m.step_sync.algo = algo                                # "direct": one-hop reduce-scatter + all-gather (DirectReduce)
            m.zero_grad()
            m.queue_ptr = 0
            m.engine.k32.copy_(k0)                                 # same momentum weights and queue for every run
            m.engine.lp_stale = True
            m.proj_queue.copy_(q0)
            loss = m.training_step(batch, 0)
            assert m.step_sync.created == 2 and m.step_sync.open == 2
            loss.backward()
            assert m.step_sync.open == 0 and m.step_sync.handle is None      # two closures: one blocking pass, no overlap handle
            torch.cuda.synchronize()
            outs.append(m.engine.g32.clone())
        assert float((outs[0] - outs[1]).abs().max()) <= 1e-4 * float(outs[1].abs().max())
        assert float((outs[0] - outs[2]).abs().max()) <= 1e-4 * float(outs[2].abs().max())
    finally:
        m.step_sync.algo = "ring"
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("dtype,Bn,layers,pdrop", [("f32", 5, 2, 0.0), ("bf16", 64, 3, 0.0), ("f32", 5, 2, 0.25), ("bf16", 64, 2, 0.1)])
def test_cls_only_tail_equals_the_dense_last_block(dtype, Bn, layers, pdrop):
    """RMCL_MODE_CLS_TAIL (include/rmcl.h): with only the cls rows of the encoder output read, the last block's row-wise part
    runs on B rows - cls features, the PGD data gradient (DATA mode) and every parameter gradient (FULL mode, incl. the last
    layer's fc1 / fc2 / proj / LayerNorm tensors reduced over B rows) must equal the dense pass up to summation order
    (f32) / up to the bf16 rounding of the dense last block (bf16: the tail keeps fp32 operands).  pdrop > 0 (round 4): the compact
    rows draw the dropout masks of the dense rows they stand for, so with the pass seeds rewound the same holds under dropout."""
    from rmcl_amd import _lib as L
    ocfg = O.default_config(num_layers=layers, num_negative=1024, per_gpu_batchsize=Bn)
    cfg = task_moco(num_layers=layers, num_negative=1024, per_gpu_batchsize=Bn, drop_rate=pdrop, image_view=True, num_gpus=1, num_nodes=1)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype=dtype)
    p = O.init_params(ocfg, 3)
    m.load_state_dict({n: t.to(DEV) for n, t in p.items()}, strict=False)
    m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
    batch = O.synthetic_batch(ocfg, Bn, 4, ragged_text=True)
    eng = m.engine
    eng.dropout_on = pdrop > 0
    pb = eng.bind_batch(batch["text_ids"], batch["text_masks"], batch["image"][0])
    op = eng.make_operand(pb, out=pb.patchesT_full)
    k = torch.nn.functional.normalize(torch.randn(Bn, 128, generator=torch.Generator().manual_seed(1)), dim=1)
    pb.k.copy_(k.to(DEV))
    res = {}
    for tail in (False, True):
        out = {}
        eng.pass_counter = 0                                     # the same pass seeds (= the same masks) for both forms
        for mode in (L.MODE_INFER, L.MODE_DATA, L.MODE_FULL):
            eng.zero_grads()
            eng.encoder_forward(pb, key=False, mode=mode, patchesT=op, cls_tail=tail)
            assert pb.tail[mode] == tail and pb.drop[mode][1] == pdrop
            eng.heads_forward(pb, key=False)
            out[("cls", mode)] = pb.cls.clone()
            if mode == L.MODE_INFER:
                continue
            eng.infonce(pb, 1.0 / Bn, want_dq=True)
            eng.heads_backward(pb, pb.dq, None, with_grads=mode == L.MODE_FULL)
            dpat = torch.zeros_like(pb.patchesT_full)
            eng.encoder_backward(pb, mode, op, pb.dcls, cls_only=True, dpatches=dpat)
            torch.cuda.synchronize()
            out[("dpat", mode)] = dpat.float().clone()
            if mode == L.MODE_FULL:
                out["g"] = eng.g32.clone()
        res[tail] = out
    # f32: the same arithmetic in another summation order.  bf16: the dense last block rounds its operands and activations to
    # bf16, the tail keeps fp32 - the difference is the dense pass's own rounding noise (gradients: a few % of the maximum)
    for key in res[False]:
        a, b = res[False][key], res[True][key]
        if key == "g":
            continue
        tol = 2e-5 if dtype == "f32" else (3e-2 if key[0] == "cls" else 0.15)
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + 1e-12, key
    ga, gb = res[False]["g"], res[True]["g"]
    lay = eng.layout
    specs = {n: (off, shape) for n, off, shape in eng.specs}
    for n, (off, shape) in specs.items():
        cnt = 1
        for v in shape:
            cnt *= v
        a, b = ga[off:off + cnt], gb[off:off + cnt]
        if float(a.abs().max()) == 0.0:
            continue
        assert float((a - b).abs().max()) <= (1e-4 if dtype == "f32" else 0.12) * float(a.abs().max()) + 1e-12, n
    # a full-row gradient cannot be back-propagated through buffers whose forward kept only the cls rows
    eng.encoder_forward(pb, key=False, mode=L.MODE_FULL, patchesT=op, cls_tail=True)
    with pytest.raises(L.RmclError):
        eng.encoder_backward(pb, L.MODE_FULL, op, pb.xn, cls_only=False, dpatches=None)


def test_public_infer_is_differentiable_like_the_reference():
    """vilt_module.py:275-351 is an ordinary differentiable forward: a scalar of the four returned feature tensors back-propagates
    into every query parameter; against the oracle's autograd (fp32 engine, ragged text)."""
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=3, adv_steps_img=1)
    m, p = build_module(ocfg, 5, "f32")
    batch = O.synthetic_batch(ocfg, 3, 7, ragged_text=True)
    g = torch.Generator().manual_seed(3)
    w_txt, w_img, w_cls, w_raw = (torch.randn(s, generator=g) for s in ((3, 40, 768), (3, 145, 768), (3, 768), (3, 768)))
    po = {n: t.clone().requires_grad_(not n.startswith("k_")) for n, t in p.items()}
    ro = O.infer(po, ocfg, batch["text_ids"], batch["text_masks"], batch["image"][0])
    fo = (ro["text_feats"] * w_txt).sum() + (ro["image_feats"] * w_img).sum() + (ro["cls_feats"] * w_cls).sum() + (ro["raw_cls_feats"] * w_raw).sum()
    fo.backward()
    m.zero_grad()
    r = m.infer(dev_batch(batch))
    assert r["cls_feats"].requires_grad and r["text_feats"].requires_grad
    f = (r["text_feats"] * w_txt.to(DEV)).sum() + (r["image_feats"] * w_img.to(DEV)).sum() + (r["cls_feats"] * w_cls.to(DEV)).sum() \
        + (r["raw_cls_feats"] * w_raw.to(DEV)).sum()
    assert abs(float(f) - float(fo)) < 2e-3 * max(1.0, abs(float(fo)))
    f.backward()
    params = dict(m.named_parameters())
    n_cmp = 0
    for n, t in po.items():
        if t.grad is None or n not in params or float(t.grad.norm()) < 1e-9:
            continue
        rel = float((params[n].grad.cpu() - t.grad).norm() / t.grad.norm())
        assert rel < 5e-3, (n, rel)
        n_cmp += 1
    assert n_cmp > 30
    with torch.no_grad():                                        # and the stash-free pass returns plain tensors
        assert not m.infer(dev_batch(batch))["cls_feats"].requires_grad
    assert not m.infer_k(dev_batch(batch))["cls_feats"].requires_grad   # the momentum pass never carries gradients


def test_half_batch_lanes_give_the_one_chain_step(monkeypatch):
    """B = 64, bf16: the front of the step as two half-batch chains on two streams (Engine.lanes: key forward on a third stream, K PGD
    iterations per lane, every per-sample buffer of a lane a view of the batch's) against the one-chain step.  Each sample's
    arithmetic is the same in both (row-wise kernels, the same k order in every GEMM), so the perturbation, the keys and the
    queries agree to the bit; the loss and the gradients come from the same attacked view."""
    ocfg = O.default_config(num_layers=3, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=2)
    m, p = build_module(ocfg, 5, "bf16")
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9, ragged_text=True))
    m.engine.cfg["dense_images"] = True                          # (full-size images: the lanes' precondition, known without a device read)
    out = {}
    for lanes in ("0", "1", "0"):
        monkeypatch.setenv("RMCL_LANES", lanes)
        m.zero_grad()
        m.queue_ptr = 0
        m.proj_queue.copy_(O.init_queue(ocfg, 0).to(DEV))
        m.shadow_momentum_encoder()
        loss = m.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        pb = m.engine.bufs(64)
        used = getattr(pb, "_lanes", None) is not None
        assert used == (lanes == "1") or lanes == "0"
        out.setdefault(lanes, []).append({"loss": float(loss), "delta": pb.delta.clone(), "q": pb.q.clone(), "k": pb.k.clone(),
                                          "att": pb.patchesT_full.clone(), "g": m.engine.g32.clone()})
    a, b, a2 = out["0"][0], out["1"][0], out["0"][1]
    assert getattr(m.engine.bufs(64), "_lanes", None) is not None, "the lanes did not run"
    for key in ("delta", "k", "att"):
        assert torch.equal(a[key], b[key]), key
    assert abs(a["loss"] - b["loss"]) < 1e-4 * abs(a["loss"])
    ref_noise = float((a["g"] - a2["g"]).norm() / a["g"].norm())                       # run-to-run (float atomics of the cls-row scatter)
    rel = float((a["g"] - b["g"]).norm() / a["g"].norm())
    assert rel < max(1e-5, 10 * ref_noise), (rel, ref_noise)


def test_share_clean_forward_under_dropout_is_an_opt_in_that_drops_one_pass():
    """config["share_clean_forward"] (default False): under dropout the reference draws one mask for the clean query forward
    (objectives.py:267) and another for PGD step 0 (pgd_attack_vilt.py:145), so by default both passes run ((5 + 2K) F); with the
    switch on, PGD step 0's forward also supplies the LOGGED clean prediction / q_original and the separate pass disappears - the
    passes that feed the loss (key, K PGD steps, attacked view) are the same list, drawn in the same order."""
    from rmcl_amd import _lib as L
    K = 2
    ocfg = O.default_config(num_layers=2, num_negative=1024, per_gpu_batchsize=4, adv_steps_img=K)
    logs = {}
    for share in (False, True):
        cfg = task_moco(num_layers=2, num_negative=1024, adv_steps_img=K, per_gpu_batchsize=4, drop_rate=0.1, image_view=True, text_view=False,
                        num_gpus=1, num_nodes=1, share_clean_forward=share)
        m = ViLTransformerSS(cfg, device=DEV, compute_dtype="bf16")
        m.load_state_dict({n: t.to(DEV) for n, t in O.init_params(ocfg, 5).items()}, strict=False)
        m.train()
        m.engine.pass_log = []
        from rmcl_amd.vilt.modules import vilt_utils
        vilt_utils.set_task(m)
        ret = m(dev_batch(O.synthetic_batch(ocfg, 4, 9)))
        ret["moco_loss"].backward()
        torch.cuda.synchronize()
        logs[share] = [(e["key"], e["mode"]) for e in m.engine.pass_log]
        assert torch.isfinite(ret["moco_loss"]) and ret["q_original"].shape == (4, 128) and all(e["p"] == 0.1 for e in m.engine.pass_log)
        assert "moco_attack/PGD_success_rate" in m.logged
    base = [(True, L.MODE_INFER)] + [(False, L.MODE_DATA)] * K + [(False, L.MODE_FULL)]
    assert logs[True] == base, logs[True]
    assert logs[False] == base[:1] + [(False, L.MODE_INFER)] + base[1:], logs[False]
