"""-m gpu: Barlow-Twins variant - the HIP head / loss kernels against torch in fp64, and the whole step
(clean projection, PGD on the cross-correlation loss, attacked view, backward) against the reference's own run
(tests/golden/barlow_*.npz from oracle/gen_golden.py run_barlow) and against the oracle at another seed."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import rmcl_pkg  # noqa: F401,E402
from oracle import rmcl_oracle as O  # noqa: E402
from rmcl_amd import _lib as L  # noqa: E402
from rmcl_amd._lib import lib, check, P  # noqa: E402
from rmcl_amd.runtime import bt_layout, stream_ptr  # noqa: E402
from rmcl_amd.vilt.config import task_barlowtwins  # noqa: E402
from rmcl_amd.vilt.modules import ViLTransformerSS  # noqa: E402
from tests.golden_util import digest, load  # noqa: E402
from tests.test_path_gpu import dev_batch  # noqa: E402

DEV = "cuda:0"
C = L.C
F = C.c_float


def torch_head(w, x, training=True, running=None):
    """the same head in torch (fp64): Linear - BN - ReLU - Linear - BN - ReLU - Linear - BN(affine=False)"""
    bn = torch.nn.functional.batch_norm
    r = running or [None] * 6
    h = x @ w["w1"].t()
    h = torch.relu(bn(h, r[0], r[1], w["g1"], w["b1"], training, 0.1, 1e-5))
    h = h @ w["w2"].t()
    h = torch.relu(bn(h, r[2], r[3], w["g2"], w["b2"], training, 0.1, 1e-5))
    h = h @ w["w3"].t()
    return bn(h, r[4], r[5], None, None, training, 0.1, 1e-5)


@pytest.mark.parametrize("B,dims", [(64, (768, 1024, 512, 768)), (8, (768, 8192, 8192, 8192))])
def test_bt_head_and_loss_match_torch_fp64(B, dims):
    D, H1, H2, H3 = dims
    cfg = {"hidden_size": D, "barlowtwins_dims": (H1, H2, H3)}
    bt, specs, n = bt_layout(cfg, 0)
    g = torch.Generator().manual_seed(5)
    arena = torch.zeros(n)
    w = {}
    for (name, off, shape), key in zip(specs, ("w1", "g1", "b1", "w2", "g2", "b2", "w3")):
        t = (torch.rand(shape, generator=g) * 2 - 1) / shape[1] ** 0.5 if len(shape) == 2 else \
            (1 + 0.1 * torch.randn(shape, generator=g) if key[0] == "g" else 0.1 * torch.randn(shape, generator=g))
        arena[off:off + t.numel()] = t.flatten()
        w[key] = t.double().requires_grad_(True)
    arena = arena.to(DEV)
    grads = torch.zeros_like(arena)
    x = torch.randn(B, D, generator=g)
    zk = torch.randn(B, H3, generator=g)
    xd = x.double().requires_grad_(True)
    running = torch.cat([torch.zeros(H1), torch.ones(H1), torch.zeros(H2), torch.ones(H2), torch.zeros(H3), torch.ones(H3)]).to(DEV)
    stash = torch.empty(int(lib.rmcl_bt_stash_floats(C.byref(bt), B)), device=DEV)
    z = torch.empty(B, H3, device=DEV)
    check(lib.rmcl_bt_head_forward(C.byref(bt), P(arena), P(x.to(DEV)), B, 1, P(running), F(0.1), P(stash), P(z), stream_ptr()))
    tr = [torch.zeros(H1).double(), torch.ones(H1).double(), torch.zeros(H2).double(), torch.ones(H2).double(), torch.zeros(H3).double(),
          torch.ones(H3).double()]
    zt = torch_head(w, xd, True, tr)
    assert float((z.cpu().double() - zt).abs().max()) < 2e-3                     # BatchNorm-normalised values, O(1)
    o = 0
    for i, n_ in enumerate((H1, H2, H3)):
        assert torch.allclose(running[o:o + n_].cpu().double(), tr[2 * i], atol=1e-5)
        assert torch.allclose(running[o + n_:o + 2 * n_].cpu().double(), tr[2 * i + 1], rtol=1e-4, atol=1e-6)
        o += 2 * n_
    # loss + gradient
    lam, bs = 0.0051, float(2 * B)
    c = zt.t() @ zk.double() / bs
    dg = torch.diagonal(c)
    on, off = ((dg - 1) ** 2).sum(), (c ** 2).sum() - (dg ** 2).sum()
    loss = on + lam * off
    loss.backward()
    cbuf = torch.empty(H3, H3, device=DEV)
    ws = torch.empty(int(lib.rmcl_bt_loss_ws_floats(H3)), device=DEV)
    loss2 = torch.empty(2, device=DEV)
    dz = torch.empty(B, H3, device=DEV)
    zkd = zk.to(DEV)
    check(lib.rmcl_bt_corr(P(z), P(zkd), B, H3, F(1.0 / bs), P(cbuf), stream_ptr()))
    check(lib.rmcl_bt_loss(P(cbuf), H3, F(lam), F(1.0), P(ws), P(loss2), stream_ptr()))
    check(lib.rmcl_bt_dz(P(zkd), P(cbuf), B, H3, F(1.0 / bs), P(dz), stream_ptr()))
    assert abs(float(loss2[0]) - float(on)) < 1e-4 * float(on) and abs(float(loss2[1]) - float(off)) < 1e-4 * float(off)
    dcls = torch.empty(B, D, device=DEV)
    check(lib.rmcl_bt_head_backward(C.byref(bt), P(arena), P(stash), P(dz), B, 1, P(grads), P(dcls), stream_ptr()))
    torch.cuda.synchronize()
    ref_dx = xd.grad
    assert float((dcls.cpu().double() - ref_dx).abs().max()) < 2e-3 * float(ref_dx.abs().max())
    for (name, off_, shape), key in zip(specs, ("w1", "g1", "b1", "w2", "g2", "b2", "w3")):
        got = grads[off_:off_ + w[key].numel()].view(shape).cpu().double()
        assert float((got - w[key].grad).abs().max()) < 2e-3 * float(w[key].grad.abs().max()), name
    # pair metrics
    rows = torch.empty(B, 3, device=DEV)
    check(lib.rmcl_bt_pair_metrics(P(z), P(zkd), B, H3, P(rows), stream_ptr()))
    zc = z.cpu().double()
    want = torch.stack([(zc - zk.double()).norm(dim=1), torch.nn.functional.cosine_similarity(zc, zk.double(), dim=1, eps=1e-6),
                        (zc * zk.double()).sum(1)], 1)
    assert torch.allclose(rows.cpu().double(), want, rtol=1e-4, atol=1e-4)
    # eval mode: running statistics, no update
    run0 = running.clone()
    check(lib.rmcl_bt_head_forward(C.byref(bt), P(arena), P(x.to(DEV)), B, 0, P(running), F(0.1), P(stash), P(z), stream_ptr()))
    with torch.no_grad():
        ze = torch_head({k: v.detach() for k, v in w.items()}, x.double(), False, tr)
    assert float((z.cpu().double() - ze).abs().max()) < 2e-3 and torch.equal(run0, running)


def make_bt_module(ocfg, seed_w, seed_h, dtype="f32", **over):
    cfg = task_barlowtwins(num_layers=ocfg["num_layers"], adv_steps_img=ocfg["adv_steps_img"], per_gpu_batchsize=ocfg["per_gpu_batchsize"],
                           drop_rate=0.0, image_view=True, text_view=False, num_gpus=1, num_nodes=1,
                           barlowtwins_dims=tuple(ocfg["barlowtwins_dims"]), adv_lr=ocfg["adv_lr"], **over)
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype=dtype)
    p = O.init_params(ocfg, seed_w)
    p.update(O.bt_init_params(ocfg, seed_h))
    sd = {n: t.to(DEV) for n, t in p.items() if not n.startswith("k_") and not n.startswith("moco_head") and not n.startswith("itm_score")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("running" in k or "num_batches" in k for k in missing), (missing, unexpected)
    m.train()
    return m, p


@pytest.mark.parametrize("tag", ["L2_B4_ragged", "L2_B4_wide"])
def test_barlowtwins_step_matches_reference_golden(tag):
    g = load(f"barlow_{tag}.npz")
    B, sw, sb, ragged, L_, K, sh, h1, h2, h3 = [int(x) for x in g["meta"]]
    ocfg = O.default_config(num_layers=L_, num_negative=1024, per_gpu_batchsize=B, adv_steps_img=K, barlowtwins_dims=(h1, h2, h3),
                            image_view=True, text_view=False)
    m, p = make_bt_module(ocfg, sw, sh, "f32")
    batch = dev_batch(O.synthetic_batch(ocfg, B, sb, ragged_text=bool(ragged)))
    m.zero_grad()
    loss = m.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    narrow = tag == "L2_B4_ragged"
    # training_step sums every returned "...loss..." value (vilt_module.py:475): 2x the Barlow-Twins loss, gradients included
    assert abs(float(loss) - float(g["total_loss"])) < 1e-3 * float(g["total_loss"])            # north_star: 1e-3 on the loss
    assert abs(float(m.logged["barlowtwins/train/loss"]) - float(g["barlowtwins_loss"])) < 1e-3 * float(g["barlowtwins_loss"])
    eng = m.engine
    # the PGD delta the attacker returned (pb.delta, patch layout -> image)
    pb = eng.bufs(B, "bt")
    delta = eng.patches_to_image(pb.delta, pb).cpu()
    np.testing.assert_allclose(delta[:, :, ::8, ::8].numpy(), g["delta_sub"], atol=5e-5 if narrow else 2e-4)
    # projections: BatchNorm over 4 samples amplifies fp32 rounding in single features (tests/test_oracle_golden.py quantifies it
    # with an fp64 run of the restatement): bound max and mean
    dk = np.abs(eng.bt_bufs(B, "k").z.cpu().numpy() - g["k"])
    dq = np.abs(eng.bt_bufs(B, "q_img").z.cpu().numpy() - g["q_image"])
    assert dk.max() < 5e-3 and dk.mean() < 1e-4, (dk.max(), dk.mean())
    assert dq.max() < (2e-2 if narrow else 0.3) and dq.mean() < (1e-3 if narrow else 5e-3), (dq.max(), dq.mean())
    logged = m.logged
    assert abs(float(logged["barlowtwins/train/barlowtwins_loss_invariance_img"]) - float(g["ret_barlowtwins_loss_invariance_img"])) \
        < 1e-3 * float(g["ret_barlowtwins_loss_invariance_img"])
    assert abs(float(logged["barlowtwins/train/barlowtwins_loss_redundancy_img"]) - float(g["ret_barlowtwins_loss_redundancy_img"])) \
        < 1e-3 * float(g["ret_barlowtwins_loss_redundancy_img"])
    for kind in ("L2", "Cosine", "Dot"):
        key = {"L2": "pos_dist", "Cosine": "pos_cosine", "Dot": "pos_dot"}[kind] + "_attacked_img"
        assert abs(float(logged[f"barlowtwins_dist_train_{kind}/Pos_attacked_img"]) - float(g["ret_" + key])) < 2e-3 * max(1.0, abs(float(g["ret_" + key])))
    assert abs(float(logged["barlowtwins_attack/train/delta"]) - float(g["log_barlowtwins_attack__train__delta"])) < 1e-5
    # gradients: l2 norm of every tensor the reference has a gradient for
    params = dict(m.named_parameters())
    gtol = 2e-3 if narrow else 1e-2
    for n, d in zip(g["grad_names"], g["grad_digest"]):
        got = digest(params[str(n)].grad)
        assert abs(got[1] - d[1]) <= gtol * d[1] + 1e-7, (n, got[:3], d[:3])
    w1g = params["barlowtwins_head.projector.0.weight"].grad[:8, :64].cpu().numpy()
    np.testing.assert_allclose(w1g, g["grad_bt_w1"], atol=(5e-3 if narrow else 0.2) * np.abs(g["grad_bt_w1"]).max())
    np.testing.assert_allclose(params["pooler.dense.weight"].grad[:8, :64].cpu().numpy(), g["grad_pooler_w"],
                               atol=(5e-3 if narrow else 5e-2) * np.abs(g["grad_pooler_w"]).max())
    # BatchNorm buffers under the reference's state-dict names
    sd = m.state_dict()
    for key in ("projector.1", "projector.4", "norm"):
        kk = key.replace(".", "__")
        np.testing.assert_allclose(sd[f"barlowtwins_head.{key}.running_mean"].cpu().numpy(), g[f"buf_{kk}__running_mean"], atol=1e-4)
        np.testing.assert_allclose(sd[f"barlowtwins_head.{key}.running_var"].cpu().numpy(), g[f"buf_{kk}__running_var"], rtol=1e-3, atol=1e-6)
        assert int(sd[f"barlowtwins_head.{key}.num_batches_tracked"]) == int(g[f"buf_{kk}__num_batches_tracked"]) == 2


def test_barlowtwins_bs64_bf16_step_and_optimizer():
    """The variant at the benchmark's size: 12 layers, bs=64, bf16 encoder, the reference's 8192-wide head: the loss tracks the
    oracle-free invariants (finite, decreases over a few AdamW steps on a fixed batch) and every head tensor receives a gradient."""
    ocfg = O.default_config(num_layers=12, num_negative=1024, per_gpu_batchsize=64, adv_steps_img=1, barlowtwins_dims=(8192, 8192, 8192),
                            image_view=True, text_view=False)
    m, p = make_bt_module(ocfg, 3, 4, "bf16", max_steps=100, warmup_steps=0, learning_rate=1e-4)
    (opt,), _ = m.configure_optimizers()
    batch = dev_batch(O.synthetic_batch(ocfg, 64, 9))
    losses = []
    for it in range(4):
        m.zero_grad()
        loss = m.training_step(batch, it)
        loss.backward()
        if it == 0:
            params = dict(m.named_parameters())
            for n in ("0.weight", "1.weight", "1.bias", "3.weight", "4.weight", "4.bias", "6.weight"):
                gr = params["barlowtwins_head.projector." + n].grad
                assert torch.isfinite(gr).all() and float(gr.abs().max()) > 0, n
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert int(m.state_dict()["barlowtwins_head.norm.num_batches_tracked"]) == 8


def test_barlowtwins_three_views_with_text_attack_match_reference_golden():
    """text_view + image_view: the reference's compute_barlowtwins_contrastive with its GreedyAttack_barlowtwins (word level, toy
    resources) and PGDAttack_bartlowtwins, loss sum of training_step, backward - attacked sentences, per-loop decisions,
    losses of the three views, distance logs, gradient norms, BatchNorm buffers (7 head calls tracked)."""
    from rmcl_amd.attack import word_substitution as WS
    g = load("barlow3_L2_B4.npz")
    B, sw, sb, L_, K, sh, h1, h2, h3, loops, n_cand = [int(x) for x in g["meta"]]
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ocfg = O.default_config(num_layers=L_, num_negative=1024, per_gpu_batchsize=B, adv_steps_img=K, barlowtwins_dims=(h1, h2, h3),
                            image_view=True, text_view=True)
    cfg = task_barlowtwins(num_layers=L_, adv_steps_img=K, per_gpu_batchsize=B, drop_rate=0.0, image_view=True, text_view=True, num_gpus=1,
                           num_nodes=1, barlowtwins_dims=(h1, h2, h3), adv_lr=ocfg["adv_lr"], max_loops=loops, n_candidates=n_cand,
                           tokenizer=os.path.join(gold, "toy_vocab.txt"), embedding_path=os.path.join(gold, "toy_counter_fitted.txt"),
                           sim_path=None, stopwords=os.path.join(gold, "toy_stopwords.txt"))
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="f32")
    p = O.init_params(ocfg, sw)
    p.update(O.bt_init_params(ocfg, sh))
    m.load_state_dict({n: t.to(DEV) for n, t in p.items() if not n.startswith(("k_", "moco_head", "itm_score"))}, strict=False)
    m.train()
    table = m.greedy_attacker.synonyms
    ref_order = {str(w): str(c).split("|") for w, c in zip(g["syn_words"], g["syn_cands"])}
    for w, c in ref_order.items():
        assert set(table(w)) == set(c)

    class RefOrder:                                                    # the reference's set-iteration order of the same candidates
        word2id = table.word2id
        __contains__ = lambda self, w: w in table.word2id
        __call__ = lambda self, w: ref_order.get(w, [w])

    m.greedy_attacker.synonyms = RefOrder()
    batch = dev_batch(O.synthetic_batch(ocfg, B, sb))
    batch["text"] = [str(t) for t in g["text_in"]]
    batch["text_ids"] = torch.from_numpy(g["text_ids_in"]).to(DEV)
    batch["text_masks"] = torch.from_numpy(g["text_masks_in"]).to(DEV)
    m.zero_grad()
    loss = m.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    att = m.greedy_attacker
    for li, (replace_idx, new_text, all_num, best) in enumerate(att.trace):
        assert [-1 if x is None else x for x in replace_idx] == g["replace_idx"][li].tolist(), li
        assert new_text == [str(t) for t in g[f"new_text_{li}"]], li
        # index 0 and -1 both mean "keep the sentence" (only an index > 0 is accepted, greedy_attack_vilt.py:568); a candidate
        # equal to the original sentence scores an exact tie in exact arithmetic, so the two are not distinguished here
        assert [max(j, 0) for j in best] == [max(int(j), 0) for j in g["best_idx"][li]], li
    assert abs(float(loss) - float(g["total_loss"])) < 1e-3 * float(g["total_loss"])
    lg = m.logged
    assert abs(float(lg["barlowtwins/train/loss"]) - float(g["ret_barlowtwins_loss"])) < 1e-3 * float(g["ret_barlowtwins_loss"])
    for name in ("text", "img", "both"):
        for part in ("invariance", "redundancy"):
            ref = float(g[f"ret_barlowtwins_loss_{part}_{name}"])
            assert abs(float(lg[f"barlowtwins/train/barlowtwins_loss_{part}_{name}"]) - ref) < 2e-3 * ref, (name, part)
    for suffix in ("txt", "img", "both"):
        for kind, key in (("L2", "pos_dist"), ("Cosine", "pos_cosine"), ("Dot", "pos_dot")):
            ref = float(g[f"ret_{key}_attacked_{suffix}"])
            assert abs(float(lg[f"barlowtwins_dist_train_{kind}/Pos_attacked_{suffix}"]) - ref) < 3e-3 * max(1.0, abs(ref)), (suffix, kind)
    assert abs(float(lg["barlowtwins_attack/train/num_changes"]) - float(g["log_barlowtwins_attack__train__num_changes"])) < 1e-9
    params = dict(m.named_parameters())
    rel = {str(n): abs(digest(params[str(n)].grad)[1] - d[1]) / (d[1] + 1e-12) for n, d in zip(g["grad_names"], g["grad_digest"])}
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:6]
    # Three BatchNorms over 4 samples make the gradient's overall scale ill-conditioned: the oracle restatement in fp64 differs
    # from its own fp32 run by 1.0 % (uniformly over all tensors) and from the reference by 0.7 %; HIP lands 1.4 % from the
    # reference.  The well-conditioned text view alone is pinned to 1e-3 below.
    assert worst[0][1] <= 2.5e-2, worst
    sd = m.state_dict()
    for key in ("projector.1", "projector.4", "norm"):
        kk = key.replace(".", "__")
        np.testing.assert_allclose(sd[f"barlowtwins_head.{key}.running_mean"].cpu().numpy(), g[f"buf_{kk}__running_mean"], atol=2e-4)
        np.testing.assert_allclose(sd[f"barlowtwins_head.{key}.running_var"].cpu().numpy(), g[f"buf_{kk}__running_var"], rtol=2e-3, atol=1e-6)
        assert int(sd[f"barlowtwins_head.{key}.num_batches_tracked"]) == int(g[f"buf_{kk}__num_batches_tracked"]) == 4


def test_barlowtwins_text_view_backward_matches_oracle():
    """The text view alone (attacked sentences taken from the reference fixture, no PGD in the chain): loss and the gradient of
    every tensor against the oracle restatement - this view is well conditioned, so the bound is tight."""
    g = load("barlow3_L2_B4.npz")
    B, sw, sb, L_, K, sh, h1, h2, h3, loops, n_cand = [int(x) for x in g["meta"]]
    ocfg = O.default_config(num_layers=L_, num_negative=1024, per_gpu_batchsize=B, adv_steps_img=K, barlowtwins_dims=(h1, h2, h3),
                            image_view=False, text_view=True)
    ids, masks = torch.from_numpy(g["text_ids_in"]), torch.from_numpy(g["text_masks_in"])
    tids, tmasks = torch.from_numpy(g["text_ids_out"]), torch.from_numpy(g["text_masks_out"])
    cfg = task_barlowtwins(num_layers=L_, adv_steps_img=K, per_gpu_batchsize=B, drop_rate=0.0, image_view=False, text_view=True, num_gpus=1,
                           num_nodes=1, barlowtwins_dims=(h1, h2, h3), adv_lr=ocfg["adv_lr"])
    m = ViLTransformerSS(cfg, device=DEV, compute_dtype="f32")
    p = O.init_params(ocfg, sw)
    p.update(O.bt_init_params(ocfg, sh))
    m.load_state_dict({n: t.to(DEV) for n, t in p.items() if not n.startswith(("k_", "moco_head", "itm_score"))}, strict=False)
    m.train()

    class Fixed:                                                       # the attack's outcome, so that only the view is under test
        def adv_attack_samples(self, pl_module, batch, k):
            return {"txt_input_ids": tids.to(DEV), "text_masks": tmasks.to(DEV), "text": batch["text"], "num_changes": 0.0, "change_rate": 0.0}

    m.greedy_attacker = Fixed()
    batch0 = O.synthetic_batch(ocfg, B, sb)
    batch = dev_batch(batch0)
    batch["text_ids"], batch["text_masks"] = ids.to(DEV), masks.to(DEV)
    m.zero_grad()
    loss = m.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    run = O.bt_running_init(ocfg)
    img = batch0["image"][0]
    with torch.no_grad():
        k = O.barlowtwins_head(p, O.infer(p, ocfg, ids, masks, img)["cls_feats"], run, True)
    q = O.barlowtwins_head(p, O.infer(p, ocfg, tids, tmasks, img)["cls_feats"], run, True)
    lo = O.barlow_loss(q, k, float(B), ocfg["adv_lr"])[0]
    (2 * lo).backward()                                                # training_step's sum: loss + its two components
    assert abs(float(loss) - 2 * float(lo)) < 1e-4 * 2 * float(lo)
    params = dict(m.named_parameters())
    for n, t in p.items():
        if t.grad is not None and n in params:
            a, b = digest(params[n].grad)[1], digest(t.grad)[1]
            assert abs(a - b) <= 1e-3 * b + 1e-7, (n, a, b)


def test_barlowtwins_validation_step_uses_running_statistics():
    """validation (module.eval()): every BatchNorm of the head - in the clean projection, in the PGD's deep copy and in the
    attacked view - normalises with the running estimates and leaves them untouched; loss and logs against the oracle."""
    g = load("barlow_L2_B4_ragged.npz")
    B, sw, sb, ragged, L_, K, sh, h1, h2, h3 = [int(x) for x in g["meta"]]
    ocfg = O.default_config(num_layers=L_, num_negative=1024, per_gpu_batchsize=B, adv_steps_img=K, barlowtwins_dims=(h1, h2, h3),
                            image_view=True, text_view=False)
    m, p = make_bt_module(ocfg, sw, sh, "f32")
    batch0 = O.synthetic_batch(ocfg, B, sb, ragged_text=bool(ragged))
    gen = torch.Generator().manual_seed(8)
    run = O.bt_running_init(ocfg)
    for key in ("projector.1", "projector.4", "norm"):                   # non-trivial running estimates
        run[f"barlowtwins_head.{key}.running_mean"].copy_(0.1 * torch.randn(run[f"barlowtwins_head.{key}.running_mean"].shape, generator=gen))
        run[f"barlowtwins_head.{key}.running_var"].copy_(0.5 + torch.rand(run[f"barlowtwins_head.{key}.running_var"].shape, generator=gen))
    m.load_state_dict({n: t.to(DEV) for n, t in run.items()}, strict=False)
    before = {n: t.clone() for n, t in m.state_dict().items() if "running" in n or "num_batches" in n}
    m.eval()
    out = m.validation_step(dev_batch(batch0), 0)
    torch.cuda.synchronize()
    for n, t in before.items():
        assert torch.equal(m.state_dict()[n], t), n
    ref = O.compute_barlowtwins_contrastive(p, ocfg, batch0, {n: t.clone() for n, t in run.items()}, training=False)
    lo = float(ref["barlowtwins_loss"])
    assert abs(float(out["barlowtwins_loss"]) - lo) < 1e-3 * lo
    assert abs(float(m.logged["barlowtwins/val/barlowtwins_loss_invariance_img"]) - float(ref["barlowtwins_loss_invariance_img"])) \
        < 1e-3 * float(ref["barlowtwins_loss_invariance_img"])
    delta = m.engine.patches_to_image(m.engine.bufs(B, "bt").delta, m.engine.bufs(B, "bt")).cpu()
    np.testing.assert_allclose(delta[:, :, ::8, ::8].numpy(), ref["delta"][:, :, ::8, ::8].numpy(), atol=5e-5)
