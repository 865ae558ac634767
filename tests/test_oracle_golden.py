"""Pins the CPU oracle (oracle/rmcl_oracle.py) against the golden vectors produced by the
reference's own code (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import rmcl_oracle as O
from tests.golden_util import cfg_from_meta, digest, load

torch.set_num_threads(8)


@pytest.fixture(scope="module", params=["L2_B4_ragged", "L12_B2"])
def moco_case(request):
    g = load(f"moco_{request.param}.npz")
    cfg, B, sw, sb, ragged = cfg_from_meta(O, g["meta"])
    p = O.init_params(cfg, sw)
    batch = O.synthetic_batch(cfg, B, sb, ragged_text=ragged)
    queue = O.init_queue(cfg, 0)
    return g, cfg, p, batch, queue


def test_infer_matches_reference(moco_case):
    g, cfg, p, batch, queue = moco_case
    with torch.no_grad():
        r = O.infer(p, cfg, batch["text_ids"], batch["text_masks"], batch["image"][0])
    np.testing.assert_allclose(r["cls_feats"].numpy(), g["cls_feats"], atol=2e-5)
    np.testing.assert_allclose(r["raw_cls_feats"].numpy(), g["raw_cls_feats"], atol=5e-5)
    np.testing.assert_allclose(r["text_feats"].numpy(), g["text_feats"], atol=5e-5)
    np.testing.assert_allclose(r["image_feats"].numpy(), g["image_feats"], atol=5e-5)


def test_pgd_matches_reference(moco_case):
    g, cfg, p, batch, queue = moco_case
    k = torch.from_numpy(g["pgd_k_input"])
    pd = {n: t.detach() for n, t in p.items()}
    delta, steps = O.pgd_attack(pd, cfg, batch, k, queue, return_steps=True)
    K = cfg["adv_steps_img"]
    for kk, d in ((1, steps[0]), (K, steps[-1])):
        np.testing.assert_allclose(d[:, :, ::8, ::8].numpy(), g[f"pgd_delta_K{kk}_sub"], atol=2e-6)
        np.testing.assert_allclose(d[:, :, :32, :32].numpy(), g[f"pgd_delta_K{kk}_patch00"], atol=2e-6)
        np.testing.assert_allclose(d.flatten(1).norm(dim=1).numpy(), g[f"pgd_delta_K{kk}_persample_l2"], rtol=1e-4)
    assert float(delta.abs().max()) <= cfg["adv_max_norm_img"] + 1e-9


def test_full_step_matches_reference(moco_case):
    g, cfg, p, batch, queue = moco_case
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    ret = O.compute_moco_contrastive(p, cfg, batch, queue, 0, training=True)
    loss = ret["moco_loss"]
    assert abs(float(loss) - float(g["moco_loss"])) < 1e-3        # north_star tolerance
    loss.backward()
    B = batch["text_ids"].shape[0]
    assert ret["ptr"] == int(g["queue_ptr_after"])
    np.testing.assert_allclose(queue[:, : 2 * B].numpy(), g["queue_head_after"], atol=2e-5)
    for a in ("pos_dist", "pos_cosine", "pos_dot", "neg_dist", "neg_cosine", "neg_dot"):
        assert abs(float(ret[f"{a}_attacked_img"]) - float(g[f"ret_{a}_attacked_img"])) < 2e-4, a
    assert abs(float(ret["delta_range"]) - float(g["log_moco_attack__train__delta"])) < 1e-6
    assert abs(float(ret["pgd_success_rate"]) - float(g["log_moco_attack__PGD_success_rate"])) < 1e-6
    # EMA'd momentum parameters
    for name, dg in zip(g["ema_names"], g["ema_digest"]):
        np.testing.assert_allclose(digest(p[str(name)]), dg, rtol=1e-5, atol=1e-6)
    # parameter gradients of the training backward
    for name, dg in zip(g["grad_names"], g["grad_digest"]):
        t = p[str(name)].grad
        assert t is not None, name
        mine = digest(t)
        scale = max(dg[1], 1e-6)                     # l2 norm of that gradient tensor
        assert abs(mine[1] - dg[1]) <= 2e-3 * scale + 1e-7, (name, mine[1], dg[1])
        np.testing.assert_allclose(mine[3:], dg[3:], atol=2e-3 * dg[2] + 1e-7, err_msg=str(name))
    np.testing.assert_allclose(p["pooler.dense.weight"].grad[:8, :64].numpy(), g["grad_pooler_w"],
                               atol=2e-3 * np.abs(g["grad_pooler_w"]).max())


@pytest.mark.parametrize("tag", ["L2_B4_ragged", "L12_B2"])
def test_itm_wpa_matches_reference(tag):
    g = load(f"itm_{tag}.npz")
    cfg, B, sw, sb, ragged = cfg_from_meta(O, g["meta"], kind="itm")
    p = O.init_params(cfg, sw)
    for n, t in p.items():
        if not n.startswith("k_"):
            t.requires_grad_(True)
    batch = O.synthetic_batch(cfg, B, sb, ragged_text=ragged)
    ret = O.compute_itm_wpa(p, cfg, batch, torch.from_numpy(g["itm_labels"]))
    assert abs(float(ret["itm_loss"]) - float(g["itm_loss"])) < 1e-5
    assert abs(float(ret["itm_wpa_loss"]) - float(g["itm_wpa_loss"])) < 1e-5
    np.testing.assert_allclose(ret["itm_logits"].detach().numpy(), g["itm_logits"], atol=2e-5)
    (ret["itm_loss"] + ret["itm_wpa_loss"]).backward()
    for name, dg in zip(g["grad_names"], g["grad_digest"]):
        t = p[str(name)].grad
        if t is None:
            assert dg[1] == 0, name
            continue
        mine = digest(t)
        assert abs(mine[1] - dg[1]) <= 2e-3 * max(dg[1], 1e-6) + 1e-7, (name, mine[1], dg[1])


def test_both_views_off_raises():
    cfg = O.default_config(num_layers=1, image_view=False, text_view=False)
    with pytest.raises(ZeroDivisionError):
        O.compute_moco_contrastive({}, cfg, {}, None, 0)
