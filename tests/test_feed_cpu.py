"""Row f3 (input pipeline -> batch dict), host side: the arrow-table dataset against the reference's own BaseDataset
(tests/golden/dataset.npz + toy_shard.arrow, oracle/gen_golden.py run_dataset), and the uint8 feed path (bytes through the
loader, normalisation on the device) against the float pipeline it replaces."""
import os
import random

import numpy as np
import torch

import rmcl_pkg  # noqa: F401
from rmcl_amd.attack import word_substitution as WS
from rmcl_amd.vilt.datasets import BaseDataset, RawUint8Batch, Uint8Batch, collate, collate_uint8, select_from_sizes
from rmcl_amd.vilt.transforms import normalize_lut, pixelbert_transform, pixelbert_uint8_transform
from tests.golden_util import digest, load

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dataset(transform="pixelbert", **kw):
    tok = WS.load_tokenizer(os.path.join(GOLD, "toy_vocab.txt"))
    return BaseDataset(GOLD, [transform], 96, ["toy_shard"], text_column_name="caption", tokenizer=tok, **kw)


def test_arrow_dataset_indexing_matches_reference():
    g = load("dataset.npz")
    for tag, kw in (("dup", dict(remove_duplicate=False)), ("imgonly", dict(remove_duplicate=False, image_only=True)),
                    ("max4", dict(remove_duplicate=False, max_num=4))):
        ds = _dataset(draw_false_image=1, draw_false_text=1, **kw)
        assert len(ds) == int(g[f"{tag}_len"])
        mine = [[ds.index_mapper[j][0], -1 if ds.index_mapper[j][1] is None else ds.index_mapper[j][1]] for j in range(len(ds))]
        np.testing.assert_array_equal(np.array(mine), g[f"{tag}_index_mapper"])
    ds = _dataset(remove_duplicate=False)
    assert ds.corpus == [str(t) for t in g["corpus"]]
    dd = _dataset()                                             # default: captions de-duplicated per image
    assert len(dd) == int(g["dedup_len"]) and [len(t) for t in dd.all_texts] == g["dedup_counts"].tolist()


def test_arrow_dataset_get_suite_matches_reference():
    """image through MinMaxResize + ToTensor / Normalize, tokenised caption, replica flag, and the false image / false text draws
    (same `random` calls in the same order as base_dataset.py:109-143 under one seed)."""
    g = load("dataset.npz")
    ds = _dataset(remove_duplicate=False, draw_false_image=1, draw_false_text=1)
    random.seed(123)
    for j in (0, 1, 4, 7):
        r = ds[j]
        assert [r["img_index"], r["cap_index"], r["raw_index"], int(r["replica"])] == g[f"s{j}_meta"].tolist()
        assert r["text"][0] == str(g[f"s{j}_text"])
        assert list(r["text"][1]["input_ids"]) == g[f"s{j}_ids"].tolist() and len(r["text"][1]["input_ids"]) == 40
        assert list(r["text"][1]["attention_mask"]) == g[f"s{j}_mask"].tolist()
        assert tuple(r["image"][0].shape) == g[f"s{j}_image"].shape
        np.testing.assert_allclose(r["image"][0].numpy(), g[f"s{j}_image"], atol=1e-6)
        np.testing.assert_allclose(digest(r["false_image_0"][0]), g[f"s{j}_false_image_digest"], rtol=1e-6, atol=1e-5)
        assert r["false_text_0"][0] == str(g[f"s{j}_false_text"]) and list(r["false_text_0"][1]["input_ids"]) == g[f"s{j}_false_ids"].tolist()


def test_uint8_transform_plus_table_is_the_float_transform():
    from PIL import Image
    g = load("pipeline.npz")
    img = Image.fromarray(g["pix_src"])
    f = pixelbert_transform(size=384)(img)
    u = pixelbert_uint8_transform(size=384)(img)
    assert u.dtype == torch.uint8 and tuple(u.shape) == (f.shape[1], f.shape[2], 3)
    via_lut = normalize_lut()[u.long()].permute(2, 0, 1)
    assert torch.equal(via_lut, f)                              # bit-identical to ToTensor + Normalize
    np.testing.assert_allclose(via_lut[:, ::16, ::16].numpy(), g["pix_out_sub"], atol=1e-6)     # and pinned to the reference's pixels
    lut = normalize_lut()
    assert float(lut[0]) == -1.0 and float(lut[255]) == 1.0 and not bool((lut == 0).any())      # a valid pixel is never exactly zero


def test_collate_uint8_is_collate_in_bytes():
    """same samples through both collates: the byte batch materialises to exactly the float batch (zero padding included),
    text keys identical; one maximum extent over ALL image keys."""
    float_ds = _dataset("pixelbert", remove_duplicate=False, draw_false_image=1)
    byte_ds = _dataset("pixelbert_uint8", remove_duplicate=False, draw_false_image=1)
    idx = [0, 2, 5, 8]
    random.seed(5)
    fb = float_ds.collate([float_ds[i] for i in idx])
    random.seed(5)
    ub = byte_ds.collate([byte_ds[i] for i in idx])
    for key in ("image", "false_image_0"):
        assert isinstance(ub[key][0], Uint8Batch) and ub[key][0].shape == tuple(fb[key][0].shape)
        assert torch.equal(ub[key][0].float_image(), fb[key][0])
    assert ub["image"][0].data.shape[1:3] == ub["false_image_0"][0].data.shape[1:3]
    for key in ("text_ids", "text_masks", "text_labels"):
        assert torch.equal(ub[key], fb[key])
    assert ub["text"] == fb["text"] and ub["img_index"] == fb["img_index"]
    sz = ub["image"][0].sizes
    assert bool((sz % 32 == 0).all()) and ub["image"][0].data.numel() * 4 == fb["image"][0].numel() * 1 * 4 // 1


def test_select_from_sizes_conventions():
    sel, counts, hw = select_from_sizes(torch.tensor([[64, 96], [96, 128], [32, 32]]), 3, 4)
    assert counts.tolist() == [6, 12, 1] and hw.tolist() == [[2, 3], [3, 4], [1, 1]]
    assert sel[0, :6].tolist() == [0, 1, 2, 4, 5, 6] and sel[0, 6:].tolist() == [3] * 6      # valid row-major, then the first non-valid patch
    assert sel[1].tolist() == list(range(12)) and sel[2].tolist() == [0] + [1] * 11


# ---------------------------------------------------------------------------------------------------------------------
# MinMaxResize on the device (round 4): the integer tables and the checker's restatement of PIL's 8-bit bicubic resize
# ---------------------------------------------------------------------------------------------------------------------

RESIZE_CASES = [(451, 300), (640, 480), (480, 640), (333, 500), (1024, 200), (50, 60), (799, 801), (384, 384), (1200, 900), (97, 1001)]


def test_resample_restatement_and_tables_are_pils_arithmetic():
    """The reference resizes with PIL (vilt/transforms/utils.py:5-26 -> Image.resize(size, BICUBIC)).  (i) the checker's restatement of
    Pillow's Resample.c (oracle/pil_resample.py) reproduces PIL's bytes on up- and down-scaling cases and the reference's own pixels of
    tests/golden/pipeline.npz; (ii) the PRODUCT's integer tables (vilt/transforms/resample.py, what the device kernel eats) equal the
    restatement's, entry by entry."""
    from PIL import Image
    from oracle import pil_resample as R
    from rmcl_amd.vilt.transforms import min_max_resize_size
    from rmcl_amd.vilt.transforms.resample import bicubic_coeffs_8bpc, kernel_size
    rng = np.random.default_rng(0)
    for w, h in RESIZE_CASES:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tw, th = min_max_resize_size(w, h, 384, 640)
        ref = np.asarray(Image.fromarray(src).resize((tw, th), Image.BICUBIC))
        assert np.array_equal(R.resize_bicubic_u8(src, tw, th), ref), (w, h)
        assert np.array_equal(R.min_max_resize(src), ref), (w, h)
        for n_in, n_out in ((w, tw), (h, th)):
            bounds, kk = bicubic_coeffs_8bpc(n_in, n_out)
            assert kk.shape == (n_out, kernel_size(n_in, n_out)) and bounds.dtype == np.int32 and kk.dtype == np.int32
            for o, (lo, ws) in enumerate(R.coeffs(n_in, n_out)):
                assert bounds[o, 0] == lo and bounds[o, 1] == len(ws) and kk[o, : len(ws)].tolist() == ws and not kk[o, len(ws):].any(), (n_in, n_out, o)
    g = load("pipeline.npz")
    out = R.min_max_resize(g["pix_src"])
    t = (torch.from_numpy(out).permute(2, 0, 1).float().div(255.0) - 0.5) / 0.5
    assert tuple(t.shape) == tuple(int(v) for v in g["pix_out_shape"])
    np.testing.assert_allclose(t[:, ::16, ::16].numpy(), g["pix_out_sub"], atol=1e-6)
    np.testing.assert_allclose(digest(t), g["pix_out_digest"], rtol=1e-6, atol=1e-5)


def test_decode_only_transform_and_raw_collate():
    """transform key "decode_uint8": a worker only decodes; collate_raw_uint8 batches the decoded bytes at their ORIGINAL sizes.  Resized on
    the host with PIL (RawUint8Batch.resized_on_host) the batch is exactly the byte path's batch (pixelbert_uint8 + collate_uint8); the
    packed tables have one row per output pixel of every sample."""
    raw_ds = _dataset("decode_uint8", remove_duplicate=False, draw_false_image=1)
    byte_ds = _dataset("pixelbert_uint8", remove_duplicate=False, draw_false_image=1)
    idx = [0, 2, 5, 8]
    random.seed(5)
    rb = raw_ds.collate([raw_ds[i] for i in idx])
    random.seed(5)
    ub = byte_ds.collate([byte_ds[i] for i in idx])
    for key in ("image", "false_image_0"):
        r, u = rb[key][0], ub[key][0]
        assert isinstance(r, RawUint8Batch) and (r.shorter, r.longer) == (96, 159)
        assert torch.equal(r.target_sizes, u.sizes) and r.shape == u.shape
        h = r.resized_on_host()
        assert torch.equal(h.sizes, u.sizes) and torch.equal(h.data, u.data)
        assert torch.equal(r.float_image(), u.float_image())
        tgt, hb, hk, vb, vk = r.tables()
        assert torch.equal(tgt, u.sizes) and hb.shape[:2] == (len(idx), u.data.shape[2]) and vb.shape[:2] == (len(idx), u.data.shape[1])
        for b, ((sh, sw), (th, tw)) in enumerate(zip(r.sizes.tolist(), tgt.tolist())):
            assert int(hb[b, :tw, 1].min()) >= 1 and int((hb[b, :tw, 0] + hb[b, :tw, 1]).max()) <= sw           # taps stay inside the source row
            assert int(vb[b, :th, 1].min()) >= 1 and int((vb[b, :th, 0] + vb[b, :th, 1]).max()) <= sh
            assert abs(int(hk[b, 0].sum()) - (1 << 22)) <= 8                                                    # weights sum to one (22-bit fixed point)
    assert rb["image"][0].data.shape[1:3] == rb["false_image_0"][0].data.shape[1:3]                              # one extent over all image keys
    for key in ("text_ids", "text_masks"):
        assert torch.equal(rb[key], ub[key])
    try:
        RawUint8Batch(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), torch.tensor([[9, 4]]), 96, 159)
        assert False, "an extent outside the padded batch must be rejected"
    except ValueError:
        pass
