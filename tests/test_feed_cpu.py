"""Row f3 (input pipeline -> batch dict), host side: the arrow-table dataset against the reference's own BaseDataset
(tests/golden/dataset.npz + toy_shard.arrow, oracle/gen_golden.py run_dataset), and the uint8 feed path (bytes through the
loader, normalisation on the device) against the float pipeline it replaces."""
import os
import random

import numpy as np
import torch

import rmcl_pkg  # noqa: F401
from rmcl_amd.attack import word_substitution as WS
from rmcl_amd.vilt.datasets import BaseDataset, Uint8Batch, collate, collate_uint8, select_from_sizes
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
