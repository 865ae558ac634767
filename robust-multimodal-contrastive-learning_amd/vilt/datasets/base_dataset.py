"""Batch assembly for the hot path: ``collate(samples, mlm_collator)`` turns per-sample dicts into the batch dict that
``ViLTransformerSS.training_step`` consumes (SURVEY 8b "Batch dict"; behaviour of BaseDataset.collate,
vilt/datasets/base_dataset.py:167-245).

  * every key containing "image" holds, per sample, a list of views [3, H, W]; view v of the batch becomes ONE tensor
    [B, 3, Hmax, Wmax], zero-filled, each sample in its top-left corner (bottom / right padding).  The on-device ragged
    ``visual_embed`` recovers the per-sample extent from the zeros.
  * every key containing "text" holds (string, encoding) pairs; it expands to <key> (strings), <key>_ids, <key>_masks,
    <key>_labels (= -100), <key>_ids_mlm, <key>_labels_mlm, all [B, collator length].
  * everything else is passed through as a list.

Arrow / PIL decoding, the tokenizer and the masked-LM collator belong to the caller (the reference loads the tokenizer by
name, datamodule_base.py:12-21 - not available offline).  ``mlm_collator`` maps the flat list of encodings to
{"input_ids", "labels"}; ``default_collator`` pads ``encoding["input_ids"]`` to ``max_text_len`` and masks nothing."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch


def default_collator(max_text_len: int = 40, pad_id: int = 0) -> Callable:
    def pad(encodings: List[dict]) -> Dict[str, torch.Tensor]:
        ids = torch.full((len(encodings), max_text_len), pad_id, dtype=torch.int64)
        for row, enc in zip(ids, encodings):
            tok = torch.as_tensor(enc["input_ids"], dtype=torch.int64)[:max_text_len]
            row[: tok.numel()] = tok
        return {"input_ids": ids, "labels": torch.full_like(ids, -100)}
    return pad


def _stack_views(per_sample: List[List[torch.Tensor]], hmax: int, wmax: int) -> List[torch.Tensor]:
    """[sample][view] tensors [3, h, w]  ->  [view] tensors [B, 3, hmax, wmax], zero padded bottom / right."""
    n_views = len(per_sample[0])
    out = [torch.zeros(len(per_sample), 3, hmax, wmax) for _ in range(n_views)]
    for b, views in enumerate(per_sample):
        for v, img in enumerate(views):
            out[v][b, :, : img.shape[1], : img.shape[2]] = img
    return out


def _expand_text(name: str, pairs: List[tuple], mlm_ids: torch.Tensor, mlm_labels: torch.Tensor) -> Dict[str, object]:
    ids = torch.zeros_like(mlm_ids)
    masks = torch.zeros_like(mlm_ids)
    for b, (_, enc) in enumerate(pairs):
        tok, att = torch.tensor(enc["input_ids"]), torch.tensor(enc["attention_mask"])
        ids[b, : tok.numel()] = tok
        masks[b, : att.numel()] = att
    return {name: [s for s, _ in pairs], f"{name}_ids": ids, f"{name}_masks": masks, f"{name}_labels": torch.full_like(ids, -100),
            f"{name}_ids_mlm": mlm_ids, f"{name}_labels_mlm": mlm_labels}


def collate(samples: List[dict], mlm_collator: Optional[Callable] = None) -> Dict[str, object]:
    B = len(samples)
    names = {k for smp in samples for k in smp}
    batch: Dict[str, object] = {k: [smp.get(k) for smp in samples] for k in names}

    image_keys = [k for k in names if "image" in k]
    if image_keys:
        shapes = [tuple(img.shape) for k in image_keys for views in batch[k] if views is not None for img in views]
        for shp in shapes:
            if len(shp) != 3:
                raise AssertionError(f"Collate error, an image should be in shape of (3, H, W), instead of given {shp}")
        hmax, wmax = max(s[1] for s in shapes), max(s[2] for s in shapes)      # one maximum over ALL image keys, like the reference
        for k in image_keys:
            batch[k] = _stack_views(batch[k], hmax, wmax)

    text_keys = [k for k in names if "text" in k]
    if text_keys:
        collator = mlm_collator or default_collator()
        flat = collator([enc for k in text_keys for _, enc in batch[k]])        # key-major, B encodings per key
        for i, k in enumerate(text_keys):
            rows = slice(B * i, B * (i + 1))
            batch.update(_expand_text(k, batch[k], flat["input_ids"][rows], flat["labels"][rows]))
    return batch
