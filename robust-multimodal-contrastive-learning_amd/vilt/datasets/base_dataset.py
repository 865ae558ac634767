"""``BaseDataset.collate`` (vilt/datasets/base_dataset.py:167-245): list of per-sample dicts -> the batch dict the hot path
consumes (SURVEY 8b "Batch dict").  Images of different sizes are zero-padded bottom/right to the batch maximum (one tensor
per view); text entries are (string, encoding) pairs padded to the collator's length; ``*_labels`` = -100.
The arrow/PIL decode, the tokenizer and the MLM collator stay with the caller (the tokenizer is loaded by NAME in the
reference, datamodule_base.py:12-21 - not available offline): ``mlm_collator`` is any callable that maps the list of
encodings to {"input_ids", "labels"}; ``default_collator`` pads ``encoding["input_ids"]`` to ``max_text_len`` with no masking."""
from __future__ import annotations

import torch


def default_collator(max_text_len: int = 40, pad_id: int = 0):
    def fn(encodings):
        ids = torch.full((len(encodings), max_text_len), pad_id, dtype=torch.int64)
        for i, e in enumerate(encodings):
            t = torch.as_tensor(e["input_ids"], dtype=torch.int64)[:max_text_len]
            ids[i, : t.numel()] = t
        return {"input_ids": ids, "labels": torch.full_like(ids, -100)}
    return fn


def collate(batch, mlm_collator=None):
    batch_size = len(batch)
    keys = set([key for b in batch for key in b.keys()])
    dict_batch = {k: [dic[k] if k in dic else None for dic in batch] for k in keys}

    img_keys = [k for k in list(dict_batch.keys()) if "image" in k]
    img_sizes = list()
    for img_key in img_keys:
        img = dict_batch[img_key]
        img_sizes += [ii.shape for i in img if i is not None for ii in i]
    for size in img_sizes:
        assert len(size) == 3, f"Collate error, an image should be in shape of (3, H, W), instead of given {size}"
    if len(img_keys) != 0:
        max_height = max([i[1] for i in img_sizes])
        max_width = max([i[2] for i in img_sizes])
    for img_key in img_keys:
        img = dict_batch[img_key]
        view_size = len(img[0])
        new_images = [torch.zeros(batch_size, 3, max_height, max_width) for _ in range(view_size)]
        for bi in range(batch_size):
            for vi in range(view_size):
                orig = img[bi][vi]
                new_images[vi][bi, :, : orig.shape[1], : orig.shape[2]] = orig
        dict_batch[img_key] = new_images

    txt_keys = [k for k in list(dict_batch.keys()) if "text" in k]
    if len(txt_keys) != 0:
        if mlm_collator is None:
            mlm_collator = default_collator()
        encodings = [[d[1] for d in dict_batch[txt_key]] for txt_key in txt_keys]
        flatten_encodings = [e for encoding in encodings for e in encoding]
        flatten_mlms = mlm_collator(flatten_encodings)
        for i, txt_key in enumerate(txt_keys):
            texts, encs = [d[0] for d in dict_batch[txt_key]], [d[1] for d in dict_batch[txt_key]]
            mlm_ids = flatten_mlms["input_ids"][batch_size * i: batch_size * (i + 1)]
            mlm_labels = flatten_mlms["labels"][batch_size * i: batch_size * (i + 1)]
            input_ids = torch.zeros_like(mlm_ids)
            attention_mask = torch.zeros_like(mlm_ids)
            for _i, encoding in enumerate(encs):
                _input_ids = torch.tensor(encoding["input_ids"])
                _attention_mask = torch.tensor(encoding["attention_mask"])
                input_ids[_i, : len(_input_ids)] = _input_ids
                attention_mask[_i, : len(_attention_mask)] = _attention_mask
            dict_batch[txt_key] = texts
            dict_batch[f"{txt_key}_ids"] = input_ids
            dict_batch[f"{txt_key}_labels"] = torch.full_like(input_ids, -100)
            dict_batch[f"{txt_key}_ids_mlm"] = mlm_ids
            dict_batch[f"{txt_key}_labels_mlm"] = mlm_labels
            dict_batch[f"{txt_key}_masks"] = attention_mask
    return dict_batch
