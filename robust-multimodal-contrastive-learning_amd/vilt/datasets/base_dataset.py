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

import io
import os
import random
from typing import Callable, Dict, List, Optional

import numpy as np
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


# ---------------------------------------------------------------------------------------------------------------------
# uint8 feed path: the batch crosses the loader's pipes and PCIe as bytes; normalisation / zero padding / patch cut run on
# the device in one kernel (csrc/embed_misc.hip u8_to_patches_kernel, Engine.bind_batch)
# ---------------------------------------------------------------------------------------------------------------------

class Uint8Batch:
    """One view of a collated batch as decoded bytes: ``data`` uint8 [B, Hmax, Wmax, 3] (HWC, every sample in its top-left
    corner, zero outside), ``sizes`` int32 [B, 2] = (h, w) of every sample (multiples of 32 after MinMaxResize).  Stands where
    the float tensor [B, 3, Hmax, Wmax] of ``collate`` stands (``batch["image"][v]``); ``float_image()`` materialises that
    tensor (the reference's values, bit for bit) for callers outside the training step."""

    def __init__(self, data: torch.Tensor, sizes: torch.Tensor):
        assert data.dtype == torch.uint8 and data.dim() == 4 and data.shape[3] == 3, tuple(data.shape)
        self.data, self.sizes = data, sizes.to(torch.int32)

    @property
    def shape(self):
        B, H, W, _ = self.data.shape
        return (B, 3, H, W)

    def to(self, device, non_blocking: bool = True) -> "Uint8Batch":
        return Uint8Batch(self.data.to(device, non_blocking=non_blocking), self.sizes)

    def pin_memory(self) -> "Uint8Batch":
        return Uint8Batch(self.data.pin_memory(), self.sizes)

    def float_image(self) -> torch.Tensor:
        from ..transforms import normalize_lut
        x = normalize_lut().to(self.data.device)[self.data.long()].permute(0, 3, 1, 2).contiguous()
        B, _, H, W = x.shape
        ys, xs = torch.arange(H, device=x.device)[None, :, None], torch.arange(W, device=x.device)[None, None, :]
        sz = self.sizes.to(x.device)
        inside = (ys < sz[:, 0, None, None]) & (xs < sz[:, 1, None, None])
        return x * inside[:, None].to(x.dtype)


class RawView:
    """What the "decode_uint8" transform returns: the DECODED image as uint8 [h, w, 3] at its original size plus the MinMaxResize it still
    owes (shorter, longer) - a loader worker then only decodes; the bicubic resize runs on the device (``RawUint8Batch``)."""
    __slots__ = ("pixels", "shorter", "longer")

    def __init__(self, pixels: torch.Tensor, shorter: int, longer: int):
        self.pixels, self.shorter, self.longer = pixels, int(shorter), int(longer)

    @property
    def shape(self):
        return tuple(self.pixels.shape)


class RawUint8Batch:
    """One view of a collated batch as decoded, NOT yet resized bytes: ``data`` uint8 [B, Hs, Ws, 3] (every sample in its top-left
    corner), ``sizes`` int32 [B, 2] = (h, w) as decoded, and the MinMaxResize parameters.  ``Engine.bind_batch`` resizes it on the device
    with PIL's own integer arithmetic (include/rmcl.h rmcl_image_resize_u8; vilt/transforms/resample.py builds the tables) into the
    ``Uint8Batch`` the byte path continues from; ``resized_on_host()`` does the same with PIL (callers outside the training step, tests)."""

    def __init__(self, data: torch.Tensor, sizes: torch.Tensor, shorter: int, longer: int, extent=None):
        """extent: (Hd, Wd) of the RESIZED batch when it must be larger than this view's own maximum - ``collate`` pads every image key
        of a batch to ONE extent (base_dataset.py:192-206), e.g. "image" and "false_image_0" of the ITM objective."""
        assert data.dtype == torch.uint8 and data.dim() == 4 and data.shape[3] == 3, tuple(data.shape)
        self.data, self.sizes, self.shorter, self.longer = data, torch.as_tensor(sizes).to(torch.int32).cpu(), int(shorter), int(longer)
        self.extent = None if extent is None else (int(extent[0]), int(extent[1]))
        B, Hs, Ws, _ = data.shape
        sz = self.sizes
        if tuple(sz.shape) != (B, 2) or bool((sz < 1).any()) or bool((sz[:, 0] > Hs).any()) or bool((sz[:, 1] > Ws).any()):
            raise ValueError(f"RawUint8Batch.sizes must be [B, 2] = (h, w) with 1 <= h <= {Hs}, 1 <= w <= {Ws} (got {sz.tolist()})")

    @property
    def target_sizes(self) -> torch.Tensor:
        """[B, 2] = (h, w) after MinMaxResize (vilt/transforms/utils.py:5-26: multiples of 32)"""
        from ..transforms import min_max_resize_size
        out = [tuple(reversed(min_max_resize_size(int(w), int(h), self.shorter, self.longer))) for h, w in self.sizes.tolist()]
        if any(h < 32 or w < 32 for h, w in out):
            raise ValueError(f"MinMaxResize({self.shorter}, {self.longer}) leaves a side below one 32-pixel patch for sizes {self.sizes.tolist()}")
        return torch.tensor(out, dtype=torch.int32)

    def out_hw(self, tgt: torch.Tensor = None):
        """(Hd, Wd) of the resized batch: this view's largest target, or the batch-wide extent given at construction"""
        t = self.target_sizes if tgt is None else tgt
        hd, wd = int(t[:, 0].max()), int(t[:, 1].max())
        return (hd, wd) if self.extent is None else (max(hd, self.extent[0]), max(wd, self.extent[1]))

    @property
    def shape(self):
        return (self.data.shape[0], 3) + self.out_hw()

    def to(self, device, non_blocking: bool = True) -> "RawUint8Batch":
        return RawUint8Batch(self.data.to(device, non_blocking=non_blocking), self.sizes, self.shorter, self.longer, self.extent)

    def pin_memory(self) -> "RawUint8Batch":
        return RawUint8Batch(self.data.pin_memory(), self.sizes, self.shorter, self.longer, self.extent)

    def tables(self):
        """PIL's integer resampling tables of every sample, packed for the device: (target sizes [B,2], hbounds [B,Wd,2], hk [B,Wd,ksh],
        vbounds [B,Hd,2], vk [B,Hd,ksv]) int32 CPU tensors.  Per-axis tables are cached per (in, out) size pair."""
        from ..transforms.resample import bicubic_coeffs_8bpc
        tgt = self.target_sizes
        B = tgt.shape[0]
        Hd, Wd = self.out_hw(tgt)
        per = [(bicubic_coeffs_8bpc(int(w), int(tw)), bicubic_coeffs_8bpc(int(h), int(th)))
               for (h, w), (th, tw) in zip(self.sizes.tolist(), tgt.tolist())]
        ksh, ksv = max(hk.shape[1] for (_, hk), _ in per), max(vk.shape[1] for _, (_, vk) in per)
        hb, hk = np.zeros((B, Wd, 2), np.int32), np.zeros((B, Wd, ksh), np.int32)
        vb, vk = np.zeros((B, Hd, 2), np.int32), np.zeros((B, Hd, ksv), np.int32)
        for b, ((bh, kh), (bv, kv)) in enumerate(per):
            hb[b, : bh.shape[0]], hk[b, : kh.shape[0], : kh.shape[1]] = bh, kh
            vb[b, : bv.shape[0]], vk[b, : kv.shape[0], : kv.shape[1]] = bv, kv
        return tgt, torch.from_numpy(hb), torch.from_numpy(hk), torch.from_numpy(vb), torch.from_numpy(vk)

    def resized_on_host(self) -> "Uint8Batch":
        from PIL import Image
        tgt = self.target_sizes
        out = torch.zeros((self.data.shape[0],) + self.out_hw(tgt) + (3,), dtype=torch.uint8)
        src = self.data.cpu().numpy()
        for b, ((h, w), (th, tw)) in enumerate(zip(self.sizes.tolist(), tgt.tolist())):
            img = Image.fromarray(src[b, :h, :w]).resize((tw, th), resample=Image.BICUBIC)
            out[b, :th, :tw] = torch.from_numpy(np.asarray(img).copy())
        return Uint8Batch(out, tgt)

    def float_image(self) -> torch.Tensor:
        return self.resized_on_host().float_image()


def select_from_sizes(sizes: torch.Tensor, gh: int, gw: int, ps: int = 32):
    """``rmcl_patch_select`` (vision_transformer.py:563-600) for a batch whose extents are KNOWN: with byte sources a pixel inside a
    sample is never exactly zero after Normalize ((2v - 255) / 255 is an odd multiple of 1/255, and so is every channel sum), so the
    pixel mask is the extent rectangle.  Returns (sel [B, gh*gw] int32: valid patches row-major, then the first non-valid patch
    repeated; counts [B]; hw [B, 2] valid patch rows / columns) - the kernel's conventions - without touching the device."""
    B, G = sizes.shape[0], gh * gw
    sel = np.zeros((B, G), dtype=np.int32)
    counts = np.zeros(B, dtype=np.int32)
    hw = np.zeros((B, 2), dtype=np.int32)
    grid = np.arange(G, dtype=np.int32).reshape(gh, gw)
    for b, (h, w) in enumerate(sizes.tolist()):
        ph, pw = min(gh, (h + ps - 1) // ps), min(gw, (w + ps - 1) // ps)
        valid = np.zeros((gh, gw), dtype=bool)
        valid[:ph, :pw] = True
        v = grid[valid]
        pad = grid[~valid]
        n = v.size
        sel[b, :n] = v
        sel[b, n:] = pad[0] if pad.size else 0
        counts[b], hw[b] = n, (ph, pw)
    return torch.from_numpy(sel), torch.from_numpy(counts), torch.from_numpy(hw)


def _stack_views_uint8(per_sample: List[List[torch.Tensor]], hmax: int, wmax: int) -> List[Uint8Batch]:
    n_views = len(per_sample[0])
    out = []
    for v in range(n_views):
        data = torch.zeros(len(per_sample), hmax, wmax, 3, dtype=torch.uint8)
        sizes = torch.zeros(len(per_sample), 2, dtype=torch.int32)
        for b, views in enumerate(per_sample):
            img = views[v]
            data[b, : img.shape[0], : img.shape[1]] = img
            sizes[b, 0], sizes[b, 1] = img.shape[0], img.shape[1]
        out.append(Uint8Batch(data, sizes))
    return out


def collate_uint8(samples: List[dict], mlm_collator: Optional[Callable] = None) -> Dict[str, object]:
    """``collate`` for samples whose image views are uint8 [h, w, 3] tensors (transform key "pixelbert_uint8"): every image key
    becomes a list of ``Uint8Batch`` (one maximum extent over ALL image keys, like the reference); text keys as in ``collate``."""
    names = {k for smp in samples for k in smp}
    image_keys = [k for k in names if "image" in k]
    stripped = [{k: v for k, v in smp.items() if k not in image_keys} for smp in samples]
    batch = collate(stripped, mlm_collator)
    if image_keys:
        shapes = [tuple(img.shape) for k in image_keys for smp in samples if smp.get(k) is not None for img in smp[k]]
        for shp in shapes:
            if len(shp) != 3 or shp[2] != 3:
                raise AssertionError(f"Collate error, a uint8 image should be in shape of (H, W, 3), instead of given {shp}")
        hmax, wmax = max(s[0] for s in shapes), max(s[1] for s in shapes)
        for k in image_keys:
            batch[k] = _stack_views_uint8([smp[k] for smp in samples], hmax, wmax)
    return batch


def collate_raw_uint8(samples: List[dict], mlm_collator: Optional[Callable] = None) -> Dict[str, object]:
    """``collate`` for samples whose image views are ``RawView`` (transform key "decode_uint8"): every image key becomes a list of
    ``RawUint8Batch`` - decoded bytes at their ORIGINAL sizes, one maximum extent over all image keys - for the device-side resize."""
    names = {k for smp in samples for k in smp}
    image_keys = [k for k in names if "image" in k]
    stripped = [{k: v for k, v in smp.items() if k not in image_keys} for smp in samples]
    batch = collate(stripped, mlm_collator)
    if image_keys:
        views = [rv for k in image_keys for smp in samples if smp.get(k) is not None for rv in smp[k]]
        for rv in views:
            if not isinstance(rv, RawView) or len(rv.shape) != 3 or rv.shape[2] != 3:
                raise AssertionError(f"Collate error, a raw view should be decoded uint8 (H, W, 3), instead of given {getattr(rv, 'shape', type(rv))}")
        hmax, wmax = max(rv.shape[0] for rv in views), max(rv.shape[1] for rv in views)
        from ..transforms import min_max_resize_size
        targets = [min_max_resize_size(rv.shape[1], rv.shape[0], rv.shorter, rv.longer) for rv in views]     # (w, h) after the resize
        extent = (max(t[1] for t in targets), max(t[0] for t in targets))                                   # one extent over ALL image keys
        for k in image_keys:
            per_sample = [smp[k] for smp in samples]
            out = []
            for v in range(len(per_sample[0])):
                data = torch.zeros(len(samples), hmax, wmax, 3, dtype=torch.uint8)
                sizes = torch.zeros(len(samples), 2, dtype=torch.int32)
                for b, vs in enumerate(per_sample):
                    px = vs[v].pixels
                    data[b, : px.shape[0], : px.shape[1]] = px
                    sizes[b, 0], sizes[b, 1] = px.shape[0], px.shape[1]
                out.append(RawUint8Batch(data, sizes, per_sample[0][v].shorter, per_sample[0][v].longer, extent))
            batch[k] = out
    return batch


# ---------------------------------------------------------------------------------------------------------------------
# arrow-table dataset: the protocol of the reference's BaseDataset (vilt/datasets/base_dataset.py:11-165)
# ---------------------------------------------------------------------------------------------------------------------

class BaseDataset(torch.utils.data.Dataset):
    """Image-text pairs out of ``{data_dir}/{name}.arrow`` tables (column "image": encoded bytes, column ``text_column_name``: list
    of captions per image), with the reference's indexing and sample protocol:

      * ``index_mapper[j] = (image row, caption number)``: one entry per caption (captions de-duplicated per image by default),
        or one entry per image with caption ``None`` for ``image_only`` / no text column (base_dataset.py:70-85);
      * ``get_image`` / ``get_false_image`` / ``get_text`` / ``get_false_text`` / ``get_suite`` return the dicts
        ``collate`` consumes (:92-165); a failing sample is replaced by a random one like the reference does (:147-164);
      * ``tokenizer``: any callable with the HF signature (the reference's datamodule attaches one loaded by name,
        datamodule_base.py:12-21 - not available offline; tests use BertTokenizer on a local vocabulary file).

    ``transform_keys``: "pixelbert" (float CHW views, the reference's), "pixelbert_uint8" (byte HWC views for the device-side
    normalisation, collate with ``collate_uint8``) or "decode_uint8" (decoded bytes at the original size: MinMaxResize too runs on the
    device, collate with ``collate_raw_uint8``).  Differences kept on purpose: the tables stay memory-mapped and the caption
    column is read once with ``to_pylist`` (no pandas round trip); de-duplication keeps first-seen order (the reference's
    ``list(set(texts))`` order depends on the hash seed)."""

    def __init__(self, data_dir: str, transform_keys: list, image_size: int, names: list, text_column_name: str = "",
                 remove_duplicate=True, max_text_len=40, draw_false_image=0, draw_false_text=0, image_only=False, max_num=-1,
                 tokenizer: Optional[Callable] = None):
        import pyarrow as pa
        from ..transforms import keys_to_transforms
        assert len(transform_keys) >= 1
        super().__init__()
        self.transforms = keys_to_transforms(transform_keys, size=image_size)
        self.text_column_name, self.names, self.max_text_len = text_column_name, names, max_text_len
        self.draw_false_image, self.draw_false_text, self.image_only, self.data_dir = draw_false_image, draw_false_text, image_only, data_dir
        self.tokenizer = tokenizer
        self.all_texts: List[List[str]] = []
        self.table = None
        if len(names) != 0:
            present = [n for n in names if os.path.isfile(f"{data_dir}/{n}.arrow")]
            tables = [pa.ipc.RecordBatchFileReader(pa.memory_map(f"{data_dir}/{n}.arrow", "r")).read_all() for n in present]
            self.table_names = [n for n, t in zip(present, tables) for _ in range(len(t))]
            self.table = pa.concat_tables(tables, promote_options="default")
            if text_column_name != "":
                texts = self.table[text_column_name].to_pylist()
                self.all_texts = [list(dict.fromkeys(t)) for t in texts] if remove_duplicate else [list(t) for t in texts]
        self.index_mapper: Dict[int, tuple] = {}
        if text_column_name != "" and not self.image_only:
            j = 0
            for i, texts in enumerate(self.all_texts[: len(self.all_texts) if max_num == -1 else max_num]):
                for _j in range(len(texts)):
                    self.index_mapper[j] = (i, _j)
                    j += 1
        elif self.table is not None:
            for i in range(len(self.table) if max_num == -1 else min(max_num, len(self.table))):
                self.index_mapper[i] = (i, None)

    @property
    def corpus(self):
        return [text for texts in self.all_texts for text in texts]

    def __len__(self):
        return len(self.index_mapper)

    # ---- sample protocol (the dict keys and the order of the RNG draws are the reference's, base_dataset.py:86-165: collate and the
    #      golden fixture tests/golden/dataset.npz depend on both; the code below is this build's own) ----
    def _decode(self, row: int, column: str):
        from PIL import Image
        return Image.open(io.BytesIO(self.table[column][row].as_py())).convert("RGB")

    def _views(self, row: int, column: str) -> list:
        """every configured transform applied to ONE decode of the image in `row`"""
        picture = self._decode(row, column)
        return [view(picture) for view in self.transforms]

    def _any_entry(self) -> int:
        """a uniformly drawn entry of the index (false samples, replacement of a broken record): ONE random.randint per call"""
        return random.randint(0, len(self.index_mapper) - 1)

    def _tokenise(self, caption: str, pad: bool):
        extra = {"padding": "max_length"} if pad else {}
        return self.tokenizer(caption, truncation=True, max_length=self.max_text_len, return_special_tokens_mask=True, **extra)

    def get_raw_image(self, index, image_key="image"):
        return self._decode(self.index_mapper[index][0], image_key)

    def get_image(self, index, image_key="image"):
        row, cap = self.index_mapper[index]
        return {"image": self._views(row, image_key), "img_index": row, "cap_index": cap, "raw_index": index}

    def get_false_image(self, rep, image_key="image"):
        row, _ = self.index_mapper[self._any_entry()]
        return {f"false_image_{rep}": self._views(row, image_key)}

    def get_text(self, raw_index):
        row, cap = self.index_mapper[raw_index]
        caption = self.all_texts[row][cap]
        return {"text": (caption, self._tokenise(caption, pad=True)), "img_index": row, "cap_index": cap, "raw_index": raw_index}

    def get_false_text(self, rep):
        row, cap = self.index_mapper[self._any_entry()]
        caption = self.all_texts[row][cap]
        return {f"false_text_{rep}": (caption, self._tokenise(caption, pad=False))}        # (the reference pads only the true caption)

    def _assemble(self, index) -> dict:
        sample = self.get_image(index)
        if not self.image_only:
            text = self.get_text(index)
            sample["replica"] = text["cap_index"] > 0             # a second / third caption of an image already seen
            sample.update(text)
        for rep in range(self.draw_false_image):                   # draws in this order: all false images, then all false texts
            sample.update(self.get_false_image(rep))
        for rep in range(self.draw_false_text):
            sample.update(self.get_false_text(rep))
        return sample

    def get_suite(self, index):
        """The sample at `index`; a record that cannot be read or decoded is reported and replaced by a randomly drawn one, again and
        again until one loads (the reference's behaviour, base_dataset.py:147-164 - a corrupt shard must not end an epoch)."""
        entry = index
        while True:
            try:
                return self._assemble(entry)
            except Exception as err:
                shard = self.names[0] if self.names else "<no shard>"
                print(f"[BaseDataset] entry {entry} of {shard} could not be loaded ({type(err).__name__}: {err}); drawing another sample")
                entry = self._any_entry()

    def __getitem__(self, index):
        return self.get_suite(index)

    def collate(self, batch, mlm_collator=None):
        views = [v for smp in batch for k, vs in smp.items() if "image" in k and vs for v in vs]
        if any(isinstance(v, RawView) for v in views):
            return collate_raw_uint8(batch, mlm_collator)
        uint8 = any(torch.is_tensor(v) and v.dtype == torch.uint8 for v in views)
        return (collate_uint8 if uint8 else collate)(batch, mlm_collator)


def write_arrow_table(path: str, images: List[bytes], captions: List[List[str]], text_column_name: str = "caption"):
    """Writes a table in the layout the reference's ``make_arrow`` scripts produce (vilt/utils/write_*.py: columns image bytes,
    caption list, image_id, split) - used by the tests and the feed benchmark to build synthetic shards."""
    import pyarrow as pa
    table = pa.table({"image": pa.array(images, type=pa.binary()), text_column_name: pa.array(captions, type=pa.list_(pa.string())),
                      "image_id": pa.array([f"{i:08d}" for i in range(len(images))]), "split": pa.array(["train"] * len(images))})
    with pa.OSFile(path, "wb") as sink:
        with pa.RecordBatchFileWriter(sink, table.schema) as writer:
            writer.write_table(table)
