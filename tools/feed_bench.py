#!/usr/bin/env python3
"""Host feed rate of the input pipeline (row f3): image-text pairs per second that `workers` DataLoader processes deliver as
collated batches out of an arrow shard - decode (PIL), MinMaxResize to the 384 configuration (bicubic, multiples of 32),
tokenise to 40 ids, collate - for the decode-only path ("decode_uint8" + collate_raw_uint8: decoded bytes at their original size,
MinMaxResize AND normalisation on the device, round 4), the byte path ("pixelbert_uint8" + collate_uint8: resized uint8 HWC batches,
normalised on the device) and the reference's float path ("pixelbert" + collate).  `--split` times the per-image stages of one
worker in this process (decode / bicubic resize / float conversion / tokenise): what a worker's time is made of.  Synthetic shard: JPEG images of COCO's typical 640x480 /
480x640 size (smooth fields + noise, quality 90, ~100 KB each), two captions per image out of the toy vocabulary.  Never touches
the GPU (bench.py runs it as a child process BEFORE it initialises HIP, so the loader's forked workers are not GPU processes).

    python tools/feed_bench.py [--workers 4] [--images 256] [--batch 64] [--seconds 8] [--json]"""
import argparse
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import rmcl_pkg  # noqa: F401
from rmcl_amd.attack import word_substitution as WS
from rmcl_amd.vilt.datasets import BaseDataset, write_arrow_table

WORDS = "a the dog cat man woman child house street car near on in at with of and two three some".split()


def make_shard(path, n, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    images, caps = [], []
    for i in range(n):
        w, h = (640, 480) if i % 3 else (480, 640)
        yy, xx = np.mgrid[0:h, 0:w]
        a = 127 + 60 * np.sin(xx / rng.uniform(20, 60))[..., None] * np.cos(yy / rng.uniform(20, 60))[..., None] + rng.normal(0, 12, (h, w, 3))
        buf = io.BytesIO()
        Image.fromarray(a.clip(0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
        images.append(buf.getvalue())
        caps.append([" ".join(rng.choice(WORDS, size=int(rng.integers(6, 14)))) for _ in range(2)])
    write_arrow_table(path, images, caps)
    return sum(len(b) for b in images) / n


def measure(data_dir, transform, workers, batch, seconds):
    tok = WS.load_tokenizer(os.path.join(ROOT, "tests", "golden", "toy_vocab.txt"))
    ds = BaseDataset(data_dir, [transform], 384, ["feed"], text_column_name="caption", tokenizer=tok)
    # one long "epoch" (sampling with replacement): with a small shard an epoch is a handful of batches and only that many workers ever
    # run - the loader would measure its own epoch seams, not the pipeline
    sampler = torch.utils.data.RandomSampler(ds, replacement=True, num_samples=1 << 22)
    dl = torch.utils.data.DataLoader(ds, batch_size=batch, sampler=sampler, num_workers=workers, collate_fn=ds.collate, drop_last=True,
                                     persistent_workers=workers > 0, prefetch_factor=2 if workers > 0 else None)
    pairs, t0, nbytes = 0, None, 0
    while True:
        for b in dl:
            if t0 is None:                                       # (the first batch carries the workers' start-up)
                t0 = time.perf_counter()
                continue
            pairs += len(b["text"])
            im = b["image"][0]
            nbytes = (im.data if hasattr(im, "data") and hasattr(im, "sizes") else im).numel() * (1 if hasattr(im, "sizes") else 4)
            if hasattr(im, "tables"):
                im.tables()                                      # the consumer's share of the decode-only path: PIL's integer tables (cached per size pair)
            if time.perf_counter() - t0 >= seconds:
                return pairs / (time.perf_counter() - t0), nbytes
        if t0 is not None and pairs == 0 and time.perf_counter() - t0 > 4 * seconds:
            return 0.0, nbytes


def stage_split(data_dir, n=96):
    """milliseconds per image of the stages of ONE worker (this process, one thread): arrow bytes -> decode -> resize -> tensor"""
    import pyarrow as pa
    from PIL import Image
    from rmcl_amd.vilt.transforms.utils import min_max_resize_size, to_normalized_tensor
    tab = pa.ipc.RecordBatchFileReader(pa.memory_map(os.path.join(data_dir, "feed.arrow"), "r")).read_all()
    tok = WS.load_tokenizer(os.path.join(ROOT, "tests", "golden", "toy_vocab.txt"))
    rows = [tab["image"][i % len(tab)].as_py() for i in range(n)]
    caps = [tab["caption"][i % len(tab)].as_py()[0] for i in range(n)]
    t = {"decode": 0.0, "resize_bicubic": 0.0, "to_uint8_tensor": 0.0, "to_float_tensor": 0.0, "tokenise": 0.0}
    for raw, cap in zip(rows, caps):
        t0 = time.perf_counter()
        img = Image.open(io.BytesIO(raw)).convert("RGB")
        img.load()
        t1 = time.perf_counter()
        small = img.resize(min_max_resize_size(img.size[0], img.size[1], 384, 640), resample=Image.BICUBIC)
        t2 = time.perf_counter()
        torch.from_numpy(np.array(small, dtype=np.uint8))
        t3 = time.perf_counter()
        to_normalized_tensor(small)
        t4 = time.perf_counter()
        tok(cap, padding="max_length", truncation=True, max_length=40, return_special_tokens_mask=True)
        t5 = time.perf_counter()
        for k, dt in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            t[k] += dt
    return {k: round(1e3 * v / n, 3) for k, v in t.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--paths", default="decode_uint8,pixelbert_uint8,pixelbert")
    ap.add_argument("--split", action="store_true", help="also time the per-image stages of one worker (decode / resize / conversion / tokenise)")
    args = ap.parse_args()
    torch.set_num_threads(1)
    with tempfile.TemporaryDirectory() as d:
        kb = make_shard(os.path.join(d, "feed.arrow"), args.images) / 1024
        out = {"workers": args.workers, "batch": args.batch, "images_in_shard": args.images, "jpeg_kib_mean": round(kb, 1),
               "host_cores": len(os.sched_getaffinity(0)),
               "what": "DataLoader pairs/s: arrow bytes -> PIL decode -> MinMaxResize(384, 640) bicubic -> tokenise(40) -> collate, synthetic 640x480 JPEGs"}
        if args.split:
            out["ms_per_image_one_worker"] = stage_split(d)
        for tr in filter(None, args.paths.split(",")):
            rate, nbytes = measure(d, tr, args.workers, args.batch, args.seconds)
            out[tr] = {"pairs_per_s": round(rate, 1), "batch_image_bytes": int(nbytes)}
    print(json.dumps(out) if args.json else json.dumps(out, indent=1), flush=True)


if __name__ == "__main__":
    main()
