"""TEST INFRASTRUCTURE (checker only: imported by tests/, never by the product path).

CPU restatement of the 8-bit bicubic resize behind the reference's ``MinMaxResize`` (vilt/transforms/utils.py:5-26 calls
``PIL.Image.resize(size, resample=Image.BICUBIC)``; pixelbert.py:9-18 puts it in front of ToTensor / Normalize).  The arithmetic
lives in a third-party dependency that is not under /root/reference: Pillow (requirements.txt:3 pins Pillow==8.2.0; this image has
12.2.0 - the resampling code of ``src/libImaging/Resample.c`` is the same in both: fixed-point 8bpc path, PRECISION_BITS = 22).
Restated here from its published algorithm, scalar and loop by loop:

  precompute_coeffs      per output index: centre = (xx + 0.5) * scale, window [centre - support, centre + support] rounded and clipped
                         to the image, weight = bicubic((x + xmin - centre + 0.5) / filterscale), weights divided by their sum
  normalize_coeffs_8bpc  weight -> int32 fixed point, rounded half away from zero, PRECISION_BITS = 32 - 8 - 2
  Horizontal / Vertical  acc = 1 << (PRECISION_BITS - 1); acc += pixel * weight; byte = clip(acc >> PRECISION_BITS, 0, 255);
                         horizontal pass first, its uint8 result is the vertical pass's input

Pinned (tests/test_feed_cpu.py): bit-identical to PIL itself on random images (up- and down-scaling, both axes, odd sizes) and to
the pixels the reference's own MinMaxResize produced for tests/golden/pipeline.npz."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size: int, out_size: int):
    """[(xmin, [int weights])] per output index."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)                     # C (int): toward zero
        xmin = max(xmin, 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = [_bicubic((x + xmin - center + 0.5) * (1.0 / filterscale)) for x in range(xmax)]
        ww = 0.0
        for w in ws:
            ww += w
        if ww != 0.0:
            ws = [w / ww for w in ws]
        fixed = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in ws]
        out.append((xmin, fixed))
    return out


def _pass(img: np.ndarray, table, axis: int) -> np.ndarray:
    """one resampling pass along `axis` (1: horizontal, 0: vertical) of a uint8 [h, w, c] image"""
    src = img.astype(np.int64)
    shape = list(img.shape)
    shape[axis] = len(table)
    dst = np.empty(shape, dtype=np.uint8)
    for o, (lo, ws) in enumerate(table):
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t, w in enumerate(ws):
            acc = acc + np.take(src, lo + t, axis=axis) * w
        val = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            dst[:, o] = val
        else:
            dst[o] = val
    return dst


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 [h, w, 3] -> uint8 [out_h, out_w, 3] like PIL.Image.resize((out_w, out_h), BICUBIC): a pass is skipped when its axis keeps
    its size (Resample.c need_horizontal / need_vertical)."""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, coeffs(w, out_w), 1)
    if out_h != h:
        img = _pass(img, coeffs(h, out_h), 0)
    return img


def min_max_resize(img: np.ndarray, shorter: int = 384, longer: int = 640) -> np.ndarray:
    """MinMaxResize (vilt/transforms/utils.py:5-26) on a uint8 [h, w, 3] array."""
    h, w = img.shape[:2]
    scale = shorter / min(w, h)
    if h < w:
        newh, neww = shorter, scale * w
    else:
        newh, neww = scale * h, shorter
    if max(newh, neww) > longer:
        scale = longer / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    newh, neww = int(newh + 0.5), int(neww + 0.5)
    newh, neww = newh // 32 * 32, neww // 32 * 32
    return resize_bicubic_u8(img, neww, newh)
