"""Coefficient tables of the DEVICE-side ``MinMaxResize``.

The reference resizes every image on the host with ``PIL.Image.resize(size, BICUBIC)`` (vilt/transforms/utils.py:5-26, called from
pixelbert.py:9-18) - after the JPEG decode the most expensive thing a loader worker does.  ``rmcl_image_resize_u8`` (csrc/embed_misc.hip)
does the same resize on the decoded bytes on the device, with PIL's OWN arithmetic for 8-bit images (Pillow ``src/libImaging/Resample.c``:
``precompute_coeffs`` -> ``normalize_coeffs_8bpc`` -> ``ImagingResampleHorizontal_8bpc`` -> ``ImagingResampleVertical_8bpc``): a separable
two-pass convolution, horizontal first, whose double-precision filter weights are normalised per output pixel and rounded ONCE to 22-bit
fixed point; each pass accumulates pixel * weight in int32 from the rounding constant 2^21, shifts by 22 and clips to a byte - the
intermediate image is uint8 again.  With the same integer tables the kernels reproduce PIL's bytes exactly (tests/test_feed_gpu.py).

This module builds the tables (float64 numpy in Resample.c's order of operations; cached per (input size, output size), a loader
sees a handful of distinct image sizes): ``bounds[xx] = (first input index, tap count)`` and ``kk[xx, :]`` = the int32 weights of output
index xx, zero-filled behind its tap count."""
from __future__ import annotations

from functools import lru_cache

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits for the result, one for the sign, one headroom bit of the accumulator
BICUBIC_SUPPORT = 2.0


def _bicubic(x: np.ndarray) -> np.ndarray:
    """Keys' cubic convolution kernel with a = -0.5 (Resample.c bicubic_filter), evaluated in its order of operations."""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def kernel_size(in_size: int, out_size: int) -> int:
    """Taps per output pixel (Resample.c: ksize = ceil(support) * 2 + 1, support = 2 * max(scale, 1))."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    return int(np.ceil(BICUBIC_SUPPORT * max(scale, 1.0))) * 2 + 1


@lru_cache(maxsize=4096)
def bicubic_coeffs_8bpc(in_size: int, out_size: int):
    """(bounds int32 [out_size, 2], kk int32 [out_size, ksize]) of one axis: Resample.c precompute_coeffs + normalize_coeffs_8bpc."""
    if in_size < 1 or out_size < 1:
        raise ValueError(f"resize of an axis of {in_size} pixels to {out_size}")
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # (double)(in1 - in0) / outSize with float box edges
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)       # (int) of a positive double = floor; negatives are clamped to 0 anyway
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    live = taps < xmax[:, None]
    w = _bicubic((taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(live, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                                                # the C loop's summation order (ww += w, tap by tap)
        ww = ww + w[:, t]
    k = np.where((ww != 0.0)[:, None], w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    k = np.where(live, k, 0.0)
    fixed = np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS))
    kk = np.trunc(fixed).astype(np.int32)                                 # (int) truncates toward zero
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    kk.setflags(write=False)
    bounds.setflags(write=False)
    return bounds, kk
