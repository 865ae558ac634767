"""Image side of the input pipeline, host side, PIL + torch only (torchvision is not required):
``MinMaxResize`` (behaviour of vilt/transforms/utils.py:5-26) and the ``pixelbert`` transform (resize -> [0,1] tensor ->
(x - 0.5) / 0.5, vilt/transforms/pixelbert.py:9-17).  Every output side is a multiple of 32 - what the on-device ragged
``visual_embed`` expects of a zero-padded batch."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image


def min_max_resize_size(w: int, h: int, shorter: int = 800, longer: int = 1333):
    """Target (width, height): the short side goes to `shorter`, then both shrink if the long side would pass `longer`,
    then round half up and floor to multiples of 32.  Float arithmetic in the reference's order of operations (the rounding
    of int(x + 0.5) depends on it)."""
    k = shorter / min(w, h)
    tw, th = (k * w, shorter) if h < w else (shorter, k * h)
    big = max(th, tw)
    if big > longer:
        k2 = longer / big
        th, tw = th * k2, tw * k2
    snap = lambda v: int(v + 0.5) // 32 * 32
    return snap(tw), snap(th)


class MinMaxResize:
    """callable(PIL image) -> PIL image, bicubic."""

    def __init__(self, shorter=800, longer=1333):
        self.min, self.max = shorter, longer

    def __call__(self, x: Image.Image) -> Image.Image:
        return x.resize(min_max_resize_size(x.size[0], x.size[1], self.min, self.max), resample=Image.BICUBIC)


def to_normalized_tensor(img: Image.Image) -> torch.Tensor:
    """uint8 HWC -> float32 CHW in [-1, 1]  (ToTensor followed by Normalize(mean 0.5, std 0.5))."""
    chw = torch.from_numpy(np.array(img.convert("RGB"), dtype=np.uint8)).permute(2, 0, 1)
    return chw.to(torch.float32).div_(255.0).sub_(0.5).div_(0.5)


def pixelbert_transform(size=800):
    resize = MinMaxResize(shorter=size, longer=int((1333 / 800) * size))
    return lambda img: to_normalized_tensor(resize(img))


def normalize_lut() -> torch.Tensor:
    """The 256 values ToTensor + Normalize(0.5, 0.5) can produce, computed with the arithmetic of ``to_normalized_tensor``: the
    device side of the uint8 feed path (rmcl_image_u8_to_patches) looks pixels up here, so its floats are bit-identical to the
    float pipeline's."""
    return torch.arange(256, dtype=torch.uint8).to(torch.float32).div_(255.0).sub_(0.5).div_(0.5)


def pixelbert_uint8_transform(size=800):
    """The pixelbert transform WITHOUT the float conversion: callable(PIL image) -> uint8 tensor [h, w, 3] (HWC, as decoded).
    Normalisation, zero padding and the patch cut happen on the device in one kernel (Engine.bind_batch on a Uint8Batch): a
    quarter of the bytes through the loader's pipes and over PCIe, no float image in host memory."""
    resize = MinMaxResize(shorter=size, longer=int((1333 / 800) * size))
    return lambda img: torch.from_numpy(np.array(resize(img).convert("RGB"), dtype=np.uint8))


def decode_uint8_transform(size=800):
    """Decode only: callable(PIL image) -> ``RawView`` (uint8 [h, w, 3] at the ORIGINAL size + the MinMaxResize parameters of the pixelbert
    transform).  The bicubic resize - after the JPEG decode the most expensive thing a loader worker does - moves to the device
    (rmcl_image_resize_u8, PIL's own integer arithmetic: the bytes the worker would have produced)."""
    from ..datasets.base_dataset import RawView
    longer = int((1333 / 800) * size)
    return lambda img: RawView(torch.from_numpy(np.array(img.convert("RGB"), dtype=np.uint8)), size, longer)


# (the RandAugment variant is a training-time augmentation, out of scope)
_transforms = {"pixelbert": pixelbert_transform, "pixelbert_uint8": pixelbert_uint8_transform, "decode_uint8": decode_uint8_transform}


def keys_to_transforms(keys: list, size=224):
    return [_transforms[key](size=size) for key in keys]
