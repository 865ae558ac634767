"""Image side of the input pipeline: ``MinMaxResize`` (vilt/transforms/utils.py:5-26) and the
``pixelbert`` transform (vilt/transforms/pixelbert.py:9-17: MinMaxResize -> ToTensor -> Normalize(mean .5, std .5)), host
side, PIL + torch only (torchvision is not required).  Every side comes out a multiple of 32, which is what the on-device
ragged ``visual_embed`` path expects of a zero-padded batch."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image


def min_max_resize_size(w: int, h: int, shorter: int = 800, longer: int = 1333):
    """(new_w, new_h) of MinMaxResize for a (w, h) image - the arithmetic of transforms/utils.py:11-24."""
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
    return neww, newh


class MinMaxResize:
    def __init__(self, shorter=800, longer=1333):
        self.min = shorter
        self.max = longer

    def __call__(self, x: Image.Image) -> Image.Image:
        w, h = x.size
        neww, newh = min_max_resize_size(w, h, self.min, self.max)
        return x.resize((neww, newh), resample=Image.BICUBIC)


def to_normalized_tensor(img: Image.Image) -> torch.Tensor:
    """transforms.ToTensor() followed by inception_normalize (mean 0.5, std 0.5): uint8 HWC -> float32 CHW in [-1, 1]."""
    a = np.asarray(img.convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
    return t.sub_(0.5).div_(0.5)


def pixelbert_transform(size=800):
    longer = int((1333 / 800) * size)
    resize = MinMaxResize(shorter=size, longer=longer)
    return lambda img: to_normalized_tensor(resize(img))


_transforms = {"pixelbert": pixelbert_transform}


def keys_to_transforms(keys: list, size=224):
    """vilt/transforms/__init__.py:13-14 (the RandAugment variant is a training-time augmentation outside the hot path)."""
    return [_transforms[key](size=size) for key in keys]
