from .utils import MinMaxResize, min_max_resize_size, pixelbert_transform, pixelbert_uint8_transform, decode_uint8_transform, normalize_lut, keys_to_transforms  # noqa: F401
