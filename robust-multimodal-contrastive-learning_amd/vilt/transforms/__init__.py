from .utils import MinMaxResize, min_max_resize_size, pixelbert_transform, keys_to_transforms  # noqa: F401
