from .base_dataset import collate  # noqa: F401
