from .base_dataset import collate, collate_uint8, Uint8Batch, BaseDataset, select_from_sizes, write_arrow_table  # noqa: F401
