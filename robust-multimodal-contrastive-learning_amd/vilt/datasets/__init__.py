from .base_dataset import (collate, collate_uint8, collate_raw_uint8, Uint8Batch, RawUint8Batch, RawView, BaseDataset, select_from_sizes,  # noqa: F401
                           write_arrow_table)
