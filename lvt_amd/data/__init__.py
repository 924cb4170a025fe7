from .dataset_mapper import DatasetMapper, prepare_slices

__all__ = ["DatasetMapper", "prepare_slices"]
