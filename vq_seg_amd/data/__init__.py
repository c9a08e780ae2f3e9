"""Data path (reference: data/dataset.py:15-62 `BaseDataset`), SURVEY 8(f) row 4."""
from .dataset import BaseDataset, write_synthetic_dataset  # noqa: F401
