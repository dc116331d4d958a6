"""Data handling for pruning runs (pruning/dataset.py:13-55).

The reference reads a Pascal-VOC-style insect dataset that does not ship with it
(data/data.py:97-244; data/ is git-ignored).  When `data_root` holds no `train/` directory the
handler serves a deterministic synthetic dataset of the configured geometry instead (the
reference's own precedent: quantization/quant_utils.py:13-19).  With several ranks the training
split is sharded by a DistributedSampler; validation and test splits are sharded by index stride.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Subset
from torch.utils.data.distributed import DistributedSampler

from .. import dist as sdist
from .. import synth


class SyntheticSegmentationDataset(Dataset):
    """Images ~ N(0,1) and blocky label maps with a share of ignore pixels."""

    def __init__(self, length: int, num_classes: int, input_size: Tuple[int, int], seed: int,
                 ignore_share: float = 0.0):
        self.length, self.num_classes, self.size = length, num_classes, tuple(input_size)
        self.seed, self.ignore_share = seed, ignore_share

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int):
        h, w = self.size
        img = synth.normal(self.seed + 17 * i, 7, 3 * h * w).reshape(3, h, w)
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        lab = (((yy // 32) * 7 + (xx // 32) * 3 + i) % self.num_classes).astype(np.int64)
        if self.ignore_share > 0:
            lab[synth.uniform(self.seed + 17 * i, 8, h * w).reshape(h, w) < self.ignore_share] = 255
        return torch.from_numpy(img), torch.from_numpy(lab)


class PruningDataHandler:
    def __init__(self, data_root: str = "data", batch_size: int = 4, num_workers: int = 2,
                 seed: int = 42, num_classes: int = 5, input_size: Tuple[int, int] = (576, 576),
                 synthetic_length: int = 32):
        self.data_root, self.batch_size, self.num_workers, self.seed = data_root, batch_size, num_workers, seed
        self.num_classes, self.input_size, self.synthetic_length = num_classes, input_size, synthetic_length

    def _dataset(self, split: str, ignore_share: float = 0.0) -> Dataset:
        if os.path.isdir(os.path.join(self.data_root, split)):
            raise NotImplementedError(
                "reading the XML-annotated image dataset is outside this package; "
                "point data_root at a directory without train/ to use synthetic data")
        offset = {"train": 0, "valid": 1000, "test": 2000}[split]
        return SyntheticSegmentationDataset(self.synthetic_length, self.num_classes, self.input_size,
                                            self.seed + offset, ignore_share)

    def _loader(self, split: str, shuffle: bool, ignore_share: float = 0.0, distribute: bool = True) -> DataLoader:
        data = self._dataset(split, ignore_share)
        rank, world = sdist.world() if distribute else (0, 1)
        g = torch.Generator()
        g.manual_seed(self.seed)
        if world > 1 and shuffle:
            sampler = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=self.seed)
            return DataLoader(data, batch_size=self.batch_size, sampler=sampler, num_workers=0, drop_last=False)
        if world > 1:
            data = Subset(data, list(range(rank, len(data), world)))
        return DataLoader(data, batch_size=self.batch_size, shuffle=shuffle, num_workers=0, generator=g,
                          drop_last=False)

    def get_dataloaders(self) -> Tuple[DataLoader, DataLoader, DataLoader]:
        """(train, valid, test), as data/data.py:170-244 returns them."""
        return (self._loader("train", True, 0.05), self._loader("valid", False, 0.05),
                self._loader("test", False, 0.05))

    def get_train_val_loaders(self) -> Tuple[DataLoader, DataLoader]:
        train, val, _ = self.get_dataloaders()
        return train, val

    def get_calibration_dataloader(self) -> DataLoader:
        # the Taylor pass uses cross-entropy without ignore_index: labels must be in [0, C).
        # Every rank iterates the same seeded order; the Taylor pass itself takes batches round-robin.
        return self._loader("train", True, 0.0, distribute=False)
