"""Seeding, logging and model bookkeeping helpers (pruning/utils.py:16-99)."""
from __future__ import annotations

import hashlib
import logging
import random
from typing import Optional

import numpy as np
import torch


def set_random_seeds(seed: int = 42, deterministic: bool = True) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def count_model_parameters(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())


def calculate_model_size_mb(model: torch.nn.Module) -> float:
    n = sum(p.numel() * p.element_size() for p in model.parameters())
    n += sum(b.numel() * b.element_size() for b in model.buffers())
    return n / (1024 ** 2)


def model_checksum(model: torch.nn.Module, limit: Optional[int] = 8) -> str:
    h = hashlib.md5()
    for i, (name, p) in enumerate(model.named_parameters()):
        if limit is not None and i >= limit:
            break
        h.update(name.encode())
        h.update(p.detach().float().cpu().numpy().tobytes()[:4096])
    return h.hexdigest()[:8]


def setup_logging(level: int = logging.INFO) -> None:
    logging.basicConfig(level=level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
