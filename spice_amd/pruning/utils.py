"""Seeding, device choice, logging and model bookkeeping helpers (pruning/utils.py:16-99)."""
from __future__ import annotations

import hashlib
import logging
import random
from typing import Optional

import numpy as np
import torch


def set_random_seeds(seed: int = 42) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def get_device(device_str: Optional[str] = None) -> torch.device:
    """None / "auto" pick the GPU when one is visible (pruning/utils.py:33-46)."""
    if device_str in (None, "auto"):
        device_str = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.device(device_str)


def generate_config_hash(config_str: str) -> str:
    """First 8 hex digits of the MD5 of the string (pruning/utils.py:49-59)."""
    return hashlib.md5(config_str.encode()).hexdigest()[:8]


def count_model_parameters(model: torch.nn.Module, only_trainable: bool = False) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad or not only_trainable)


def calculate_model_size_mb(model: torch.nn.Module) -> float:
    """Parameters at 4 bytes each, in MiB; buffers are not counted (pruning/utils.py:78-87)."""
    return count_model_parameters(model) * 4 / (1024 * 1024)


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
