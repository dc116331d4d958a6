"""CPU oracle for channel importance (test infrastructure; never imported by spice_amd/).

Taylor part: restates pruning/sensitivity_utils.py:15-73, pinned by tests/golden/taylor_*.npz.
Magnitude part: restates torch-pruning 1.6.0 `MagnitudeImportance` (group_reduction="mean",
normalizer="mean"), which the reference only constructs (pruning/pruner.py:268-270,
pruning/pruning_core.py:143-145, pruning/strategies.py:53); torch-pruning is absent from
/root/reference and from this image: parity unpinned for that part.
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def taylor_rows(weight: torch.Tensor, grad: torch.Tensor) -> torch.Tensor:
    """(grad.abs() * weight.abs()).sum(dim=(1,2,3)), pruning/sensitivity_utils.py:63-65."""
    return (grad.detach().abs() * weight.detach().abs()).sum(dim=tuple(range(1, weight.dim())))


def taylor_rows_f64(weight: np.ndarray, grad: np.ndarray) -> np.ndarray:
    w = np.abs(weight.astype(np.float64)).reshape(weight.shape[0], -1)
    g = np.abs(grad.astype(np.float64)).reshape(grad.shape[0], -1)
    return (w * g).sum(axis=1)


def compute_taylor_scores(model: nn.Module, batches: Iterable, device: str = "cpu",
                          max_batches: int = 100) -> Dict[nn.Module, torch.Tensor]:
    """pruning/sensitivity_utils.py:15-73: eval-mode forward/backward per calibration batch,
    per-Conv2d accumulation of the Taylor rows, division by the batches seen."""
    was_training = model.training
    model.eval()
    convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    totals = {m: torch.zeros(m.out_channels, device=device) for m in convs}
    seen = 0
    for i, batch in enumerate(batches):
        if i >= max_batches:
            break
        if isinstance(batch, (list, tuple)):
            images = batch[0].to(device)
            targets = batch[1].to(device) if len(batch) > 1 else None
        else:
            images, targets = batch.to(device), None
        model.zero_grad(set_to_none=True)
        out = model(images)
        if isinstance(out, dict):
            out = out.get("out", list(out.values())[0])
        if targets is None:
            loss = out.abs().mean()
        elif targets.dim() == 3:
            loss = F.cross_entropy(out, targets.long())
        else:
            loss = F.cross_entropy(out, targets)
        loss.backward()
        for m in convs:
            if m.weight.grad is not None:
                totals[m] += taylor_rows(m.weight, m.weight.grad)
        seen += 1
    for m in convs:
        totals[m] = totals[m] / max(seen, 1)
    model.train(was_training)
    return totals


def magnitude_rows(weight: torch.Tensor, p: int) -> torch.Tensor:
    """Out-channel term: weight.flatten(1).abs().pow(p).sum(1) (no p-th root)."""
    return weight.detach().flatten(1).abs().pow(p).sum(1)


def magnitude_cols(weight: torch.Tensor, p: int) -> torch.Tensor:
    """In-channel term of a consumer: weight.transpose(0,1).flatten(1).abs().pow(p).sum(1)."""
    return weight.detach().transpose(0, 1).flatten(1).abs().pow(p).sum(1)


def magnitude_rows_f64(weight: np.ndarray, p: int) -> np.ndarray:
    return (np.abs(weight.astype(np.float64)).reshape(weight.shape[0], -1) ** p).sum(axis=1)


def magnitude_cols_f64(weight: np.ndarray, p: int) -> np.ndarray:
    w = np.abs(weight.astype(np.float64)) ** p
    return np.swapaxes(w, 0, 1).reshape(weight.shape[1], -1).sum(axis=1)


def group_magnitude(member_terms: Sequence[torch.Tensor], normalizer: str = "mean") -> torch.Tensor:
    """Group reduction "mean" over the members' per-channel terms, then the normalizer:
    "mean" (MagnitudeImportance's default) or "lamp" (LAMPImportance: every score divided by the
    sum of all scores not smaller than it, i.e. a cumulative sum over the descending order)."""
    stacked = torch.stack([t.float() for t in member_terms], dim=0)
    imp = stacked.sum(dim=0) / len(member_terms)
    if normalizer == "mean":
        return imp / imp.mean()
    if normalizer == "lamp":
        order = torch.argsort(imp, descending=True, stable=True)
        srt = imp[order]
        scaled = srt / torch.cumsum(srt, dim=0)
        out = torch.empty_like(scaled)
        out[order] = scaled
        return out
    raise ValueError(normalizer)


def group_taylor(member_terms: Sequence[torch.Tensor]) -> torch.Tensor:
    """tp.importance.TaylorImportance with its defaults on the producers of a group: mean over
    members of sum |w * dL/dw| per out-channel, divided by the mean over channels."""
    stacked = torch.stack([t.float() for t in member_terms], dim=0)
    imp = stacked.sum(dim=0) / len(member_terms)
    return imp / imp.mean()
