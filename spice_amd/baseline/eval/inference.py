"""Evaluation loop of the un-pruned path (baseline/eval/inference.py:16-78).

Forward under autocast as the reference; the arg-max over classes (inference.py:47) and the
confusion update (inference.py:48) are ONE kernel launch that reads the logits once and never
writes the [N,H,W] prediction map."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from ... import dist as sdist
from .metrics import compute_metrics_from_confusion, update_confusion_matrix_from_model


def _split_batch(batch, device):
    if isinstance(batch, (list, tuple)):
        return batch[0].to(device, non_blocking=True), batch[1].to(device, non_blocking=True).long()
    return batch["image"].to(device, non_blocking=True), batch["mask"].to(device, non_blocking=True).long()


@torch.no_grad()
def inference(model: nn.Module, test_loader: DataLoader, device: torch.device, num_classes: int = 5,
              shard: bool = True):
    """-> {"miou", "mean_acc", "per_class_iou", "per_class_acc"}.

    With torch.distributed initialised (one process per GPU) and `shard`, every rank iterates the
    same loader, evaluates the batches whose index is congruent to its rank, and the [C,C] int64
    matrices are summed with one all-reduce: integer counts, so every rank ends with exactly the
    single-process matrix whatever the order."""
    device = torch.device(device)
    if hasattr(model, "eval"):
        model.eval()
    model = model.to(device)
    conf = torch.zeros(num_classes, num_classes, dtype=torch.long, device=device)
    rank, world = sdist.world() if shard else (0, 1)
    with torch.amp.autocast(device.type, enabled=(device.type == "cuda")):
        for k, batch in enumerate(test_loader):
            if k % world != rank:
                continue
            images, masks = _split_batch(batch, device)
            # forward + arg-max + histogram (for models that end in a bilinear up-sampling: the up-sampling inside the
            # histogram launch)
            update_confusion_matrix_from_model(conf, model, images, masks, num_classes)
    if world > 1:
        sdist.all_reduce_sum_(conf)
    iou, acc, miou = compute_metrics_from_confusion(conf)
    return {"miou": miou, "mean_acc": float(np.mean(acc)), "per_class_iou": iou.tolist(),
            "per_class_acc": acc.tolist()}


@torch.no_grad()
def evaluate_model(model: nn.Module, model_path: str, data_loader: DataLoader, device: torch.device,
                   num_classes: int = 5):
    """Load a state dict (non-strict) and evaluate; None when the file does not exist."""
    if not Path(model_path).exists():
        return None
    state = torch.load(model_path, map_location=device, weights_only=True)
    model.load_state_dict(state, strict=False)
    return inference(model, data_loader, device, num_classes)
