"""Confusion-matrix metrics of the un-pruned evaluation path (baseline/eval/metrics.py:8-36)."""
from __future__ import annotations

import torch

from ... import ops


def update_confusion_matrix(conf: torch.Tensor, preds: torch.Tensor, target: torch.Tensor,
                            num_classes: int) -> None:
    """conf[t, p] += 1 over pixels with 0 <= t < num_classes, in place (HIP histogram kernel)."""
    ops.confusion_update(conf, preds.reshape(-1), target.reshape(-1), num_classes)


def update_confusion_matrix_from_logits(conf: torch.Tensor, logits: torch.Tensor, target: torch.Tensor,
                                        num_classes: int) -> None:
    """The arg-max of baseline/eval/inference.py:47 fused into the histogram: no prediction map."""
    ops.argmax_confusion_update(conf, logits.contiguous(), target.contiguous(), num_classes)


def compute_metrics_from_confusion(conf: torch.Tensor):
    """(per-class IoU, per-class accuracy, mean IoU over all classes), fp32 as the reference."""
    m = conf.float()
    hit = torch.diag(m)
    rows, cols = m.sum(dim=1), m.sum(dim=0)
    iou = hit / (rows + cols - hit + 1e-10)
    acc = hit / (rows + 1e-10)
    return iou.cpu().numpy(), acc.cpu().numpy(), float(iou.mean().item())
