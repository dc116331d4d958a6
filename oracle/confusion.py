"""CPU oracle for the mIoU confusion matrix and the metrics derived from it
(test infrastructure; never imported by spice_amd/).

Pinned by tests/golden/confusion_*.npz generated from the reference's
baseline/eval/metrics.py.  `macro_metrics_present_classes` restates torchmetrics 1.8.1
(absent from /root/reference and from this image): parity unpinned.
"""
from __future__ import annotations

from typing import Any, Dict, Tuple

import numpy as np
import torch


def update_confusion(conf: np.ndarray, preds: np.ndarray, target: np.ndarray, num_classes: int) -> None:
    """update_confusion_matrix, baseline/eval/metrics.py:8-19 (in place; rows = target)."""
    p = np.asarray(preds).reshape(-1)
    t = np.asarray(target).reshape(-1)
    keep = (t >= 0) & (t < num_classes)
    flat = t[keep].astype(np.int64) * num_classes + p[keep].astype(np.int64)
    if flat.size and flat.min() < 0:
        raise ValueError("negative histogram index (the reference's bincount rejects it)")
    hist = np.bincount(flat, minlength=num_classes * num_classes)
    conf += hist.reshape(num_classes, num_classes)  # raises if a prediction overflowed the table


def update_confusion_torch(conf: torch.Tensor, preds: torch.Tensor, target: torch.Tensor, num_classes: int) -> None:
    """The same function as the torch op sequence the reference runs (baseline/eval/metrics.py:12-19: two
    flattening views, the range mask, two compacting gathers, multiply-add, bincount, in-place add), on CPU
    tensors: what the CPU leg of the benchmark times beside the HIP kernel."""
    p = preds.reshape(-1)
    t = target.reshape(-1)
    valid = (t >= 0) & (t < num_classes)
    p = p[valid]
    t = t[valid]
    idx = t * num_classes + p
    binc = torch.bincount(idx, minlength=num_classes * num_classes)
    conf += binc.view(num_classes, num_classes)


def argmax_first(logits: np.ndarray) -> np.ndarray:
    """logits.argmax(dim=1) as used at baseline/eval/inference.py:47 (first maximum wins)."""
    return np.argmax(logits, axis=1).astype(np.int64)


def metrics_from_confusion(conf: torch.Tensor) -> Tuple[np.ndarray, np.ndarray, float]:
    """compute_metrics_from_confusion, baseline/eval/metrics.py:22-36 (fp32, all classes)."""
    m = conf.float()
    hit = torch.diag(m)
    rows = m.sum(dim=1)
    cols = m.sum(dim=0)
    iou = hit / (rows + cols - hit + 1e-10)
    acc = hit / (rows + 1e-10)
    return iou.numpy(), acc.numpy(), float(iou.mean().item())


def per_class_metrics(conf: torch.Tensor) -> Dict[str, Any]:
    """compute_metrics, pruning/eval/metrics.py:106-144 (per-class loop, numpy means)."""
    if not isinstance(conf, torch.Tensor):
        conf = torch.tensor(conf)
    ious, accs = [], []
    for k in range(conf.shape[0]):
        inter = conf[k, k].float()
        gt = conf[k, :].sum().float()
        pr = conf[:, k].sum().float()
        ious.append((inter / (gt + pr - inter + 1e-10)).item())
        accs.append((inter / (gt + 1e-10)).item())
    return {"miou": np.mean(ious), "mean_acc": np.mean(accs), "per_class_iou": ious,
            "per_class_acc": accs}


def macro_metrics_present_classes(conf: np.ndarray) -> Dict[str, Any]:
    """torchmetrics 1.8.1 multiclass JaccardIndex/Accuracy, average "macro"/"none", as called at
    pruning/eval/metrics.py:19-37,85-89: 0/0 -> 0 and classes absent from both prediction and
    target are left out of the macro mean."""
    m = conf.astype(np.float64)
    hit = np.diag(m)
    rows, cols = m.sum(1), m.sum(0)
    union = rows + cols - hit
    iou = np.divide(hit, union, out=np.zeros_like(hit), where=union > 0)
    acc = np.divide(hit, rows, out=np.zeros_like(hit), where=rows > 0)
    seen = (rows + cols) > 0
    return {"miou": float(iou[seen].mean()) if seen.any() else 0.0,
            "mean_acc": float(acc[seen].mean()) if seen.any() else 0.0,
            "per_class_iou": iou.tolist(), "per_class_acc": acc.tolist()}
