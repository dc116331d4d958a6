"""Segmentation metrics of the pruned-model evaluator (pruning/eval/metrics.py:11-144).

The reference keeps five torchmetrics objects that each re-scan the flattened predictions; here
one int64 confusion matrix is accumulated on the device by the HIP histogram kernel and every
reported number derives from it (torchmetrics 1.8.1 semantics: rows = target, macro averages leave
out classes absent from both prediction and target, 0/0 = 0)."""
from __future__ import annotations

from typing import Any, Dict

import numpy as np
import torch

from ... import ops


class SegmentationMetrics:
    def __init__(self, num_classes: int = 5, device: str = "cpu"):
        self.num_classes = num_classes
        self.device = device
        self.conf = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)
        self._bad_base = self._bad_count()

    def _bad_count(self) -> int:
        # the counter is shared by every user of this device and stream and only ever grows:
        # this object reports what was added since its last reset
        return ops.invalid_prediction_count(self.conf.device, self.num_classes) if self.conf.is_cuda else 0

    def reset(self):
        self.conf.zero_()
        self._bad_base = self._bad_count()

    def update(self, predictions: torch.Tensor, targets: torch.Tensor):
        """predictions, targets: class indices [B,H,W]; targets outside [0,C) are skipped."""
        ops.confusion_update(self.conf, predictions.reshape(-1).long(), targets.reshape(-1).long(),
                             self.num_classes)

    def update_from_logits(self, logits: torch.Tensor, targets: torch.Tensor):
        ops.argmax_confusion_update(self.conf, logits.contiguous(), targets.long().contiguous(),
                                    self.num_classes)

    def update_from_model(self, model, images: torch.Tensor, targets: torch.Tensor):
        """Forward + update_from_logits, inside the caller's autocast region; models that end in a bilinear
        up-sampling are stopped at their head's convolution and up-sampled inside the histogram launch."""
        from ...baseline.eval.metrics import update_confusion_matrix_from_model
        update_confusion_matrix_from_model(self.conf, model, images, targets.long(), self.num_classes)

    def compute_metrics(self) -> Dict[str, Any]:
        bad = self._bad_count() - self._bad_base
        if bad:
            raise RuntimeError(f"{bad} predictions were outside [0, {self.num_classes})")
        m = self.conf.cpu().numpy().astype(np.float64)
        hit, rows, cols = np.diag(m), m.sum(1), m.sum(0)
        union = rows + cols - hit
        iou = np.divide(hit, union, out=np.zeros_like(hit), where=union > 0)
        acc = np.divide(hit, rows, out=np.zeros_like(hit), where=rows > 0)
        seen = (rows + cols) > 0
        return {"miou": float(iou[seen].mean()) if seen.any() else 0.0,
                "mean_acc": float(acc[seen].mean()) if seen.any() else 0.0,
                "per_class_iou": iou.astype(np.float32).tolist(),
                "per_class_acc": acc.astype(np.float32).tolist(),
                "confusion_matrix": self.conf.cpu().numpy()}

    def get_class_names(self) -> list:
        return [f"Class_{i}" for i in range(self.num_classes)]


def compute_metrics(confusion_matrix) -> Dict[str, Any]:
    """Per-class IoU / accuracy with the +1e-10 guard, means over ALL classes
    (pruning/eval/metrics.py:106-144)."""
    if not isinstance(confusion_matrix, torch.Tensor):
        confusion_matrix = torch.tensor(confusion_matrix)
    ious, accs = [], []
    for k in range(confusion_matrix.shape[0]):
        inter = confusion_matrix[k, k].float()
        gt = confusion_matrix[k, :].sum().float()
        pr = confusion_matrix[:, k].sum().float()
        ious.append((inter / (gt + pr - inter + 1e-10)).item())
        accs.append((inter / (gt + 1e-10)).item())
    return {"miou": np.mean(ious), "mean_acc": np.mean(accs), "per_class_iou": ious, "per_class_acc": accs}
