"""Confusion-matrix metrics of the un-pruned evaluation path (baseline/eval/metrics.py:8-36)."""
from __future__ import annotations

import os

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


#: evaluation batches of models whose head ends in nn.UpsamplingBilinear2d: up-sampling, arg-max and histogram in one
#: launch over the head convolution's small output (SPICE_EVAL_FUSED_HEAD=0: the model's own forward, then the fused
#: arg-max + histogram launch)
FUSED_UPSAMPLE = os.environ.get("SPICE_EVAL_FUSED_HEAD", "1") != "0"


def update_confusion_matrix_from_model(conf: torch.Tensor, model, images: torch.Tensor, target: torch.Tensor,
                                       num_classes: int) -> None:
    """conf[t, argmax model(images)] += 1 (baseline/eval/inference.py:45-48), called inside the caller's autocast
    region.  Where the model can stop at its head's convolution (SegmentationModel.forward_to_head) the bilinear
    up-sampling happens inside the histogram launch and the full-size logits are never written: the same matrix, the
    interpolated values being the up-sampling kernel's bit for bit."""
    parts = None
    if FUSED_UPSAMPLE and hasattr(model, "forward_to_head") and target.dim() == 3:
        parts = model.forward_to_head(images)
    if parts is None:
        outputs = model(images)
        if isinstance(outputs, dict):
            outputs = outputs.get("out", outputs.get("logits", outputs))
    else:
        low, up = parts
        if up is None:
            outputs = low
        else:
            sf = up.scale_factor
            whole = float(sf).is_integer() and tuple(target.shape[1:]) == (low.shape[2] * int(sf), low.shape[3] * int(sf))
            if (whole and low.shape[1] == num_classes and target.dtype == torch.int64
                    and ops.upsample_argmax_confusion(conf, low, target.contiguous(), align_corners=True,
                                                      scale_factor=float(sf))):
                return
            outputs = up(low)
    update_confusion_matrix_from_logits(conf, outputs, target, num_classes)


def compute_metrics_from_confusion(conf: torch.Tensor):
    """(per-class IoU, per-class accuracy, mean IoU over all classes), fp32 as the reference."""
    m = conf.float()
    hit = torch.diag(m)
    rows, cols = m.sum(dim=1), m.sum(dim=0)
    iou = hit / (rows + cols - hit + 1e-10)
    acc = hit / (rows + 1e-10)
    return iou.cpu().numpy(), acc.cpu().numpy(), float(iou.mean().item())
