"""CPU oracle for the KD-Lite loss (test infrastructure; never imported by spice_amd/).

`kdlite_loss` replays the reference's torch op sequence on CPU tensors (so it rounds like the
reference does); `kdlite_loss_f64` is an independent numpy float64 evaluation of the same
formulas, used to judge the fp32 paths against the exact value.
Pinned by tests/golden/kdlite_*.npz, generated from the reference's pruning/losses.py.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

FIVE_CLASS_WEIGHTS = (0.5, 1.0, 1.1, 1.5, 2.0)  # pruning/losses.py:66


def pixel_kd(student: torch.Tensor, teacher: torch.Tensor, mask: Optional[torch.Tensor],
             temperature: float) -> torch.Tensor:
    """PixelKDLoss.forward, pruning/losses.py:22-55."""
    log_ps = F.log_softmax(student / temperature, dim=1)
    pt = F.softmax(teacher / temperature, dim=1)
    per_pixel = F.kl_div(log_ps, pt, reduction="none").sum(dim=1)
    if mask is None:
        value = per_pixel.mean()
    else:
        per_pixel = per_pixel * mask.float()
        denom = mask.sum()
        value = per_pixel.sum() / denom if denom > 0 else per_pixel.sum() * 0
    return value * (temperature ** 2)


def segmentation_ce(student: torch.Tensor, labels: torch.Tensor, num_classes: int,
                    ignore_index: int) -> torch.Tensor:
    """SegmentationLoss.forward, pruning/losses.py:58-84 (class weights iff num_classes == 5)."""
    weight = torch.tensor(FIVE_CLASS_WEIGHTS, dtype=torch.float) if num_classes == 5 else None
    return F.cross_entropy(student, labels, weight=weight, ignore_index=ignore_index)


def boundary_mask(labels: torch.Tensor) -> torch.Tensor:
    """KDLiteLoss._get_boundary_mask, pruning/losses.py:109-119: 3x3 dilate != erode."""
    x = labels.float().unsqueeze(1)
    hi = F.max_pool2d(x, 3, stride=1, padding=1)
    lo = -F.max_pool2d(-x, 3, stride=1, padding=1)
    return (hi != lo).squeeze(1).float()


def teacher_confidence(teacher: torch.Tensor) -> torch.Tensor:
    """KDLiteLoss._get_confidence_weights, pruning/losses.py:121-125."""
    return F.softmax(teacher, dim=1).max(dim=1).values


def kdlite_loss(student: torch.Tensor, teacher: torch.Tensor, labels: torch.Tensor, *,
                temperature: float = 4.0, alpha: float = 0.5, ignore_index: int = 255,
                boundary_weight: float = 1.0, confidence_weight: bool = False,
                num_classes: int = 5) -> Dict[str, torch.Tensor]:
    """KDLiteLoss.forward, pruning/losses.py:127-162."""
    ce = segmentation_ce(student, labels, num_classes, ignore_index)
    valid = labels != ignore_index
    kd = pixel_kd(student, teacher, valid, temperature)
    if boundary_weight > 1.0:
        edge = boundary_mask(labels) * valid.float()
        kd = kd + (boundary_weight - 1.0) * pixel_kd(student, teacher, edge, temperature)
    if confidence_weight:
        cw = teacher_confidence(teacher) * valid.float()
        kd = 0.5 * kd + 0.5 * pixel_kd(student, teacher, cw, temperature)
    return {"total": alpha * ce + (1 - alpha) * kd, "ce": ce, "kd": kd}


def kdlite_loss_and_grad(student: torch.Tensor, teacher: torch.Tensor, labels: torch.Tensor,
                         **kw) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """Loss dict plus d(total)/d(student) by autograd (what trainer.py:179-182 obtains)."""
    s = student.detach().clone().requires_grad_(True)
    out = kdlite_loss(s, teacher.detach(), labels, **kw)
    (grad,) = torch.autograd.grad(out["total"], s, allow_unused=True)
    if grad is None:
        grad = torch.zeros_like(s)
    return {k: v.detach() for k, v in out.items()}, grad


def _log_softmax64(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def kdlite_loss_f64(student: np.ndarray, teacher: np.ndarray, labels: np.ndarray, *,
                    temperature: float = 4.0, alpha: float = 0.5, ignore_index: int = 255,
                    boundary_weight: float = 1.0, confidence_weight: bool = False,
                    num_classes: int = 5, want_grad: bool = False):
    """float64 evaluation of pruning/losses.py:127-162 (and its student gradient)."""
    s = student.astype(np.float64)
    t = teacher.astype(np.float64)
    n, c, h, w = s.shape
    T = float(temperature)
    lps = _log_softmax64(s / T)
    lpt = _log_softmax64(t / T)
    pt = np.exp(lpt)
    kl = (pt * (lpt - lps)).sum(axis=1)  # [n,h,w]
    valid = labels != ignore_index
    vf = valid.astype(np.float64)

    def masked(mask):
        d = mask.sum()
        return ((kl * mask).sum() / d if d > 0 else 0.0) * T * T, d

    kd, n_valid = masked(vf)
    pix_w = np.where(n_valid > 0, vf / max(n_valid, 1e-300), 0.0)
    if boundary_weight > 1.0:
        lab = labels.astype(np.float64)
        pad_hi = np.pad(lab, ((0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
        pad_lo = np.pad(lab, ((0, 0), (1, 1), (1, 1)), constant_values=np.inf)
        hi = np.full_like(lab, -np.inf)
        lo = np.full_like(lab, np.inf)
        for dy in range(3):
            for dx in range(3):
                hi = np.maximum(hi, pad_hi[:, dy:dy + h, dx:dx + w])
                lo = np.minimum(lo, pad_lo[:, dy:dy + h, dx:dx + w])
        edge = (hi != lo).astype(np.float64) * vf
        kb, n_b = masked(edge)
        kd = kd + (boundary_weight - 1.0) * kb
        if n_b > 0:
            pix_w = pix_w + (boundary_weight - 1.0) * edge / n_b
    if confidence_weight:
        conf = np.exp(_log_softmax64(t)).max(axis=1) * vf
        kc, sum_c = masked(conf)
        kd = 0.5 * kd + 0.5 * kc
        pix_w = 0.5 * pix_w + (0.5 * conf / sum_c if sum_c > 0 else np.zeros_like(conf))

    lp1 = _log_softmax64(s)
    in_range = valid & (labels >= 0) & (labels < c)
    safe = np.where(in_range, labels, 0)
    cls_w = np.asarray(FIVE_CLASS_WEIGHTS if num_classes == 5 else [1.0] * c, dtype=np.float64)
    wy = np.where(in_range, cls_w[safe], 0.0)
    nll = -np.take_along_axis(lp1, safe[:, None], axis=1)[:, 0]
    w_sum = wy.sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        ce = (wy * nll).sum() / w_sum
    out = {"total": alpha * ce + (1 - alpha) * kd, "ce": ce, "kd": kd}
    if not want_grad:
        return out
    onehot = np.zeros_like(s)
    np.put_along_axis(onehot, safe[:, None], 1.0, axis=1)
    g_ce = (np.exp(lp1) - onehot) * (wy / w_sum if w_sum > 0 else np.zeros_like(wy))[:, None]
    g_kd = (np.exp(lps) - pt) * (T * pix_w)[:, None]  # T^2 * (1/T)
    return out, alpha * g_ce + (1 - alpha) * g_kd
