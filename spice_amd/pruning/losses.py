"""KD-Lite losses with the reference's module interface (pruning/losses.py:15-162), computed by
the fused HIP kernel: one pass over the logits yields ce, kd and total (and, when the student
needs a gradient, d(total)/d(student) in the same pass)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops

_FIVE_CLASS_WEIGHTS = (0.5, 1.0, 1.1, 1.5, 2.0)  # pruning/losses.py:66


class _KDLiteFunction(torch.autograd.Function):
    """(student, teacher, labels) -> (total, ce, kd).  The forward launch already writes the
    gradient for upstream = (1, 0, 0); backward rescales it on the device, or recomputes with the
    actual upstream gradients when ce / kd are differentiated on their own."""

    @staticmethod
    def forward(ctx, student, teacher, labels, opts, want):
        # `want` (will a backward through the student come?) is decided by the caller, _apply below:
        # autograd runs forward() with grad mode switched off, and needs_input_grad ignores torch.no_grad()
        ctx.set_materialize_grads(False)
        out3, grad, _ = ops.kdlite_forward(student.detach(), teacher.detach(), labels, want_grad=want, **opts)
        # (a recomputation in backward takes the label pass again: the prepared normalisers belong to this launch)
        ctx.opts = {k: v for k, v in opts.items() if k not in ("workspace", "normalisers_ready")}
        ctx.saved_grad = grad
        ctx.save_for_backward(student, teacher, labels)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def forward_nograd(student, teacher, labels, opts):
        out3, _, _ = ops.kdlite_forward(student, teacher, labels, want_grad=False, **opts)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_total, g_ce, g_kd):
        student, teacher, labels = ctx.saved_tensors
        if g_ce is None and g_kd is None:
            if g_total is None:
                return None, None, None, None, None
            grad = ctx.saved_grad
            if grad is None:  # a second backward through the same graph: the first one consumed the buffer
                _, grad, _ = ops.kdlite_forward(student.detach(), teacher.detach(), labels,
                                                want_grad=True, **ctx.opts)
            ops.scale_inplace(grad, g_total.reshape(1).float().contiguous())
            ctx.saved_grad = None
            return grad, None, None, None, None
        zero = torch.zeros(1, device=student.device)
        up = torch.cat([(g if g is not None else zero).reshape(1).float() for g in (g_total, g_ce, g_kd)])
        _, grad, _ = ops.kdlite_forward(student.detach(), teacher.detach(), labels, want_grad=True,
                                        upstream=up.contiguous(), **ctx.opts)
        return grad, None, None, None, None


def _apply(student, teacher, labels, opts):
    want = torch.is_grad_enabled() and student.requires_grad
    return _KDLiteFunction.apply(student, teacher, labels, opts, want)


def _prepare(logits: torch.Tensor) -> torch.Tensor:
    if logits.dtype != torch.float32:
        logits = logits.float()
    return logits if logits.is_contiguous() else logits.contiguous()


class KDLiteLoss(nn.Module):
    """alpha * CE + (1 - alpha) * KD, optionally boundary- and confidence-weighted."""

    def __init__(self, temperature: float = 4.0, alpha: float = 0.5, ignore_index: int = 255,
                 boundary_weight: float = 1.0, confidence_weight: bool = False, num_classes: int = 5):
        super().__init__()
        self.temperature = temperature
        self.alpha = alpha
        self.ignore_index = ignore_index
        self.boundary_weight = boundary_weight
        self.confidence_weight = confidence_weight
        self.num_classes = num_classes
        self.ce_loss = SegmentationLoss(num_classes, ignore_index)
        self.kd_loss = PixelKDLoss(temperature)
        self._side = self._ws = self._prepared = None

    def prepare_labels(self, labels: torch.Tensor) -> bool:
        """Optional, before the networks' forwards of a training step: what the loss divides by (the pixels that are
        not ignored, the class weights of the labelled ones) depends on the labels alone, so the loss kernel's label
        pass is launched now, on a stream of its own, and the next forward() on the same labels tensor finds the
        result ready instead of launching the pass in front of its main kernel.  False (nothing done) for host
        tensors, other dtypes and the boundary / confidence options, whose pass reads the teacher's logits."""
        self._prepared = None
        if (not labels.is_cuda or labels.dtype != torch.int64 or not labels.is_contiguous() or labels.dim() != 3
                or self.boundary_weight > 1.0 or self.confidence_weight):
            return False
        dev = labels.device
        if self._side is None or self._ws.device != dev:
            self._side, self._ws = torch.cuda.Stream(device=dev), ops.kdlite_workspace(dev)
        # behind what produced the labels, and behind the last loss launch that read this workspace
        self._side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._side):
            done = ops.kdlite_label_pass(labels, self.num_classes, self._ws, ignore_index=self.ignore_index,
                                         class_weight=self.ce_loss.class_weights)
            ready = torch.cuda.Event()
            ready.record()
        if done:
            self._prepared = (labels, labels._version, ready)
        return done

    def forward(self, student_logits: torch.Tensor, teacher_logits: torch.Tensor,
                labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        opts = dict(temperature=self.temperature, alpha=self.alpha, ignore_index=self.ignore_index,
                    boundary_weight=self.boundary_weight, confidence_weight=self.confidence_weight,
                    class_weight=self.ce_loss.class_weights)
        prepared, self._prepared = self._prepared, None
        if (prepared is not None and prepared[0] is labels and prepared[1] == labels._version
                and student_logits.shape[1] == self.num_classes and torch.is_grad_enabled() and student_logits.requires_grad):
            torch.cuda.current_stream(labels.device).wait_event(prepared[2])
            opts.update(workspace=self._ws, normalisers_ready=True)
        total, ce, kd = _apply(_prepare(student_logits), _prepare(teacher_logits), labels.long().contiguous(), opts)
        return {"total": total, "ce": ce, "kd": kd}


class PixelKDLoss(nn.Module):
    """T^2 * masked mean of KL(softmax(t/T) || softmax(s/T)) (pruning/losses.py:22-55)."""

    def __init__(self, temperature: float = 4.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, student_logits: torch.Tensor, teacher_logits: torch.Tensor,
                mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        n, _, h, w = student_logits.shape
        if mask is None:
            labels = torch.zeros(n, h, w, dtype=torch.int64, device=student_logits.device)
        elif mask.dtype == torch.bool:
            labels = torch.where(mask, 0, 255).to(torch.int64)
        else:
            raise NotImplementedError(
                "PixelKDLoss with a float mask is only available through KDLiteLoss "
                "(boundary_weight / confidence_weight)")
        opts = dict(temperature=self.temperature, alpha=0.0, ignore_index=255)
        _, _, kd = _apply(_prepare(student_logits), _prepare(teacher_logits), labels.contiguous(), opts)
        return kd


class SegmentationLoss(nn.Module):
    """Cross-entropy with ignore_index and the 5-class weights (pruning/losses.py:58-84)."""

    def __init__(self, num_classes: int = 5, ignore_index: int = 255):
        super().__init__()
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        if num_classes == 5:
            self.register_buffer("class_weights", torch.tensor(_FIVE_CLASS_WEIGHTS, dtype=torch.float))
        else:
            self.class_weights = None

    def forward(self, predictions: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        opts = dict(temperature=1.0, alpha=1.0, ignore_index=self.ignore_index,
                    class_weight=self.class_weights)
        p = _prepare(predictions)
        # alpha = 1: total == ce, and its gradient is the one the fused pass already wrote
        total, _, _ = _apply(p, p.detach(), targets.long().contiguous(), opts)
        return total
