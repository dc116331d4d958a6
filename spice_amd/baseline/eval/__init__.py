"""Mirror of the reference's baseline.eval exports used on the evaluation path."""
from .inference import evaluate_model, inference
from .metrics import compute_metrics_from_confusion, update_confusion_matrix

__all__ = ["inference", "evaluate_model", "update_confusion_matrix", "compute_metrics_from_confusion"]
