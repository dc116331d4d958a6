from .metrics import compute_metrics_from_confusion, update_confusion_matrix  # noqa: F401
