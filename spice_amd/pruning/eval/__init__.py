from .metrics import SegmentationMetrics, compute_metrics  # noqa: F401
