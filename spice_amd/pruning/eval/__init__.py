"""Mirror of the reference's pruning.eval exports that belong to the evaluation path
(pruning/eval/__init__.py:5-18; the plotting classes are not part of this package)."""
from .config import EvaluationConfig  # noqa: F401
from .evaluator import PrunedModelEvaluator  # noqa: F401
from .loader import PrunedModelLoader  # noqa: F401
from .metrics import SegmentationMetrics, compute_metrics  # noqa: F401
