from .core_evaluator import CoreEvaluator, ModelInfoCalculator

__all__ = ["CoreEvaluator", "ModelInfoCalculator"]
