from .cache_manager import CacheManager
from .config_manager import ConfigManager
from .core_evaluator import CoreEvaluator, ModelInfoCalculator
from .result_processor import ResultFormatter, ResultProcessor, ResultSaver

__all__ = ["CacheManager", "ConfigManager", "CoreEvaluator", "ModelInfoCalculator", "ResultFormatter",
           "ResultProcessor", "ResultSaver"]
