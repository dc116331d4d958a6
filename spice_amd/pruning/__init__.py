"""Drop-in mirror of the reference's `pruning` package exports (pruning/__init__.py:5-31)."""
from .config import PruningConfig
from .model_configs import AVAILABLE_STRATEGIES, MODEL_CONFIGS
from .models import ModelFactory
from .pruner import StructuredPruner, prune_model
from .sensitivity import GroupSyncTaylorImportance, TaylorWeightImportance, create_taylor_importance
from .strategies import PruningStrategyFactory
from .trainer import PruningTrainer
from .utils import set_random_seeds

__all__ = ["PruningConfig", "MODEL_CONFIGS", "AVAILABLE_STRATEGIES", "ModelFactory",
           "PruningStrategyFactory", "PruningTrainer", "StructuredPruner", "prune_model",
           "TaylorWeightImportance", "GroupSyncTaylorImportance", "create_taylor_importance",
           "set_random_seeds"]
