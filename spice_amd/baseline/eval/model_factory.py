"""Model instances for the un-pruned evaluation (baseline/eval/model_factory.py:10-52); the three
architectures come from this package's own definitions (same layer names as smp's)."""
from ... import nets
from .config import MODEL_CONFIGS


def create_model_from_config(config: dict):
    architecture = config["architecture"]
    if architecture not in ("DeepLabV3Plus", "Unet", "FPN"):
        raise ValueError(f"Unknown architecture: {architecture}")
    if config.get("activation") is not None or config.get("encoder_weights") is not None:
        raise ValueError("activation and encoder_weights must be None (neither is built here)")
    return nets.create_model(architecture, config["encoder_name"], classes=config["classes"],
                             in_channels=config.get("in_channels", 3))


def get_model_by_name(model_name: str):
    """(model instance, checkpoint path relative to the repository root)."""
    if model_name not in MODEL_CONFIGS:
        raise ValueError(f"Unknown model: {model_name}")
    config = MODEL_CONFIGS[model_name]
    return create_model_from_config(config), config["checkpoint_path"]
