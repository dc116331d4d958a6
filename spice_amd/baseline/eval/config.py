"""Constants and the six model entries of the un-pruned evaluation (baseline/eval/config.py:5-65)."""

NUM_CLASSES = 5
IMG_SIZE = 576
BATCH_SIZE = 64
NUM_WORKERS = 16


def _entry(architecture: str, encoder: str, name: str) -> dict:
    return {"architecture": architecture, "encoder_name": encoder, "encoder_weights": None, "in_channels": 3,
            "classes": NUM_CLASSES, "activation": None, "checkpoint_path": f"checkpoints/fullsize/{name}/{name}.pt"}


MODEL_CONFIGS = {
    "DeepLabV3Plus_resnet": _entry("DeepLabV3Plus", "resnet50", "DeepLabV3Plus_resnet"),
    "DeepLabV3Plus_efficientnet": _entry("DeepLabV3Plus", "timm-efficientnet-b3", "DeepLabV3Plus_efficientnet"),
    "UNET_resnet": _entry("Unet", "resnet50", "UNET_resnet"),
    "UNET_efficientnet": _entry("Unet", "timm-efficientnet-b3", "UNET_efficientnet"),
    "FPN_resnet": _entry("FPN", "resnet50", "FPN_resnet"),
    "FPN_efficientnet": _entry("FPN", "timm-efficientnet-b3", "FPN_efficientnet"),
}
