"""Model and strategy names accepted by the toolkit (pruning/model_configs.py:5-69)."""

_ARCH = {"DeepLabV3Plus": "DeepLabV3Plus", "UNET": "Unet", "FPN": "FPN"}
_ENC = {"resnet": "resnet50", "efficientnet": "timm-efficientnet-b3"}


def _entry(arch_key: str, enc_key: str) -> dict:
    name = f"{arch_key}_{enc_key}"
    return {
        "architecture": _ARCH[arch_key],
        "encoder_name": _ENC[enc_key],
        "encoder_weights": None,
        "in_channels": 3,
        "classes": 5,
        "activation": None,
        "checkpoint_path": f"checkpoints/fullsize/{name}/{name}.pt",
    }


MODEL_CONFIGS = {f"{a}_{e}": _entry(a, e) for a in ("DeepLabV3Plus", "UNET", "FPN")
                 for e in ("resnet", "efficientnet")}

AVAILABLE_STRATEGIES = ["magnitude", "random", "taylor", "lamp", "taylor_weight", "magnitude_taylor"]
