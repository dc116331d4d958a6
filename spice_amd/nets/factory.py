"""Model factory: (architecture, encoder_name) -> SegmentationModel."""
from __future__ import annotations

import torch.nn as nn

from .decoders import (DeepLabV3PlusDecoder, FPNDecoder, SegmentationHead, SegmentationModel,
                       UnetDecoder)
from .efficientnet import EfficientNetB3Encoder
from .resnet import ResNet50Encoder

ENCODERS = {"resnet50": ResNet50Encoder, "timm-efficientnet-b3": EfficientNetB3Encoder}
ARCHITECTURES = ("DeepLabV3Plus", "Unet", "FPN")


def create_model(architecture: str, encoder_name: str, classes: int, in_channels: int = 3) -> nn.Module:
    if encoder_name not in ENCODERS:
        raise ValueError(f"Unsupported encoder: {encoder_name}")
    enc_cls = ENCODERS[encoder_name]
    if architecture == "DeepLabV3Plus":
        encoder = enc_cls(in_channels, output_stride=16)
        decoder = DeepLabV3PlusDecoder(encoder.out_channels)
        head = SegmentationHead(decoder.out_channels, classes, kernel=1, upsampling=4)
    elif architecture == "Unet":
        encoder = enc_cls(in_channels, output_stride=32)
        decoder = UnetDecoder(encoder.out_channels)
        head = SegmentationHead(decoder.out_channels, classes, kernel=3, upsampling=1)
    elif architecture == "FPN":
        encoder = enc_cls(in_channels, output_stride=32)
        decoder = FPNDecoder(encoder.out_channels)
        head = SegmentationHead(decoder.out_channels, classes, kernel=1, upsampling=4)
    else:
        raise ValueError(f"Unsupported architecture: {architecture}")
    if architecture in ("DeepLabV3Plus", "FPN") and hasattr(encoder, "skip_stem_feature"):
        encoder.skip_stem_feature = True   # these decoders never read the stride-2 feature map
    return SegmentationModel(encoder, decoder, head, f"{architecture}-{encoder_name}")
