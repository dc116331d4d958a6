"""Plain-torch definitions of the segmentation networks the toolkit prunes.

The reference builds them with segmentation_models_pytorch 0.5.0 + timm 1.0.19
(pruning/models.py:23-43, pruning/model_configs.py:5-60), neither of which is available; these
modules restate the same architectures (encoder stages, decoder wiring, `segmentation_head`
attribute) with random initialisation.  Convolutions run on MIOpen; nothing here is a kernel.
All forwards are torch.fx-traceable, which is how spice_amd.engine derives coupled channels.
"""
from .factory import create_model, ENCODERS, ARCHITECTURES  # noqa: F401
