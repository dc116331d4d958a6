"""EfficientNet-B3 feature encoder (width 1.2, depth 1.4; symmetric padding, SiLU, SE from the
block input width), the architecture timm 1.0.19 builds for "efficientnet_b3"."""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn

from .. import ops
from . import layers
from .layers import BatchNormAct2d, Conv1x1, DepthwiseConvKxK, conv1x1_bn, conv3x3_bn, depthwise_bn


def _round_channels(ch: float, divisor: int = 8) -> int:
    new = max(divisor, int(ch + divisor / 2) // divisor * divisor)
    if new < 0.9 * ch:
        new += divisor
    return new


class SqueezeExcite(nn.Module):
    def __init__(self, channels: int, reduced: int):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, reduced, 1)
        self.act1 = nn.SiLU(inplace=True)
        self.conv_expand = nn.Conv2d(reduced, channels, 1)

    def served_by_library(self, x_dtype: torch.dtype) -> bool:
        """Whether forward() on fp16 activations is the library's four short launches (no MIOpen call, no solver lookup for
        the two 1x1 convolutions)."""
        a, b = self.conv_reduce, self.conv_expand
        return (layers.library_forward() and type(a) is nn.Conv2d and type(b) is nn.Conv2d
                and a.kernel_size == (1, 1) and b.kernel_size == (1, 1) and a.stride == (1, 1) and b.stride == (1, 1)
                and a.padding == (0, 0) and b.padding == (0, 0) and a.groups == 1 and b.groups == 1
                and a.bias is not None and b.bias is not None and a.weight.dtype == torch.float32
                and a.in_channels == b.out_channels and a.out_channels == b.in_channels
                and a.in_channels <= 4096 and a.out_channels <= 1024 and type(self.act1) is nn.SiLU
                and x_dtype == torch.float16
                and not (layers._hooked(a) or layers._hooked(b) or layers._hooked(self.act1)))

    def forward(self, x):
        if (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.is_contiguous()
                and self.served_by_library(x.dtype)):
            return ops.squeeze_excite(x, self.conv_reduce.weight, self.conv_reduce.bias, self.conv_expand.weight,
                                      self.conv_expand.bias)
        s = x.mean((2, 3), keepdim=True)
        s = self.conv_expand(self.act1(self.conv_reduce(s)))
        return x * torch.sigmoid(s)


class DepthwiseSeparable(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel: int, stride: int, dilation: int = 1):
        super().__init__()
        self.has_skip = stride == 1 and in_chs == out_chs
        self.conv_dw = DepthwiseConvKxK(in_chs, kernel, stride, dilation)
        self.bn1 = BatchNormAct2d(in_chs, silu=True)   # batch-norm + SiLU: one launch on the inference path
        self.act1 = nn.Identity()
        self.se = SqueezeExcite(in_chs, max(1, round(in_chs * 0.25)))
        self.conv_pw = Conv1x1(in_chs, out_chs)
        self.bn2 = BatchNormAct2d(out_chs)

    def forward(self, x):
        y = self.se(self.act1(depthwise_bn(self.conv_dw, self.bn1, x)))   # depthwise + batch-norm + SiLU: one launch
        # projection + batch-norm (+ the block's input): one launch of the matrix-core kernel on the inference path
        return conv1x1_bn(self.conv_pw, self.bn2, y, x if self.has_skip else None)


class InvertedResidual(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel: int, stride: int, expand: int = 6,
                 dilation: int = 1):
        super().__init__()
        mid = in_chs * expand
        self.has_skip = stride == 1 and in_chs == out_chs
        self.conv_pw = Conv1x1(in_chs, mid)
        self.bn1 = BatchNormAct2d(mid, silu=True)
        self.act1 = nn.Identity()
        self.conv_dw = DepthwiseConvKxK(mid, kernel, stride, dilation)
        self.bn2 = BatchNormAct2d(mid, silu=True)
        self.act2 = nn.Identity()
        self.se = SqueezeExcite(mid, max(1, round(in_chs * 0.25)))
        self.conv_pwl = Conv1x1(mid, out_chs)
        self.bn3 = BatchNormAct2d(out_chs)

    def forward(self, x):
        # on the inference path: expansion + batch-norm + SiLU and projection + batch-norm (+ the block's input) one
        # launch of the matrix-core kernel each, depthwise convolution + batch-norm + SiLU one launch of the tiled kernel
        y = self.act1(conv1x1_bn(self.conv_pw, self.bn1, x))
        y = self.se(self.act2(depthwise_bn(self.conv_dw, self.bn2, y)))
        return conv1x1_bn(self.conv_pwl, self.bn3, y, x if self.has_skip else None)


# (block type, repeats, out channels, kernel, stride) of EfficientNet-B0 before scaling
_B0 = (("ds", 1, 16, 3, 1), ("ir", 2, 24, 3, 2), ("ir", 2, 40, 5, 2), ("ir", 3, 80, 3, 2),
       ("ir", 3, 112, 5, 1), ("ir", 4, 192, 5, 2), ("ir", 1, 320, 3, 1))


class EfficientNetB3Encoder(nn.Module):
    """Returns [x, /2 (40), /4 (32), /8 (48), /16 (136), /32 or /16 (384)]."""

    out_channels = (3, 40, 32, 48, 136, 384)

    def __init__(self, in_channels: int = 3, output_stride: int = 32, width: float = 1.2,
                 depth: float = 1.4):
        super().__init__()
        stem = _round_channels(32 * width)
        self.conv_stem = nn.Conv2d(in_channels, stem, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNormAct2d(stem, silu=True)
        self.act1 = nn.Identity()
        stages = []
        prev = stem
        cur_stride, dilation = 2, 1
        for kind, rep, ch, k, s in _B0:
            out = _round_channels(ch * width)
            blocks = []
            for i in range(int(math.ceil(rep * depth))):
                stride = s if i == 0 else 1
                if stride > 1 and cur_stride >= output_stride:  # trade stride for dilation
                    dilation *= stride
                    stride = 1
                cur_stride *= stride
                d = dilation if (stride == 1 and dilation > 1) else 1
                if kind == "ds":
                    blocks.append(DepthwiseSeparable(prev, out, k, stride, d))
                else:
                    blocks.append(InvertedResidual(prev, out, k, stride, 6, d))
                prev = out
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)

    #: a pass of a prefix.PrefixTape, attached by the caller for the duration of one forward (None: the plain forward)
    _tape = None

    def prefix_units(self, prefix: str = "encoder"):
        """The units of this encoder in execution order: the stem, then every block (prefix.PrefixTape)."""
        units = [(f"{prefix}.stem", (f"{prefix}.conv_stem", f"{prefix}.bn1", f"{prefix}.act1"))]
        for i, stage in enumerate(self.blocks):
            units += [(f"{prefix}.blocks.{i}.{j}", (f"{prefix}.blocks.{i}.{j}",)) for j in range(len(stage))]
        self._unit_names = [n for n, _ in units]
        return units

    def prefix_skip_units(self, prefix: str = "encoder"):
        """The units whose outputs leave the encoder as feature maps: the stem and the last block of stages 1, 2, 4, 6."""
        return [f"{prefix}.stem"] + [f"{prefix}.blocks.{i}.{len(self.blocks[i]) - 1}" for i in (1, 2, 4, 6)]

    def _stem(self, x):
        y = conv3x3_bn(self.conv_stem, self.bn1, x)   # convolution + batch-norm + SiLU in one launch where that applies
        return self.act1(self.bn1(self.conv_stem(x)) if y is None else y)

    def _forward_taped(self, x, tape) -> List[torch.Tensor]:
        names = iter(self._unit_names)
        h = tape.run(next(names), self._stem, x)
        feats = [x, h]
        for i, stage in enumerate(self.blocks):
            for block in stage:
                h = tape.run(next(names), block, h)
            if i in (1, 2, 4, 6):   # the stages whose outputs are the feature maps
                feats.append(h)
        return feats

    def forward(self, x) -> List[torch.Tensor]:
        if self._tape is not None:
            return self._forward_taped(x, self._tape)
        f0 = x
        f1 = self._stem(x)
        y = self.blocks[1](self.blocks[0](f1))
        f2 = y
        f3 = self.blocks[2](f2)
        f4 = self.blocks[4](self.blocks[3](f3))
        f5 = self.blocks[6](self.blocks[5](f4))
        return [f0, f1, f2, f3, f4, f5]
