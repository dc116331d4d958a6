"""ResNet-50 feature encoder (torchvision layout, optionally dilated to output stride 16)."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import ops
from . import layers
from .layers import BatchNormAct2d, Conv1x1, conv1x1_bn, conv3x3_bn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, dilation: int = 1,
                 downsample: bool = False):
        super().__init__()
        self.conv1 = Conv1x1(inplanes, planes)
        self.bn1 = BatchNormAct2d(planes, relu=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation,
                               bias=False)
        self.bn2 = BatchNormAct2d(planes, relu=True)
        self.conv3 = Conv1x1(planes, planes * 4)
        self.bn3 = BatchNormAct2d(planes * 4, relu=True)   # applied to bn3(x) + identity
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(
                Conv1x1(inplanes, planes * 4, stride=stride),
                nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        m = self._modules   # the sub-modules from their table: nn.Module's attribute fallback is several probes per name
        conv2, bn2, conv3, bn3, downsample = m["conv2"], m["bn2"], m["conv3"], m["bn3"], m.get("downsample")
        if downsample is not None and (type(downsample) is not nn.Sequential or len(downsample) != 2
                                       or layers._hooked(downsample)):
            # a shortcut that is not the plain (convolution, batch-norm) pair, or one somebody listens to: called as
            # the module it is (its hooks fire, a replacement's forward runs)
            h2 = self.bn2(self.conv2(self.bn1(self.conv1(x))))
            return self.bn3(self.conv3(h2), downsample(x))
        h1 = conv1x1_bn(m["conv1"], m["bn1"], x)
        out = conv3x3_bn(conv2, bn2, h1)   # one launch of the library's kernel at inference, else None
        if out is not None:
            # (the block's result straight into a captured graph's input buffer where a pass has named one)
            dest = None
            if layers._unit_destination is not None:
                dest = layers.take_unit_destination((out.shape[0], conv3.out_channels, out.shape[2], out.shape[3]), out.device)
            if downsample is None:
                return conv1x1_bn(conv3, bn3, out, x, out=dest)
            ds = downsample._modules
            return conv1x1_bn(conv3, bn3, out, ds["0"](x), ds["1"], out=dest)
        # the 3x3 convolution on torch's route: bn2 (+ ReLU) rides along with conv3 at inference, whose kernel applies it
        # to the convolution's raw result on the way in
        raw = self.conv2(h1)
        if self.downsample is None:
            return conv1x1_bn(self.conv3, self.bn3, raw, x, pre_norm=self.bn2)
        if not isinstance(x, torch.Tensor):
            # symbolic tracing (the channel-coupling analysis): the shortcut as the plain composition it is
            return self.bn3(self.conv3(self.bn2(raw)), self.downsample(x))
        # the shortcut's batch-norm rides along with bn3: relu(bn3(.) + bn_ds(conv_ds(x))) in one launch at inference
        return conv1x1_bn(self.conv3, self.bn3, raw, self.downsample[0](x), self.downsample[1], pre_norm=self.bn2)


class ResNet50Encoder(nn.Module):
    """Returns [x, /2 (64), /4 (256), /8 (512), /16 (1024), /32 or /16 (2048)]."""

    out_channels = (3, 64, 256, 512, 1024, 2048)

    def __init__(self, in_channels: int = 3, output_stride: int = 32):
        super().__init__()
        if output_stride not in (16, 32):
            raise ValueError("output_stride must be 16 or 32")
        self.conv1 = layers.StemConv7x7(in_channels, 64)
        self.bn1 = BatchNormAct2d(64, relu=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(64, 3, stride=1, dilation=1)
        self.layer2 = self._make_layer(128, 4, stride=2, dilation=1)
        self.layer3 = self._make_layer(256, 6, stride=2, dilation=1)
        if output_stride == 16:  # stride of the last stage replaced by dilation 2
            self.layer4 = self._make_layer(512, 3, stride=1, dilation=2)
        else:
            self.layer4 = self._make_layer(512, 3, stride=2, dilation=1)

    def _make_layer(self, planes: int, blocks: int, stride: int, dilation: int) -> nn.Sequential:
        layers = [Bottleneck(self.inplanes, planes, stride, dilation, downsample=True)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, 1, dilation))
        return nn.Sequential(*layers)

    #: set by a model whose decoder never reads the first feature map (DeepLabV3+): the stem then does not store it
    skip_stem_feature = False
    #: a pass of a prefix.PrefixTape, attached by the caller for the duration of one forward (None: the plain forward)
    _tape = None
    _STAGES = ("layer1", "layer2", "layer3", "layer4")

    def prefix_units(self, prefix: str = "encoder"):
        """The units of this encoder in execution order: the stem, then every bottleneck (prefix.PrefixTape)."""
        units = [(f"{prefix}.stem", (f"{prefix}.conv1", f"{prefix}.bn1", f"{prefix}.maxpool"))]
        for stage in self._STAGES:
            units += [(f"{prefix}.{stage}.{j}", (f"{prefix}.{stage}.{j}",)) for j in range(len(self._modules[stage]))]
        self._unit_names = [n for n, _ in units]
        return units

    def prefix_skip_units(self, prefix: str = "encoder"):
        """The units whose outputs leave the encoder as feature maps (a decoder may read them far behind the unit that
        follows): the stem and the last bottleneck of every stage."""
        return [f"{prefix}.stem"] + [f"{prefix}.{stage}.{len(self._modules[stage]) - 1}" for stage in self._STAGES]

    def _stem(self, x):
        """(bn1(conv1(x)), its max-pooling).  On the inference path of a HIP device the batch-norm, the ReLU and the
        pooling are ONE launch over the convolution's result (ops.bn_act_maxpool: same values as the separate ops);
        modules carrying hooks, symbolic tracing, training and every other situation run them one after the other."""
        y = self.conv1(x)
        mp = self.maxpool
        if (isinstance(y, torch.Tensor) and layers.library_forward() and not torch.is_grad_enabled() and y.is_cuda
                and not torch.nn.modules.module._global_forward_hooks
                and not (layers._hooked(self.bn1) or layers._hooked(mp)) and layers._plain_inference_norm(self.bn1)
                and type(mp) is nn.MaxPool2d and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
                and mp.dilation == 1 and not mp.ceil_mode and not mp.return_indices
                and ops.bn_act_maxpool_supported(y)):
            return ops.bn_act_maxpool(y, self.bn1, relu=self.bn1.relu_after, want_full=not self.skip_stem_feature)
        f1 = self.bn1(y)
        return f1, mp(f1)

    def _forward_taped(self, x, tape) -> List[torch.Tensor]:
        names = iter(self._unit_names)
        f1, h = tape.run(next(names), self._stem, x)
        feats = [x, f1]
        for stage in self._STAGES:
            for block in self._modules[stage]._modules.values():
                h = tape.run(next(names), block, h)
            feats.append(h)
        return feats

    def forward(self, x) -> List[torch.Tensor]:
        if self._tape is not None:
            return self._forward_taped(x, self._tape)
        f0 = x
        f1, pooled = self._stem(x)
        f2 = self.layer1(pooled)
        f3 = self.layer2(f2)
        f4 = self.layer3(f3)
        f5 = self.layer4(f4)
        return [f0, f1, f2, f3, f4, f5]
