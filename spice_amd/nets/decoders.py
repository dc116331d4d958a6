"""DeepLabV3+, U-Net and FPN decoders and the assembled segmentation models (the wiring of
segmentation_models_pytorch 0.5.0: encoder -> decoder -> segmentation_head)."""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers
from .layers import BatchNormAct2d, Conv1x1, ConvBnAct, DepthwiseConv3x3, conv1x1_bn


class SeparableConv2d(nn.Sequential):
    def __init__(self, in_ch: int, out_ch: int, kernel: int, padding: int = 0, dilation: int = 1,
                 bias: bool = True):
        if kernel == 3 and padding == dilation:
            depthwise = DepthwiseConv3x3(in_ch, dilation)
        else:
            depthwise = nn.Conv2d(in_ch, in_ch, kernel, padding=padding, dilation=dilation, groups=in_ch, bias=False)
        super().__init__(depthwise, nn.Conv2d(in_ch, out_ch, 1, bias=True) if bias else Conv1x1(in_ch, out_ch))


class BilinearUp(nn.UpsamplingBilinear2d):
    """nn.UpsamplingBilinear2d (bilinear, aligned corners) whose forward on a HIP device, when no
    gradient is wanted, is one launch of the library's up-sampling kernel instead of torch's.

    Under autocast torch runs this op in fp32 and returns fp32; so does this module, reading the
    fp16 input directly.  With `follow_input_dtype` the result is written in the input's dtype:
    for a consumer that autocast casts to fp16 anyway (a convolution, also behind a
    concatenation) the values it sees are the same, and the fp32 round trip through memory is
    gone.  With gradients, on the CPU or for other dtypes the torch op runs."""

    def __init__(self, scale_factor, follow_input_dtype: bool = False):
        super().__init__(scale_factor=scale_factor)
        self.follow_input_dtype = follow_input_dtype

    def forward(self, x, out=None):
        """`out`: where to write the result (a channel slice of a concatenation buffer; the kernel writes there
        directly, the torch path copies)."""
        wants_grad = torch.is_grad_enabled() and x.requires_grad
        # fp16 input is taken under autocast only, where torch would first cast it to fp32: torch's own fp16
        # kernel rounds its last multiply-add and the conversion in one step, which differs on ties
        if (not layers.library_forward() or not x.is_cuda or wants_grad or x.dim() != 4
                or not (x.dtype == torch.float32 or (x.dtype == torch.float16 and torch.is_autocast_enabled("cuda")))):
            return layers._into(out, super().forward(x))
        from .. import ops
        out_dtype = x.dtype
        if torch.is_autocast_enabled("cuda") and not self.follow_input_dtype:
            out_dtype = torch.float32
        if out is not None and out.dtype != out_dtype:
            return layers._into(out, ops.upsample_bilinear(x.contiguous(), scale_factor=float(self.scale_factor),
                                                           align_corners=True, out_dtype=out_dtype))
        return ops.upsample_bilinear(x.contiguous(), scale_factor=float(self.scale_factor), align_corners=True,
                                     out_dtype=out_dtype, out=out)


class BilinearLike(nn.Upsample):
    """F.interpolate(y, size=like.shape[-2:], mode="bilinear", align_corners=False): spreads the
    pooled 1x1 map of the ASPP image branch over the feature map.  Kernel launch under the same
    conditions as BilinearUp; the result keeps y's dtype (its consumer is the projection
    convolution behind the concatenation of the branches, fp16 under autocast either way)."""

    def __init__(self):
        super().__init__(mode="bilinear", align_corners=False)

    def forward(self, y, like, out=None):
        size = tuple(like.shape[-2:])
        wants_grad = torch.is_grad_enabled() and y.requires_grad
        if (not layers.library_forward() or not y.is_cuda or wants_grad or y.dim() != 4
                or not (y.dtype == torch.float32 or (y.dtype == torch.float16 and torch.is_autocast_enabled("cuda")))):
            return layers._into(out, F.interpolate(y, size=size, mode="bilinear", align_corners=False))
        from .. import ops
        if out is not None and out.dtype != y.dtype:
            return layers._into(out, ops.upsample_bilinear(y.contiguous(), size=size, align_corners=False))
        return ops.upsample_bilinear(y.contiguous(), size=size, align_corners=False, out=out)


def _fp16_inference(x) -> bool:
    """Real tensors on a HIP device, inference, fp16 autocast, the library's forward switched on."""
    return (isinstance(x, torch.Tensor) and layers.library_forward() and x.is_cuda and not torch.is_grad_enabled()
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16)


def _writes_in_place(x) -> bool:
    """Whether the producers of a concatenation may write their slices of it directly: fp16 inference on the
    library's kernels (every producer's result is fp16 there, as the consumer convolution's cast of the
    concatenation would make it)."""
    return _fp16_inference(x)


def _conv_bn_relu(conv: nn.Module, ch: int) -> nn.Sequential:
    return ConvBnAct(conv, BatchNormAct2d(ch, relu=True))


class ASPPPooling(nn.Module):
    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.conv = Conv1x1(in_ch, out_ch)
        self.bn = nn.BatchNorm2d(out_ch)
        self.relu = nn.ReLU()
        self.up = BilinearLike()

    def served_by_library(self, x) -> bool:
        """The whole branch in the library's two launches (ops.pooled_branch): inference under fp16 autocast on a
        HIP device, the modules as built and not listened to."""
        return (_fp16_inference(x) and x.dtype == torch.float16 and x.dim() == 4 and x.is_contiguous()
                and not self.bn.training and type(self.relu) is nn.ReLU and type(self.pool) is nn.AdaptiveAvgPool2d
                and self.conv.bias is None and self.conv.weight.is_cuda
                and not any(layers._hooked(m) for m in (self.pool, self.conv, self.bn, self.relu, self.up)))

    def forward(self, x, out=None):
        if self.served_by_library(x):
            from .. import ops
            return ops.pooled_branch(x, layers._fp16_matrix(self.conv.weight), self.bn, 1, out)
        y = self.relu(self.bn(self.conv(self.pool(x))))
        return self.up(y, x) if out is None else self.up(y, x, out=out)


class ASPP(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, rates: Sequence[int], separable: bool = True):
        super().__init__()
        branches = [_conv_bn_relu(Conv1x1(in_ch, out_ch), out_ch)]
        for r in rates:
            conv = (SeparableConv2d(in_ch, out_ch, 3, padding=r, dilation=r, bias=False) if separable
                    else nn.Conv2d(in_ch, out_ch, 3, padding=r, dilation=r, bias=False))
            branches.append(_conv_bn_relu(conv, out_ch))
        branches.append(ASPPPooling(in_ch, out_ch))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(Conv1x1(len(branches) * out_ch, out_ch),
                                     BatchNormAct2d(out_ch, relu=True), nn.Dropout(0.5))

    def forward(self, x):
        if _writes_in_place(x):
            # every branch writes its channels of the concatenation itself: no concatenation pass
            n, _, h, w = x.shape
            widths = [b[1].num_features if isinstance(b, ConvBnAct) else b.bn.num_features for b in self.convs]
            y = torch.empty(n, sum(widths), h, w, device=x.device, dtype=torch.float16)
            at = 0
            for b, c in zip(self.convs, widths):
                b(x, out=y[:, at:at + c])
                at += c
        else:
            y = torch.cat([b(x) for b in self.convs], dim=1)
        pr = self.project
        if type(pr) is not nn.Sequential or len(pr) != 3 or layers._hooked(pr):
            return pr(y)   # a replaced or listened-to container: called as the module it is
        return pr[2](conv1x1_bn(pr[0], pr[1], y))


class DeepLabV3PlusDecoder(nn.Module):
    def __init__(self, encoder_channels: Sequence[int], out_channels: int = 256,
                 rates: Sequence[int] = (12, 24, 36), highres_channels: int = 48):
        super().__init__()
        self.aspp = nn.Sequential(
            ASPP(encoder_channels[-1], out_channels, rates, separable=True),
            SeparableConv2d(out_channels, out_channels, 3, padding=1, bias=False),
            BatchNormAct2d(out_channels, relu=True))
        # the consumer (block2, behind the concatenation) is a convolution: fp16 under autocast either way
        self.up = BilinearUp(scale_factor=4, follow_input_dtype=True)
        self.block1 = _conv_bn_relu(Conv1x1(encoder_channels[2], highres_channels),
                                    highres_channels)
        self.block2 = _conv_bn_relu(
            SeparableConv2d(highres_channels + out_channels, out_channels, 3, padding=1, bias=False),
            out_channels)
        self.out_channels = out_channels

    #: a pass of a prefix.PrefixTape, attached by the caller for the duration of one forward (None: the plain forward)
    _tape = None

    def prefix_units(self, prefix: str = "decoder"):
        """The first units of this decoder in execution order (prefix.PrefixTape): the ASPP block, then the separable
        convolution + batch-norm behind it.  What follows (up-sampling, high-resolution branch, fusion block) writes
        into a buffer of its own per forward and always runs."""
        units = [(f"{prefix}.aspp.0", (f"{prefix}.aspp.0",)),
                 (f"{prefix}.aspp.tail", (f"{prefix}.aspp.1", f"{prefix}.aspp.2"))]
        self._unit_names = [n for n, _ in units]
        return units

    def _plain_aspp(self) -> bool:
        a = self.aspp
        return (type(a) is nn.Sequential and len(a) == 3 and type(a[1]) is SeparableConv2d and type(a[1][1]) is Conv1x1
                and not (layers._hooked(a) or layers._hooked(a[1])))

    def _aspp_tail(self, y):
        a = self.aspp   # separable 3x3 (depthwise, pointwise) -> batch-norm + ReLU: the pointwise pair in one call
        if self._plain_aspp():
            return conv1x1_bn(a[1][1], a[2], a[1][0](y))
        return a[2](a[1](y))

    def forward(self, features: List[torch.Tensor]):
        a = self.aspp   # ASPP -> separable 3x3 (depthwise, pointwise) -> batch-norm + ReLU
        tape = self._tape
        if tape is not None:
            deep = tape.run(self._unit_names[1], self._aspp_tail, tape.run(self._unit_names[0], a[0], features[-1]))
        elif self._plain_aspp():
            deep = self._aspp_tail(a[0](features[-1]))
        else:
            deep = a(features[-1])   # a replaced or listened-to container: called as the module it is
        skip = features[2]
        s = int(self.up.scale_factor)
        if (_writes_in_place(skip) and self.up.scale_factor == s and deep.dim() == 4
                and (deep.shape[2] * s, deep.shape[3] * s) == tuple(skip.shape[2:])):
            # both producers write their channels of the concatenation themselves: no concatenation pass
            c_deep, c_high = deep.shape[1], self.block1[1].num_features
            both = torch.empty(skip.shape[0], c_deep + c_high, skip.shape[2], skip.shape[3], device=skip.device,
                               dtype=torch.float16)
            self.up(deep, out=both[:, :c_deep])
            self.block1(skip, out=both[:, c_deep:])
            return self.block2(both)
        return self.block2(torch.cat([self.up(deep), self.block1(skip)], dim=1))


class UnetDecoderBlock(nn.Module):
    def __init__(self, in_ch: int, skip_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = _conv_bn_relu(nn.Conv2d(in_ch + skip_ch, out_ch, 3, padding=1, bias=False), out_ch)
        self.conv2 = _conv_bn_relu(nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False), out_ch)
        self.has_skip = skip_ch > 0

    def forward(self, x, skip=None):
        if (_fp16_inference(x) and x.dtype == torch.float16 and x.dim() == 4
                and (not self.has_skip or (skip.dtype == torch.float16 and skip.shape[0] == x.shape[0]
                                           and skip.shape[2] == 2 * x.shape[2] and skip.shape[3] == 2 * x.shape[3]))):
            # Autocast runs the nearest up-sampling (and with it the concatenation) in fp32 and the convolution casts
            # the result back: three passes over the block's widest tensor that change no value (copies of fp16
            # values).  The same copies in fp16, outside autocast's casting; the convolution reads them as they are.
            # ... and each straight into its channels of the concatenation: one strided copy of x with every pixel
            # named four times, one copy of the skip tensor (67 us where interpolate + cat take 147, [8,64,128,128] and
            # a 64-channel skip)
            n, c1, h, w = x.shape
            c2 = skip.shape[1] if self.has_skip else 0
            both = torch.empty(n, c1 + c2, 2 * h, 2 * w, device=x.device, dtype=torch.float16)
            both[:, :c1].view(n, c1, h, 2, w, 2).copy_(x.reshape(n, c1, h, 1, w, 1).expand(n, c1, h, 2, w, 2))
            if self.has_skip:
                both[:, c1:].copy_(skip)
            x = both
        else:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            if self.has_skip:
                x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels: Sequence[int], decoder_channels: Sequence[int] = (256, 128, 64, 32, 16)):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_chs = [enc[0]] + list(decoder_channels[:-1])
        skip_chs = enc[1:] + [0]
        self.blocks = nn.ModuleList(UnetDecoderBlock(i, s, o)
                                    for i, s, o in zip(in_chs, skip_chs, decoder_channels))
        self.out_channels = decoder_channels[-1]

    #: a pass of a prefix.PrefixTape, attached by the caller for the duration of one forward (None: the plain forward)
    _tape = None

    def prefix_units(self, prefix: str = "decoder"):
        """The units of this decoder in execution order: its blocks (prefix.PrefixTape)."""
        units = [(f"{prefix}.blocks.{i}", (f"{prefix}.blocks.{i}",)) for i in range(len(self.blocks))]
        self._unit_names = [n for n, _ in units]
        return units

    def forward(self, features: List[torch.Tensor]):
        feats = features[1:][::-1]
        x = feats[0]
        skips = feats[1:]
        tape = self._tape
        for i, block in enumerate(self.blocks):
            skip = skips[i] if i < len(skips) else None
            x = block(x, skip) if tape is None else tape.run(self._unit_names[i], block, x, skip)
        return x


class FPNBlock(nn.Module):
    def __init__(self, pyramid_ch: int, skip_ch: int):
        super().__init__()
        self.skip_conv = layers.Conv1x1Bias(skip_ch, pyramid_ch)   # (1x1 with bias: the library's kernel on the inference path)

    def forward(self, x, skip):
        return F.interpolate(x, scale_factor=2.0, mode="nearest") + self.skip_conv(skip)


class Conv3x3GNReLU(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, upsample: bool):
        super().__init__()
        self.up = BilinearUp(scale_factor=2) if upsample else None
        self.block = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False),
                                   nn.GroupNorm(32, out_ch), nn.ReLU(inplace=True))

    def forward(self, x):
        blk = self.block
        y = None
        if type(blk) is nn.Sequential and len(blk) == 3 and not (layers._hooked(blk) or layers._hooked(blk[1]) or layers._hooked(blk[2])):
            y = layers.conv3x3_plain(blk[0], x)   # the convolution on the library's kernel; group-norm and ReLU on torch's
        x = blk(x) if y is None else blk[2](blk[1](y))
        return x if self.up is None else self.up(x)


class FPNDecoder(nn.Module):
    def __init__(self, encoder_channels: Sequence[int], pyramid_ch: int = 256, seg_ch: int = 128):
        super().__init__()
        enc = list(encoder_channels)[::-1][:4]
        self.p5 = layers.Conv1x1Bias(enc[0], pyramid_ch)
        self.p4 = FPNBlock(pyramid_ch, enc[1])
        self.p3 = FPNBlock(pyramid_ch, enc[2])
        self.p2 = FPNBlock(pyramid_ch, enc[3])
        blocks = []
        for n_up in (3, 2, 1, 0):
            stack = [Conv3x3GNReLU(pyramid_ch, seg_ch, upsample=bool(n_up))]
            stack += [Conv3x3GNReLU(seg_ch, seg_ch, upsample=True) for _ in range(1, n_up)]
            blocks.append(nn.Sequential(*stack))
        self.seg_blocks = nn.ModuleList(blocks)
        self.dropout = nn.Dropout2d(0.2)
        self.out_channels = seg_ch

    def forward(self, features: List[torch.Tensor]):
        c2, c3, c4, c5 = features[-4:]
        p5 = self.p5(c5)
        p4 = self.p4(p5, c4)
        p3 = self.p3(p4, c3)
        p2 = self.p2(p3, c2)
        maps = [b(p) for b, p in zip(self.seg_blocks, (p5, p4, p3, p2))]
        return self.dropout(maps[0] + maps[1] + maps[2] + maps[3])


class SegmentationHead(nn.Sequential):
    def __init__(self, in_ch: int, classes: int, kernel: int, upsampling: int):
        up = BilinearUp(scale_factor=upsampling) if upsampling > 1 else nn.Identity()
        conv = (layers.Conv1x1Bias(in_ch, classes) if kernel == 1 else
                layers.Conv3x3Bias(in_ch, classes) if kernel == 3 else
                nn.Conv2d(in_ch, classes, kernel, padding=kernel // 2))
        super().__init__(conv, up, nn.Identity())


class SegmentationModel(nn.Module):
    """encoder -> decoder -> segmentation_head; the head's convolutions are never pruned
    (pruning/models.py:75-101 looks them up under the attribute name `segmentation_head`)."""

    def __init__(self, encoder: nn.Module, decoder: nn.Module, head: nn.Module, name: str):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.segmentation_head = head
        self.name = name

    def forward(self, x):
        return self.segmentation_head(self.decoder(self.encoder(x)))

    def prefix_units(self):
        """(unit name, prefixes of the names of its modules) in execution order, for prefix.PrefixTape: the encoder's
        units followed by the decoder's first ones.  Empty when the network's parts do not name any, or when the
        forward is not this class's (a subclass may run its parts in another order)."""
        if type(self).forward is not SegmentationModel.forward or not hasattr(self.encoder, "prefix_units"):
            return []
        units = list(self.encoder.prefix_units("encoder"))
        if hasattr(self.decoder, "prefix_units"):
            units += list(self.decoder.prefix_units("decoder"))
        return units

    def prefix_skip_units(self):
        """Names of the units whose outputs may be read far behind the unit that follows them (the encoder's feature
        maps); None when the encoder does not say.  The other units' outputs feed the next unit only."""
        if not self.prefix_units() or not hasattr(self.encoder, "prefix_skip_units"):
            return None
        return list(self.encoder.prefix_skip_units("encoder"))

    def attach_tape(self, tape_pass) -> None:
        """Route the next forwards through `tape_pass` (a pass of a prefix.PrefixTape built from prefix_units());
        None detaches."""
        if tape_pass is not None and not hasattr(self.encoder, "_unit_names"):
            self.prefix_units()   # a copy of the network that was never asked: its parts learn their unit names
        self.encoder._tape = tape_pass
        if hasattr(self.decoder, "prefix_units"):
            self.decoder._tape = tape_pass

    def forward_to_head(self, x):
        """The forward up to the head's convolution, for the kernels that up-sample on the fly (the tail of a sweep
        candidate, the arg-max histogram of an evaluation batch): (low, up) = the head convolution's fp16 output and
        the up-sampling module still to be applied, when the head is convolution -> nn.UpsamplingBilinear2d ->
        identity and the call is an fp16-autocast inference pass on a HIP device through the library's kernels;
        (logits, None) if the pass turned out ordinary after all; None, without having run anything, in every other
        situation (the caller then calls the model)."""
        head = self.segmentation_head
        if (not isinstance(x, torch.Tensor) or not x.is_cuda or torch.is_grad_enabled() or not layers.library_forward()
                or not torch.is_autocast_enabled("cuda") or torch.get_autocast_dtype("cuda") != torch.float16
                or type(self).forward is not SegmentationModel.forward
                or not isinstance(head, nn.Sequential) or len(head) != 3
                or not (isinstance(head[1], BilinearUp) or type(head[1]) is nn.Identity)
                or type(head[2]) is not nn.Identity or layers._hooked(head) or layers._hooked(head[1]) or layers._hooked(self)
                or torch.nn.modules.module._global_forward_hooks):
            return None
        low = head[0](self.decoder(self.encoder(x)))
        if type(head[1]) is nn.Identity:
            return low, None                # a head without up-sampling (the U-Nets): these are the logits
        if low.dtype != torch.float16 or low.dim() != 4:
            return head[1](low), None   # finished the ordinary way
        return low.contiguous(), head[1]
