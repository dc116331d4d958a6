"""Building blocks shared by the encoders and decoders whose inference forward on a HIP device
is a launch of the library instead of a chain of torch kernels."""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

# SPICE_TORCH_FORWARD=1 keeps every module below (and the up-sampling modules of the decoders) on its
# torch path, for A/B measurements of the library's forward-side kernels.
_LIBRARY_FORWARD = os.environ.get("SPICE_TORCH_FORWARD", "0") != "1"


def library_forward() -> bool:
    return _LIBRARY_FORWARD


def set_library_forward(enabled: bool) -> None:
    global _LIBRARY_FORWARD
    _LIBRARY_FORWARD = bool(enabled)


class BatchNormAct2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with what follows it in a residual network folded into the module:
    y = relu?( bn(x) [+ residual] ).

    In eval mode under no_grad on a HIP device that is ONE kernel launch (x and the residual read
    once, y written once) instead of MIOpen's batch-norm, torch's add and torch's clamp; the kernel
    rounds every intermediate where those would have stored it, so the values are the same.  In
    every other situation (training, gradients wanted, CPU, fp16 planes with an odd element count,
    mixed dtypes) the torch ops run one after the other."""

    def __init__(self, num_features: int, relu: bool = False, silu: bool = False, **kwargs):
        super().__init__(num_features, **kwargs)
        if relu and silu:
            raise ValueError("one activation behind the batch-norm, not two")
        self.relu_after = relu
        self.silu_after = silu   # y = silu( bn(x) [+ residual] ): the EfficientNet blocks

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None,
                residual_norm: Optional[nn.BatchNorm2d] = None) -> torch.Tensor:
        """`residual_norm`: a batch-norm still to be applied to `residual` (the shortcut of a residual block
        with a projection): y = relu?( bn(x) + residual_norm(residual) ), one launch on the inference path."""
        inference = _LIBRARY_FORWARD and not (self.training or torch.is_grad_enabled()) and x.is_cuda
        if (inference and self.track_running_stats and self.affine and self.weight.dtype == torch.float32
                and self.running_mean.dtype == torch.float32):
            if ops.bn_act_supported(x) and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                                                  and residual.is_contiguous())):
                if residual_norm is None:
                    return ops.bn_act(x, self.running_mean, self.running_var, self.weight, self.bias, self.eps,
                                      residual=residual, relu=self.relu_after, act=2 if self.silu_after else 0)
                if (residual is not None and not self.silu_after and not residual_norm.training and residual_norm.track_running_stats
                        and residual_norm.affine and residual_norm.weight.dtype == torch.float32
                        and residual_norm.running_mean.dtype == torch.float32
                        and residual_norm.num_features == self.num_features):
                    return ops.bn_act_dual(x, self, residual, residual_norm, relu=self.relu_after)
        y = super().forward(x)
        if residual is not None:
            y = y + (residual if residual_norm is None else residual_norm(residual))
        if self.silu_after:
            return F.silu(y)
        return F.relu(y) if self.relu_after else y

    def extra_repr(self) -> str:
        return super().extra_repr() + f", relu={self.relu_after}" + (", silu=True" if self.silu_after else "")


class DepthwiseConv3x3(nn.Conv2d):
    """nn.Conv2d(ch, ch, 3, padding=dilation, dilation=dilation, groups=ch, bias=False).  On fp16
    activations (autocast) under no_grad on a HIP device, with dilation 1 or a multiple of 4, the
    forward is the library's streaming kernel: fp32 sums of the nine (exact) products in tap order,
    one rounding, which is what torch's own depthwise kernel computes (torch serves the dilated
    calls with it; the plain ones go to MIOpen's reference kernel, whose exact sum is within one
    fp16 step).  In every other situation torch's conv2d runs."""

    def __init__(self, channels: int, dilation: int = 1):
        super().__init__(channels, channels, 3, padding=dilation, dilation=dilation, groups=channels, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = self.dilation[0]
        plain = (self.dilation == (d, d) and self.padding == (d, d) and self.stride == (1, 1) and self.bias is None
                 and self.groups == self.in_channels == self.out_channels)
        if (_LIBRARY_FORWARD and not torch.is_grad_enabled() and x.is_cuda and (d == 1 or d % 4 == 0) and plain
                and (x.dtype == torch.float16 or torch.is_autocast_enabled("cuda"))):
            if x.dtype != torch.float16 and torch.get_autocast_dtype("cuda") == torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            if ops.depthwise_conv3x3_supported(x, self.weight):
                return ops.depthwise_conv3x3(x, self.weight, d)
        return super().forward(x)


class DepthwiseConvKxK(nn.Conv2d):
    """nn.Conv2d(ch, ch, k, stride, (k // 2) * dilation, dilation, groups=ch, bias=False), k = 3 or 5: the depthwise
    convolutions of the EfficientNet blocks.  On fp16 activations (autocast) under no_grad on a HIP device, stride 1 or
    2, dilation 1 or 2 and an even width, the forward is the library's tiled kernel: fp32 sums of the k * k (exact)
    products in tap order, one rounding - the arithmetic of torch's own depthwise kernel (MIOpen serves most of these
    calls image by image with an unfold pass and a matrix product).  In every other situation torch's conv2d runs."""

    def __init__(self, channels: int, kernel: int, stride: int = 1, dilation: int = 1):
        super().__init__(channels, channels, kernel, stride, (kernel // 2) * dilation, dilation, groups=channels, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        k, s, d = self.kernel_size[0], self.stride[0], self.dilation[0]
        pad = (k // 2) * d
        plain = (self.kernel_size == (k, k) and self.stride == (s, s) and self.dilation == (d, d)
                 and self.padding == (pad, pad) and self.padding_mode == "zeros" and self._parameters["bias"] is None
                 and self.groups == self.in_channels == self.out_channels)
        if (_LIBRARY_FORWARD and not torch.is_grad_enabled() and x.is_cuda and plain
                and (x.dtype == torch.float16 or torch.is_autocast_enabled("cuda"))):
            if x.dtype != torch.float16 and torch.get_autocast_dtype("cuda") == torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            w = self._parameters["weight"]
            if ops.depthwise_conv_supported(x, w, s, d):
                return ops.depthwise_conv(x, w, s, d)
        return super().forward(x)


def depthwise_bn(conv: "DepthwiseConvKxK", bn: BatchNormAct2d, x):
    """bn(conv(x)) for a depthwise k x k convolution and the batch-norm (+ ReLU / SiLU) behind it as ONE launch of the
    library's tiled kernel on the inference path of a HIP device; the two modules one after the other in every other
    situation (hooks, training, shapes the kernel declines)."""
    if (isinstance(x, torch.Tensor) and _LIBRARY_FORWARD and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and not torch.nn.modules.module._global_forward_hooks and not (_hooked(conv) or _hooked(bn))
            and type(conv) is DepthwiseConvKxK and _plain_inference_norm(bn)
            and (x.dtype == torch.float16 or
                 (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16))):
        k, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
        pad = (k // 2) * d
        if (conv.kernel_size == (k, k) and conv.stride == (s, s) and conv.dilation == (d, d) and conv.padding == (pad, pad)
                and conv.padding_mode == "zeros" and conv._parameters["bias"] is None
                and conv.groups == conv.in_channels == conv.out_channels == bn.num_features):
            if x.dtype != torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            w = conv._parameters["weight"]
            if ops.depthwise_conv_supported(x, w, s, d):
                return ops.depthwise_conv(x, w, s, d, bn, 2 if bn.silu_after else (1 if bn.relu_after else 0))
    return bn(conv(x))


# fp16 copies of convolution weights, per Parameter object: id -> (weak reference, version, [Cout, Cin * kh * kw] fp16).
# The same bookkeeping autocast does for the weights it casts: one cast per Parameter and version, dropped
# with the Parameter.
_FP16_WEIGHTS = {}


def _fp16_matrix(weight: torch.Tensor) -> torch.Tensor:
    key = id(weight)
    ent = _FP16_WEIGHTS.get(key)
    if ent is None or ent[0]() is not weight or ent[1] != weight._version:
        w16 = weight.detach().to(torch.float16).reshape(weight.shape[0], -1)
        ent = (weakref.ref(weight, lambda _, k=key: _FP16_WEIGHTS.pop(k, None)), weight._version, w16)
        _FP16_WEIGHTS[key] = ent
    return ent[2]


def _within_kernel_offsets(cin: int, cout: int, taps: int, height: int, width: int, stride: int) -> bool:
    """The limits of the library's matrix-core convolution beyond the layout ones: it addresses the weights and one
    image of x with 32-bit byte offsets (both below 2^30 - 2^20 bytes) and takes at most 2^28 result pixels per image.
    Past them its entry points decline (SPICE_ERR_UNSUPPORTED), so the predicates below must say "torch's conv2d"."""
    limit = (1 << 30) - (1 << 20)
    return (cout * taps * cin * 2 <= limit and cin * height * width * 2 <= limit
            and (height // stride) * (width // stride) <= (1 << 28) and cin <= (1 << 16) and cout <= (1 << 20))


class StemConv7x7(nn.Conv2d):
    """nn.Conv2d(in_ch, out_ch, 7, stride=2, padding=3, bias=False): the ResNet stem.  On fp16 activations (autocast)
    under no_grad on a HIP device, an even height and a width that is a multiple of 16, the forward is one launch of
    the library's matrix-core kernel, which picks the 49 shifted, column-halved views of x itself (MIOpen serves this
    convolution with an NHWC kernel between two layout transposes and a zero-fill).  Otherwise torch's conv2d."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__(in_channels, out_channels, 7, stride=2, padding=3, bias=False)

    def served_by_library(self, x_dtype: torch.dtype, height: int, width: int) -> bool:
        """Whether forward() on a plane of this size is a launch of the library (no MIOpen call, no solver lookup)."""
        return (_LIBRARY_FORWARD and _LIBRARY_CONV7X7 and self.bias is None and self.groups == 1
                and self.kernel_size == (7, 7) and self.stride == (2, 2) and self.padding == (3, 3)
                and self.dilation == (1, 1) and self.padding_mode == "zeros" and height % 2 == 0 and width % 16 == 0
                and self.in_channels <= 1024
                and _within_kernel_offsets(self.in_channels, self.out_channels, 49, height, width, 2)
                and (x_dtype == torch.float16 or
                     (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and not _hooked(self)
                and self.served_by_library(x.dtype, x.shape[2], x.shape[3])):
            x16 = x if x.dtype == torch.float16 else x.to(torch.float16)   # autocast's cast of a convolution's input
            return ops.conv7x7_s2(x16.contiguous(), _fp16_matrix(self.weight))
        return super().forward(x)


class Conv1x1(nn.Conv2d):
    """nn.Conv2d(in_ch, out_ch, 1, bias=False).  On fp16 activations (autocast) under no_grad on a HIP
    device the forward is the matrix product it is, W[Cout,Cin] @ X[n][Cin,HW] (fp16 inputs, fp32
    accumulation, as MIOpen's 1x1 solvers): through the library's matrix-core kernel or its hipBLASLt
    entry point with stride 1, and through the matrix-core kernel, which picks the even pixels itself,
    with stride 2 (MIOpen serves that case with a strided copy, two layout transposes and a product).
    In every other situation torch's conv2d runs."""

    def __init__(self, in_channels: int, out_channels: int, stride: int = 1):
        super().__init__(in_channels, out_channels, 1, stride=stride, bias=False)

    def _plain_fp16(self, x_dtype: torch.dtype) -> bool:
        return (_LIBRARY_FORWARD and self._parameters["bias"] is None and self.groups == 1 and self.padding == (0, 0)
                and (x_dtype == torch.float16 or
                     (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16)))

    def takes_matrix_path(self, x_dtype: torch.dtype) -> bool:
        return self.stride == (1, 1) and self._plain_fp16(x_dtype)

    def served_by_library(self, x_dtype: torch.dtype, height: int, width: int) -> bool:
        """Whether forward() on a plane of this size is a launch of the library (no MIOpen call, no solver lookup)."""
        if self.takes_matrix_path(x_dtype):
            return True
        return (self.stride == (2, 2) and self._plain_fp16(x_dtype) and height % 2 == 0 and width % 16 == 0
                and _within_kernel_offsets(self.in_channels, self.out_channels, 1, height, width, 2))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4:
            if self.takes_matrix_path(x.dtype):
                if x.dtype != torch.float16:
                    x = x.to(torch.float16)   # what autocast does to a convolution's input
                if x.is_contiguous():
                    return ops.conv1x1_f16(x, _fp16_matrix(self.weight))
            elif self.served_by_library(x.dtype, x.shape[2], x.shape[3]):
                # the projection shortcut of a stage that halves the resolution: the product over the even pixels
                if x.dtype != torch.float16:
                    x = x.to(torch.float16)
                if x.is_contiguous():
                    return ops.conv1x1_f16(x, _fp16_matrix(self.weight), stride=2)
        return super().forward(x)


class Conv1x1Bias(nn.Conv2d):
    """nn.Conv2d(in_ch, out_ch, 1) with its bias: the convolution of a segmentation head.  On fp16 activations
    (autocast) under no_grad on a HIP device, planes of a multiple of 8 pixels, the forward is one launch of the
    library's matrix-core kernel with the bias added where a batch-norm's shift would be (MIOpen serves this
    convolution through a solver it looks up per channel count, and a sweep candidate that prunes the layer in front
    brings a new count).  In every other situation torch's conv2d runs."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__(in_channels, out_channels, 1, bias=True)

    def served_by_library(self, x_dtype: torch.dtype, height: int, width: int) -> bool:
        """Whether forward() on a plane of this size is a launch of the library (no MIOpen call, no solver lookup)."""
        b = self._parameters["bias"]
        return (_LIBRARY_FORWARD and b is not None and b.dtype == torch.float32 and self.groups == 1
                and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                and (height * width) % 8 == 0
                and (x_dtype == torch.float16 or
                     (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16))
                and _within_kernel_offsets(self.in_channels, self.out_channels, 1, height, width, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
                and self.served_by_library(x.dtype, x.shape[2], x.shape[3])):
            if x.dtype != torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            if x.is_contiguous():
                return ops.conv1x1_bias(x, _fp16_matrix(self.weight), self._parameters["bias"])
        return super().forward(x)


def _plain_inference_norm(bn: nn.BatchNorm2d) -> bool:
    # parameters and buffers straight from their tables (nn.Module's attribute fallback is several dictionary probes
    # per name; this runs once per batch-norm per forward)
    return (not bn.training and bn.track_running_stats and bn.affine
            and bn._parameters["weight"].dtype == torch.float32 and bn._buffers["running_mean"].dtype == torch.float32)


# Where the unit that is running may put its result: set by a pass that stops in front of a captured launch graph
# (nets/prefix.py) to that graph's input buffer for the unit, so that the unit's last launch writes there and no copy
# into the buffer follows.  A unit that can do so takes it (once); every other unit leaves it, and the copy is made.
_unit_destination: Optional[torch.Tensor] = None


def set_unit_destination(t: Optional[torch.Tensor]) -> None:
    global _unit_destination
    _unit_destination = t


def take_unit_destination(shape, device) -> Optional[torch.Tensor]:
    """The pending destination if it is a contiguous fp16 tensor of that shape on that device (then it is used up)."""
    global _unit_destination
    t = _unit_destination
    if t is None or t.dtype != torch.float16 or tuple(t.shape) != tuple(shape) or t.device != device or not t.is_contiguous():
        return None
    _unit_destination = None
    return t


def _hooked(m: nn.Module) -> bool:
    return bool(m._forward_hooks or m._forward_pre_hooks)


def _into(out, y):
    """`y` written into `out` when one was given (a channel slice of a concatenation buffer)."""
    if out is None:
        return y
    out.copy_(y)
    return out


def conv1x1_bn(conv: Conv1x1, bn: BatchNormAct2d, x, residual=None, residual_norm: Optional[nn.BatchNorm2d] = None,
               out: Optional[torch.Tensor] = None, pre_norm: Optional[BatchNormAct2d] = None):
    """bn(conv(x) [, residual [, residual_norm]]).  On the inference path of a HIP device, where `conv` is the
    library's matrix product and `bn` the library's fused tail, the pair is ONE call of the extension (one launch of
    the matrix-core kernel that normalises the product before storing it; two launches where that kernel does not
    take the shapes) instead of two module calls.  Modules carrying forward hooks, symbolic tracing, training and
    every other situation take the modules one after the other.  `out`: where to write the result (a channel slice
    of a concatenation buffer); the one-launch route writes there directly, every other route copies.
    `pre_norm`: x is the raw result of the convolution before and this batch-norm (+ its ReLU) is still to be applied
    to it: bn(conv(pre_norm(x)) ...).  The one-launch route applies it to x on the way in (no launch, no round trip
    of the activated tensor through memory); every other route calls the module first."""
    if (isinstance(x, torch.Tensor) and _LIBRARY_FORWARD and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and not torch.nn.modules.module._global_forward_hooks
            and not (_hooked(conv) or _hooked(bn)) and conv.takes_matrix_path(x.dtype) and _plain_inference_norm(bn)
            and (x.shape[2] * x.shape[3]) % 2 == 0):
        ok = residual is None or (residual.dtype == torch.float16 and residual.is_contiguous()
                                  and residual.shape[1] == conv.out_channels and residual.shape[2:] == x.shape[2:]
                                  and residual.shape[0] == x.shape[0])
        if residual_norm is not None:
            ok = ok and residual is not None and _plain_inference_norm(residual_norm) and not _hooked(residual_norm) \
                and residual_norm.num_features == bn.num_features
        if out is not None:
            ok = ok and out.dtype == torch.float16
        pre = None
        if pre_norm is not None and x.dtype == torch.float16 and _plain_inference_norm(pre_norm) and not _hooked(pre_norm) \
                and pre_norm.num_features == conv.in_channels and not pre_norm.silu_after:
            pre = pre_norm   # the extension falls back to a launch of its own where the kernel cannot take it along
        elif pre_norm is not None:
            x = pre_norm(x)
        if ok:
            if x.dtype != torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            if x.is_contiguous():
                return ops.conv1x1_bn_act(x, _fp16_matrix(conv._parameters["weight"]), bn, residual, residual_norm, bn.relu_after, out=out,
                                          pre_bn=pre, pre_relu=pre.relu_after if pre is not None else False,
                                          act=2 if bn.silu_after else 0)
        if pre is not None:
            x = pre_norm(x)
    elif pre_norm is not None:
        x = pre_norm(x)
    if residual is None:
        return _into(out, bn(conv(x)))
    if residual_norm is None:
        return _into(out, bn(conv(x), residual))
    return _into(out, bn(conv(x), residual, residual_norm))


# fp16 tap-major copies of 3x3 convolution weights, [Cout, 9 * Cin], kept like _FP16_WEIGHTS
_FP16_TAPS = {}


def _fp16_taps(weight: torch.Tensor) -> torch.Tensor:
    key = id(weight)
    ent = _FP16_TAPS.get(key)
    if ent is None or ent[0]() is not weight or ent[1] != weight._version:
        # the cast and the move of the taps in front of the channels in ONE copy launch (a sweep candidate makes this
        # copy of its sliced 3x3 weights once per evaluation)
        cout, cin = weight.shape[0], weight.shape[1]
        w9 = torch.empty((cout, 3, 3, cin), dtype=torch.float16, device=weight.device)
        w9.copy_(weight.detach().permute(0, 2, 3, 1))
        w9 = w9.view(cout, 9 * cin)
        ent = (weakref.ref(weight, lambda _, k=key: _FP16_TAPS.pop(k, None)), weight._version, w9)
        _FP16_TAPS[key] = ent
    return ent[2]


# SPICE_CONV3X3=0 keeps the 3x3 convolutions with MIOpen (A/B measurements)
_LIBRARY_CONV7X7 = os.environ.get("SPICE_CONV7X7", "1") != "0"
_LIBRARY_CONV3X3 = os.environ.get("SPICE_CONV3X3", "1") != "0"


def conv3x3_served_by_library(conv: nn.Conv2d, x_dtype: torch.dtype, height: int, width: int) -> bool:
    """Whether conv3x3_bn takes this convolution on a plane of this size (then no MIOpen call, no solver lookup)."""
    s, d = conv.stride[0], conv.dilation[0]
    return (_LIBRARY_FORWARD and _LIBRARY_CONV3X3 and type(conv) is nn.Conv2d and conv.kernel_size == (3, 3)
            and conv.stride == (s, s) and s in (1, 2) and conv.dilation == (d, d) and 1 <= d <= 7 and conv.padding == (d, d)
            and conv.groups == 1 and conv._parameters["bias"] is None and conv.padding_mode == "zeros"
            and (x_dtype == torch.float16 or
                 (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16))
            and height % s == 0 and width % s == 0 and (width // s) % 8 == 0 and width >= 8
            and _within_kernel_offsets(conv.in_channels, conv.out_channels, 9, height, width, s))


def conv3x3_bn(conv: nn.Conv2d, bn: BatchNormAct2d, x):
    """bn(conv(x)) for a 3x3 convolution and the batch-norm (+ ReLU) behind it as ONE launch of the library's
    matrix-core kernel, on the inference path of a HIP device; None where that does not apply (hooks, tracing,
    training, other shapes), for the caller to run the modules."""
    if (isinstance(x, torch.Tensor) and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and not torch.nn.modules.module._global_forward_hooks and not (_hooked(conv) or _hooked(bn))
            and _plain_inference_norm(bn) and conv3x3_served_by_library(conv, x.dtype, x.shape[2], x.shape[3])):
        if x.dtype != torch.float16:
            x = x.to(torch.float16)   # what autocast does to a convolution's input
        if x.is_contiguous():
            return ops.conv3x3_bn_act(x, _fp16_taps(conv._parameters["weight"]), bn, bn.relu_after, conv.stride[0], conv.dilation[0],
                                      2 if bn.silu_after else 0)
    return None


class Conv3x3Bias(nn.Conv2d):
    """nn.Conv2d(in_ch, out_ch, 3, padding=1) with its bias: the convolution of a segmentation head that has no
    up-sampling behind it (the U-Nets).  On fp16 activations (autocast) under no_grad on a HIP device, a width that is
    a multiple of 8, the forward is one launch of the library's matrix-core kernel with the bias added where a
    batch-norm's shift would be.  In every other situation torch's conv2d runs."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__(in_channels, out_channels, 3, padding=1, bias=True)

    def served_by_library(self, x_dtype: torch.dtype, height: int, width: int) -> bool:
        """Whether forward() on a plane of this size is a launch of the library (no MIOpen call, no solver lookup)."""
        b = self._parameters["bias"]
        return (_LIBRARY_FORWARD and _LIBRARY_CONV3X3 and b is not None and b.dtype == torch.float32 and self.groups == 1
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.padding_mode == "zeros" and width % 8 == 0 and width >= 8
                and (x_dtype == torch.float16 or
                     (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16))
                and _within_kernel_offsets(self.in_channels, self.out_channels, 9, height, width, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
                and self.served_by_library(x.dtype, x.shape[2], x.shape[3])):
            if x.dtype != torch.float16:
                x = x.to(torch.float16)   # what autocast does to a convolution's input
            if x.is_contiguous():
                return ops.conv3x3_bias(x, _fp16_taps(self._parameters["weight"]), self._parameters["bias"])
        return super().forward(x)


def conv3x3_plain(conv: nn.Conv2d, x):
    """conv(x) for a 3x3 convolution without bias that no batch-norm follows (the FPN decoder's, in front of a
    group-norm) as one launch of the library's matrix-core kernel on the inference path of a HIP device; None where
    that does not apply, for the caller to run the module."""
    if (isinstance(x, torch.Tensor) and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and not torch.nn.modules.module._global_forward_hooks and not _hooked(conv)
            and conv3x3_served_by_library(conv, x.dtype, x.shape[2], x.shape[3])):
        if x.dtype != torch.float16:
            x = x.to(torch.float16)   # what autocast does to a convolution's input
        if x.is_contiguous():
            return ops.conv3x3_plain(x, _fp16_taps(conv._parameters["weight"]), conv.stride[0], conv.dilation[0])
    return None


class ConvBnAct(nn.Sequential):
    """Sequential(convolution, BatchNormAct2d); a 1x1 convolution and its batch-norm go through conv1x1_bn."""

    def __init__(self, conv: nn.Module, bn: BatchNormAct2d):
        super().__init__(conv, bn)

    def forward(self, x, out: Optional[torch.Tensor] = None):
        # (the sub-modules from their table: nn.Sequential's indexing walks an iterator per call)
        mods = self._modules
        conv, bn = mods["0"], mods["1"]
        if type(conv) is Conv1x1:
            return conv1x1_bn(conv, bn, x, out=out)
        if isinstance(conv, nn.Sequential) and len(conv._modules) == 2:
            inner = conv._modules
            if type(inner.get("1")) is Conv1x1 and "0" in inner:
                return conv1x1_bn(inner["1"], bn, inner["0"](x), out=out)   # separable: depthwise, then the pointwise pair
        if type(conv) is nn.Conv2d and conv.kernel_size == (3, 3):
            y = conv3x3_bn(conv, bn, x)   # None where the library's 3x3 kernel does not apply
            if y is not None:
                return _into(out, y)
        return _into(out, bn(conv(x)))
