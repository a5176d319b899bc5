"""nn.Module wrappers of the hot path (reference: holocron/nn/modules/*.py)."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import functional as F

__all__ = ["HardMish", "GlobalAvgPool2d", "FocalLoss", "DiceLoss", "PolyLoss", "CrossEntropyLoss", "DropBlock2d", "SPP", "FReLU", "SlimConv2d", "NormConv2d", "ConcatDownsample2d"]


class HardMish(nn.Module):
    """HardMish activation (holocron/nn/modules/activation.py:28-38)."""

    def __init__(self, inplace: bool = False) -> None:
        super().__init__()
        self.inplace = inplace

    def forward(self, x: Tensor) -> Tensor:
        return F.hard_mish(x, inplace=self.inplace)

    def extra_repr(self) -> str:
        return "inplace=True" if self.inplace else ""


class GlobalAvgPool2d(nn.Module):
    """Global average pooling (holocron/nn/modules/downsample.py:57-77)."""

    def __init__(self, flatten: bool = False) -> None:
        super().__init__()
        self.flatten = flatten

    def forward(self, x: Tensor) -> Tensor:
        y = F.global_avg_pool2d(x)
        if self.flatten:
            return y
        return y.view(y.shape[0], y.shape[1], 1, 1)

    def extra_repr(self) -> str:
        return "flatten=True" if self.flatten else ""


class _Loss(nn.Module):
    def __init__(self, weight=None, ignore_index: int = -100, reduction: str = "mean") -> None:
        super().__init__()
        if isinstance(weight, (float, int)):
            self.register_buffer("weight", torch.Tensor([weight, 1 - weight]))
        elif isinstance(weight, list):
            self.register_buffer("weight", torch.Tensor(weight))
        else:
            self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        if reduction not in {"none", "mean", "sum"}:
            raise NotImplementedError("argument reduction received an incorrect input")
        self.reduction = reduction


class FocalLoss(_Loss):
    """Focal loss module (holocron/nn/modules/loss.py:50-84)."""

    def __init__(self, gamma: float = 2.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.gamma = gamma

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        return F.focal_loss(x, target, self.weight, self.ignore_index, self.reduction, self.gamma)

    def extra_repr(self) -> str:
        return f"gamma={self.gamma}, reduction='{self.reduction}'"


class DiceLoss(_Loss):
    """Dice loss module (holocron/nn/modules/loss.py:195-219)."""

    def __init__(self, weight=None, gamma: float = 1.0, eps: float = 1e-8) -> None:
        super().__init__(weight)
        self.gamma = gamma
        self.eps = eps

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        return F.dice_loss(x, target, self.weight, self.gamma, self.eps)

    def extra_repr(self) -> str:
        return f"reduction='{self.reduction}', gamma={self.gamma}, eps={self.eps}"


class PolyLoss(_Loss):
    """Poly-1 loss module (holocron/nn/modules/loss.py:222-246)."""

    def __init__(self, *args, eps: float = 2.0, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.eps = eps

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        return F.poly_loss(x, target, self.eps, self.weight, self.ignore_index, self.reduction)

    def extra_repr(self) -> str:
        return f"eps={self.eps}, reduction='{self.reduction}'"


class CrossEntropyLoss(nn.CrossEntropyLoss):
    """``torch.nn.CrossEntropyLoss`` (the criterion of references/classification/train.py:194) whose common case - ``[N, K]`` logits on
    the GPU, class-index targets, no class weights, mean reduction - runs as one HIP launch forward and one backward
    (``F.cross_entropy``); every other configuration is torch's own."""

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        if (x.is_cuda and x.dim() == 2 and target.dim() == 1 and not target.is_floating_point() and self.weight is None
                and self.reduction == "mean"):
            return F.cross_entropy(x, target, self.label_smoothing, self.ignore_index)
        return super().forward(x, target)


class DropBlock2d(nn.Module):
    """DropBlock module (holocron/nn/modules/dropblock.py:14-41); ``drop_prob = p / block_size**2`` and the
    functional divides by ``block_size**2`` again, exactly as the reference does."""

    def __init__(self, p: float = 0.1, block_size: int = 7, inplace: bool = False) -> None:
        super().__init__()
        self.p = p
        self.block_size = block_size
        self.inplace = inplace

    @property
    def drop_prob(self) -> float:
        return self.p / self.block_size**2

    def forward(self, x: Tensor) -> Tensor:
        return F.dropblock2d(x, self.drop_prob, self.block_size, self.inplace, self.training)

    def extra_repr(self) -> str:
        return f"p={self.p}, block_size={self.block_size}, inplace={self.inplace}"


class SPP(nn.ModuleList):
    """Spatial pyramid pooling (holocron/nn/modules/downsample.py:154-167): ``cat([x] + [maxpool_k(x)], dim=1)`` with
    stride 1 and same padding.  The HIP kernel covers the (5, 9, 13) pyramid YOLOv4 uses (yolov4.py:188)."""

    def __init__(self, kernel_sizes) -> None:
        super().__init__([nn.MaxPool2d(k, stride=1, padding=k // 2) for k in kernel_sizes])
        self.kernel_sizes = list(kernel_sizes)

    def forward(self, x: Tensor) -> Tensor:
        from ..ops.nhwc import spp_cl
        if self.kernel_sizes != [5, 9, 13]:
            raise NotImplementedError("the HIP SPP kernel implements the (5, 9, 13) pyramid only")
        return spp_cl(x)


class ConcatDownsample2d(nn.Module):
    """Loss-less downsampling by stacking the ``scale_factor`` x ``scale_factor`` neighbours on the channel axis
    (holocron/nn/modules/downsample.py:26-39, the YOLOv2 passthrough)."""

    def __init__(self, scale_factor: int) -> None:
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x: Tensor) -> Tensor:
        return F.concat_downsample2d(x, self.scale_factor)

    def forward_hip(self, x: Tensor) -> Tensor:       # unit protocol of run_conv_sequence
        return self.forward(x)


class FReLU(nn.Module):
    """Funnel activation (holocron/nn/modules/activation.py:58-82): ``max(x, bn(depthwise_conv(x)))``.  The depthwise
    3x3 conv, the BatchNorm passes and the max run on the HIP kernels (channel counts are padded to 16 internally)."""

    def __init__(self, in_channels: int, kernel_size: int = 3) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size, padding=kernel_size // 2, groups=in_channels)
        self.bn = nn.BatchNorm2d(in_channels)

    def forward(self, x: Tensor) -> Tensor:
        from .mbconv_op import _PadChannelsFn, ceil16, elementwise_max, padded_conv_bn_act
        if self.conv.kernel_size != (3, 3):
            raise NotImplementedError("the HIP depthwise kernel implements kernel_size=3 (the FReLU default)")
        c = x.shape[1]
        xp = _PadChannelsFn.apply(x, ceil16(c))
        out = elementwise_max(xp, padded_conv_bn_act(xp, self.conv, self.bn, None))
        return out if out.shape[1] == c else out[:, :c]


class SlimConv2d(nn.Module):
    """SlimConv2d (holocron/nn/modules/conv.py:262-370): channel gate from the pooled input, "fold" of the two halves
    weighted by the gate / the flipped gate, a kxk conv on the top path and 1x1 -> kxk on the bottom path, concatenated
    to 3C/4 channels.  Gate + fold is one kernel (hc_slim_fold_*), the convolutions are the padded gather-conv units."""

    def __init__(self, in_channels: int, kernel_size: int, stride: int = 1, padding: int = 0, dilation: int = 1, groups: int = 1,
                 bias: bool = True, padding_mode: str = "zeros", r: int = 32, L: int = 2) -> None:  # noqa: N803
        super().__init__()
        self.fc1 = nn.Conv2d(in_channels, max(in_channels // r, L), 1)
        self.bn = nn.BatchNorm2d(max(in_channels // r, L))
        self.fc2 = nn.Conv2d(max(in_channels // r, L), in_channels, 1)
        self.conv_top = nn.Conv2d(in_channels // 2, in_channels // 2, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.conv_bot1 = nn.Conv2d(in_channels // 2, in_channels // 4, 1)
        self.conv_bot2 = nn.Conv2d(in_channels // 4, in_channels // 4, kernel_size, stride, padding, dilation, groups, bias,
                                   padding_mode)

    def _gate_logits(self, pooled: Tensor) -> Tensor:
        from .mbconv_op import padded_conv_bias, padded_conv_bn_act
        z = padded_conv_bn_act(pooled, self.fc1, self.bn, nn.ReLU())      # bn(fc1(z)) then relu (conv.py:355-356)
        return padded_conv_bias(z, self.fc2)

    def forward(self, x: Tensor) -> Tensor:
        from .mbconv_op import SlimGateFn, _PadChannelsFn, ceil16, padded_conv_bias
        c = x.shape[1]
        if c % 4 or c != self.fc1.in_channels:
            raise ValueError("SlimConv2d expects the configured number of input channels (a multiple of 4)")
        for conv in (self.conv_top, self.conv_bot2):
            if conv.groups != 1 or conv.dilation != (1, 1) or conv.padding_mode != "zeros":
                raise NotImplementedError("SlimConv2d on the HIP path: groups=1, dilation=1, zero padding only")
        top, bot = SlimGateFn.apply(_PadChannelsFn.apply(x, ceil16(c)), self._gate_logits, c)
        top = padded_conv_bias(top, self.conv_top)
        bot = padded_conv_bias(padded_conv_bias(bot, self.conv_bot1), self.conv_bot2)
        return torch.cat((top[:, :c // 2], bot[:, :c // 4]), dim=1)


class NormConv2d(nn.modules.conv._ConvNd):
    """Normalised convolution (holocron/nn/modules/conv.py:55-147): every unfolded input patch is standardised (mean and
    biased variance over its Cin*KH*KW entries, ``eps`` inside the square root) before the filters are applied.  Same
    constructor, parameters and ``state_dict`` as the reference; see holocron_amd/nn/normconv_op.py for the kernels."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0, dilation: int = 1,
                 groups: int = 1, bias: bool = True, padding_mode: str = "zeros", eps: float = 1e-14) -> None:
        pair = nn.modules.utils._pair
        super().__init__(in_channels, out_channels, pair(kernel_size), pair(stride), pair(padding), pair(dilation), False, pair(0),
                         groups, bias, padding_mode)
        self.normalize_slices = False     # the reference stores False here (conv.py:118) and always normalises
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        from .normconv_op import norm_conv2d_module
        return norm_conv2d_module(x, self)
