"""ReXNet on the MI355X kernels (reference: holocron/models/classification/rexnet.py).

Same module tree and ``state_dict`` keys as the reference (``features.0/1`` stem, ``features.<i>.conv.<j>`` blocks with
the squeeze-excite block at ``conv.5.conv.{0,1,3}``, ``features.<last>``, ``head.1``).  Activations travel between the
units as NHWC bf16 with ``ceil16(C)`` channels per pixel (holocron_amd/nn/mbconv_op.py); every
[Conv2d, BatchNorm2d, act] run is one fused unit, the depthwise 3x3 has its own HBM-bound kernels, the squeeze-excite
gate and the ReLU6 after it are one pass, and the partial-width shortcut ``out[:, :Cin] += x`` (rexnet.py:140-141)
rides in the BatchNorm pass of the projection conv.
"""
import functools
import operator
from collections import OrderedDict
from math import ceil
from typing import Any, Callable, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ...nn import GlobalAvgPool2d
from ...nn import init
from ...nn.convbn_op import act_code, conv_bn_act
from ...nn.mbconv_op import (SeGateFn, _PadChannelsFn, ceil16, padded_conv_bias, padded_conv_bn_act, padded_model_scope, se_gate_fused,
                             se_mlp_fusable)
from ...nn.repblock_op import POOL
from ..utils import conv_sequence

__all__ = ["SEBlock", "ReXBlock", "ReXNet", "rexnet1_0x", "rexnet1_3x", "rexnet1_5x", "rexnet2_0x", "rexnet2_2x"]


class SEBlock(nn.Module):
    """Squeeze-excite (rexnet.py:38-66): GAP -> 1x1 (C/r) + BN + act -> 1x1 + bias -> sigmoid -> scale."""

    def __init__(self, channels: int, se_ratio: int = 12, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__()
        self.pool = GlobalAvgPool2d(flatten=False)
        self.conv = nn.Sequential(
            *conv_sequence(channels, channels // se_ratio, act_layer, norm_layer, drop_layer, kernel_size=1, stride=1,
                           bias=(norm_layer is None)),
            *conv_sequence(channels // se_ratio, channels, nn.Sigmoid(), None, drop_layer, kernel_size=1, stride=1),
        )

    def _gate_logits(self, pooled: Tensor) -> Tensor:
        """The two tiny convs on the pooled [N, Cp, 1, 1] tensor, without the final sigmoid (applied by the gate kernel)."""
        mods = list(self.conv)
        if not (len(mods) == 5 and isinstance(mods[1], nn.BatchNorm2d) and isinstance(mods[4], nn.Sigmoid)):
            raise NotImplementedError("SEBlock layout outside the HIP path (expects conv, norm, act, conv+bias, Sigmoid)")
        h = padded_conv_bn_act(pooled, mods[0], mods[1], mods[2])
        return padded_conv_bias(h, mods[3])

    def forward_gated(self, z: Tensor, act: int) -> Tensor:
        """act(z * sigmoid(mlp(mean(z)))) on a channel-padded activation; ``act`` 0 | 6 (ReLU6)."""
        mods = list(self.conv)
        if (len(mods) == 5 and isinstance(mods[4], nn.Sigmoid) and z.is_cuda and z.shape[1] == ceil16(mods[0].in_channels)
                and se_mlp_fusable(mods[0], mods[1], mods[2], mods[3])):
            return se_gate_fused(z, mods[0], mods[1], mods[2], mods[3], act)     # the MLP as 2 + 4 launches (csrc/se_mlp.hip)
        return SeGateFn.apply(z, self._gate_logits, act)

    def forward(self, x: Tensor) -> Tensor:
        c = x.shape[1]
        out = self.forward_gated(_PadChannelsFn.apply(x, ceil16(c)), 0)
        return out if out.shape[1] == c else out[:, :c]


class ReXBlock(nn.Module):
    def __init__(self, in_channels: int, channels: int, t: int, stride: int, use_se: bool = True, se_ratio: int = 12,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__()
        act_layer = nn.ReLU6(inplace=True) if act_layer is None else act_layer
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        self.use_shortcut = stride == 1 and in_channels <= channels
        self.in_channels = in_channels
        self.out_channels = channels
        b = norm_layer is None
        layers = []
        if t != 1:
            dw_channels = in_channels * t
            layers.extend(conv_sequence(in_channels, dw_channels, nn.SiLU(inplace=True), norm_layer, drop_layer,
                                        kernel_size=1, stride=1, bias=b))
        else:
            dw_channels = in_channels
        layers.extend(conv_sequence(dw_channels, dw_channels, None, norm_layer, drop_layer, kernel_size=3, stride=stride,
                                    padding=1, bias=b, groups=dw_channels))
        if use_se:
            layers.append(SEBlock(dw_channels, se_ratio, act_layer, norm_layer, drop_layer))
        layers.append(act_layer)
        layers.extend(conv_sequence(dw_channels, channels, None, norm_layer, drop_layer, kernel_size=1, stride=1, bias=b))
        self.conv = nn.Sequential(*layers)

    def _plan(self):
        """(expand | None, depthwise, se | None, act, project) from the reference's module order (rexnet.py:90-134)."""
        mods = list(self.conv)
        i = 0
        expand = None
        if mods[0].groups == 1:
            expand = (mods[0], mods[1], mods[2])
            i = 3
        dw = (mods[i], mods[i + 1])
        i += 2
        se = None
        if isinstance(mods[i], SEBlock):
            se = mods[i]
            i += 1
        act = mods[i]
        project = (mods[i + 1], mods[i + 2])
        if i + 3 != len(mods) or act_code(act) is None:
            raise NotImplementedError("ReXBlock layout outside the HIP path")
        return expand, dw, se, act, project

    def forward_padded(self, x: Tensor) -> Tensor:
        """Channel-padded in, channel-padded out."""
        expand, dw, se, act, project = self._plan()
        h = x
        if expand is not None:
            h = padded_conv_bn_act(h, *expand)
        if se is None:
            h = padded_conv_bn_act(h, dw[0], dw[1], act)
        else:
            code, _ = act_code(act)
            if code not in (0, 6):
                raise NotImplementedError("the squeeze-excite gate kernel fuses ReLU6 (or nothing) only")
            h = se.forward_gated(padded_conv_bn_act(h, dw[0], dw[1], None), code)
        return padded_conv_bn_act(h, project[0], project[1], None, residual=x if self.use_shortcut else None)

    def forward(self, x: Tensor) -> Tensor:
        _lib.require_gpu(x)
        out = self.forward_padded(_PadChannelsFn.apply(x, ceil16(self.in_channels)))
        return out if out.shape[1] == self.out_channels else out[:, :self.out_channels]


class ReXNet(nn.Sequential):
    def __init__(self, width_mult: float = 1.0, depth_mult: float = 1.0, num_classes: int = 1000, in_channels: int = 3,
                 in_planes: int = 16, final_planes: int = 180, use_se: bool = True, se_ratio: int = 12,
                 dropout_ratio: float = 0.2, bn_momentum: float = 0.9, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        act_layer = nn.SiLU(inplace=True) if act_layer is None else act_layer
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        num_blocks = [ceil(e * depth_mult) for e in [1, 2, 2, 3, 3, 5]]
        strides = functools.reduce(operator.iadd, [[s] + [1] * (num_blocks[i] - 1) for i, s in enumerate([1, 2, 2, 2, 1, 2])], [])
        depth = sum(num_blocks)
        stem_channel = 32 / width_mult if width_mult < 1.0 else 32
        inplanes = in_planes / width_mult if width_mult < 1.0 else in_planes
        chans = [round(width_mult * stem_channel)]
        chans.extend([round(width_mult * (inplanes + idx * final_planes / depth)) for idx in range(depth)])
        ses = [False] * (num_blocks[0] + num_blocks[1]) + [use_se] * sum(num_blocks[2:])
        b = norm_layer is None
        layers = conv_sequence(in_channels, chans[0], act_layer, norm_layer, drop_layer, kernel_size=3, stride=2, padding=1, bias=b)
        t = 1
        for in_c, c, s, se in zip(chans[:-1], chans[1:], strides, ses):
            layers.append(ReXBlock(in_channels=in_c, channels=c, t=t, stride=s, use_se=se, se_ratio=se_ratio))
            t = 6
        pen_channels = int(width_mult * 1280)
        layers.extend(conv_sequence(chans[-1], pen_channels, act_layer, norm_layer, drop_layer, kernel_size=1, stride=1,
                                    padding=0, bias=b))
        super().__init__(OrderedDict([
            ("features", nn.Sequential(*layers)),
            ("pool", GlobalAvgPool2d(flatten=True)),
            ("head", nn.Sequential(nn.Dropout(dropout_ratio), nn.Linear(pen_channels, num_classes))),
        ]))
        init.init_module(self, nonlinearity="relu")
        self.default_cfg = None

    def forward(self, x: Tensor) -> Tensor:  # type: ignore[override]
        _lib.require_gpu(x)
        POOL.begin(x.device)
        scope = padded_model_scope(self)
        entered = False
        try:
            scope.__enter__()          # may raise (weight repack): POOL.end() below must still run, the registry must not stay ours
            entered = True
            mods = list(self.features)
            blocks = [m for m in mods if isinstance(m, ReXBlock)]
            first, last = mods.index(blocks[0]), mods.index(blocks[-1])
            stem, tail = mods[:first], mods[last + 1:]
            if not (len(stem) == 3 and len(tail) == 3):
                raise NotImplementedError("ReXNet stem / penultimate layout outside the HIP path")
            h = conv_bn_act(x, stem[0], stem[1], stem[2])            # 3 -> 32 (im2col gather-conv)
            for blk in blocks:
                h = blk.forward_padded(h)
            h = padded_conv_bn_act(h, tail[0], tail[1], tail[2])
            pooled = self.pool(h)
            if pooled.shape[1] != tail[0].out_channels:
                pooled = pooled[:, :tail[0].out_channels]
            return self.head(pooled)
        finally:
            if entered:
                scope.__exit__(None, None, None)
            POOL.end()


def _rexnet(width_mult: float, depth_mult: float, pretrained: bool, checkpoint: Any, **kwargs: Any) -> ReXNet:
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return ReXNet(width_mult, depth_mult, **kwargs)


def rexnet1_0x(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> ReXNet:
    """ReXNet-1.0x (rexnet.py:341-367)."""
    return _rexnet(1, 1, pretrained, checkpoint, **kwargs)


def rexnet1_3x(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> ReXNet:
    """ReXNet-1.3x (rexnet.py:370-396)."""
    return _rexnet(1.3, 1, pretrained, checkpoint, **kwargs)


def rexnet1_5x(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> ReXNet:
    """ReXNet-1.5x (rexnet.py:399-425)."""
    return _rexnet(1.5, 1, pretrained, checkpoint, **kwargs)


def rexnet2_0x(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> ReXNet:
    """ReXNet-2.0x (rexnet.py:428-454)."""
    return _rexnet(2, 1, pretrained, checkpoint, **kwargs)


def rexnet2_2x(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> ReXNet:
    """ReXNet-2.2x (rexnet.py:457-483)."""
    return _rexnet(2.2, 1, pretrained, checkpoint, **kwargs)
