"""MobileOne on the MI355X kernels (reference: holocron/models/classification/mobileone.py).

Same module tree, constructor arguments and ``state_dict`` keys as the reference: ``DepthConvBlock`` / ``PointConvBlock`` are
``nn.ModuleList`` s of parallel [conv, BatchNorm2d] branches (plus a bare BatchNorm2d when the shapes allow the identity),
``MobileOneBlock`` is ``Sequential(depth, act, point, act)``, ``MobileOne`` is ``features / pool / head``.  The training form
runs each block as one fused unit (holocron_amd/nn/mobileone_op.py); ``reparametrize()`` folds every block to a single
convolution with bias exactly like the reference (:69-98, :123-151) and the folded form runs on the same kernels.
"""
from collections import OrderedDict
from typing import Any, Callable, List, Optional, cast

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ...nn import GlobalAvgPool2d, init
from ...nn.convbn_op import ConvState, act_code, cl_ld
from ...nn.mbconv_op import _PadChannelsFn, ceil16
from ...nn.mobileone_op import BlockState, DepthRepFn, DwBiasActFn, PointBiasActFn, PointRepFn, _bn_info
from ...nn.repblock_op import POOL
from ..utils import conv_sequence, fuse_conv_bn

__all__ = ["DepthConvBlock", "PointConvBlock", "MobileOneBlock", "MobileOne", "mobileone_s0", "mobileone_s1", "mobileone_s2",
           "mobileone_s3"]


def _relu_code(act) -> int:
    """0 (no activation) | 1 (ReLU): the block kernels rebuild the activation mask from the stored output."""
    if act is None:
        return 0
    code = act_code(act)
    if code is None or code[0] != 1:
        raise NotImplementedError(f"MobileOne blocks on the HIP path fuse ReLU only, got {act}")
    return 1


def _enter(x: Tensor, channels: int) -> Tensor:
    """A logical [N, channels, H, W] tensor or an already padded activation -> dense NHWC bf16 with ceil16(channels)."""
    _lib.require_gpu(x)
    cp = ceil16(channels)
    if x.shape[1] == cp and cl_ld(x) == cp:
        return x
    if x.shape[1] != channels:
        raise ValueError(f"expected {channels} channels, got {x.shape[1]}")
    return _PadChannelsFn.apply(x, cp)


def _branches(block: nn.ModuleList):
    """(identity BatchNorm2d or None, [(conv, bn), ...]) of a multi-branch block."""
    mods = list(block)
    bn_id = mods[0] if isinstance(mods[0], nn.BatchNorm2d) else None
    pairs = []
    for seq in mods[(1 if bn_id is not None else 0):]:
        if not (isinstance(seq, nn.Sequential) and len(seq) == 2 and type(seq[0]) is nn.Conv2d and isinstance(seq[1], nn.BatchNorm2d)
                and seq[0].bias is None):
            raise NotImplementedError(f"branch outside the HIP path (expects [Conv2d(bias=False), BatchNorm2d]): {seq}")
        pairs.append((seq[0], seq[1]))
    return bn_id, pairs


class DepthConvBlock(nn.ModuleList):
    """Re-parametrizable depth-wise block (mobileone.py:31-98): BN(x) [stride 1] + dw1x1+BN + ``num_blocks`` x (dw3x3+BN)."""

    def __init__(self, channels: int, num_blocks: int, stride: int = 1,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        layers = [norm_layer(channels)] if stride == 1 else []
        layers.append(nn.Sequential(*conv_sequence(channels, channels, kernel_size=1, stride=stride, norm_layer=norm_layer,
                                                   groups=channels)))
        layers.extend([
            nn.Sequential(*conv_sequence(channels, channels, kernel_size=3, padding=1, stride=stride, norm_layer=norm_layer,
                                         groups=channels))
            for _ in range(num_blocks)
        ])
        super().__init__(layers)
        self._hcs = BlockState()

    def forward_padded(self, x: Tensor, act: int = 0) -> Tensor:
        bn_id, pairs = _branches(self)
        conv0 = pairs[0][0]
        channels, stride = conv0.in_channels, conv0.stride[0]
        for conv, _ in pairs:
            if not (conv.groups == conv.in_channels == conv.out_channels == channels and conv.stride == (stride, stride)
                    and conv.kernel_size in ((1, 1), (3, 3)) and conv.padding == ((conv.kernel_size[0] - 1) // 2,) * 2
                    and stride in (1, 2)):
                raise NotImplementedError(f"depth-wise branch outside the HIP path: {conv}")
        x = _enter(x, channels)
        bns = ([bn_id] if bn_id is not None else []) + [bn for _, bn in pairs]
        training = bns[0].training
        params = ([bn_id.weight, bn_id.bias] if bn_id is not None else [])
        for conv, bn in pairs:
            params += [conv.weight, bn.weight, bn.bias]
        meta = (stride, bn_id is not None, tuple(_bn_info(bn) for bn in bns), training, act, channels)
        out = DepthRepFn.apply(x, self._hcs, meta, *params)
        if self._hcs.last_out_stats is not None:
            out._hc_stats = self._hcs.last_out_stats
            self._hcs.last_out_stats = None
        return out

    def forward(self, x: Tensor) -> Tensor:
        c = x.shape[1]
        out = self.forward_padded(x)
        return out if out.shape[1] == c else out[:, :c]

    def reparametrize(self) -> nn.Conv2d:
        """One depth-wise 3x3 with bias equal to the eval-mode block (mobileone.py:69-98)."""
        bn_id, pairs = _branches(self)
        conv0 = pairs[0][0]
        chans = conv0.in_channels
        conv = nn.Conv2d(chans, chans, 3, padding=1, bias=True, stride=conv0.stride, groups=chans).to(conv0.weight.device)
        conv.weight.data.zero_()
        conv.bias.data.zero_()  # type: ignore[union-attr]
        if bn_id is not None:
            scale = bn_id.weight.data / torch.sqrt(bn_id.running_var + bn_id.eps)
            conv.bias.data += bn_id.bias.data - scale * bn_id.running_mean  # type: ignore[union-attr]
            conv.weight.data[..., 1, 1] += scale.unsqueeze(1)
        for c, bn in pairs:
            k, b = fuse_conv_bn(c, bn)
            conv.bias.data += b  # type: ignore[union-attr]
            if c.kernel_size == (1, 1):
                conv.weight.data[..., 1:2, 1:2] += k
            else:
                conv.weight.data += k
        return conv


class PointConvBlock(nn.ModuleList):
    """Re-parametrizable point-wise block (mobileone.py:101-151): BN(x) [equal widths] + ``num_blocks`` x (1x1+BN)."""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        layers = [norm_layer(out_channels)] if out_channels == in_channels else []
        layers.extend([
            nn.Sequential(*conv_sequence(in_channels, out_channels, kernel_size=1, norm_layer=norm_layer))
            for _ in range(num_blocks)
        ])
        super().__init__(layers)
        self._hcs = BlockState()

    def forward_padded(self, x: Tensor, act: int = 0) -> Tensor:
        bn_id, pairs = _branches(self)
        conv0 = pairs[0][0]
        cin, cout = conv0.in_channels, conv0.out_channels
        for conv, _ in pairs:
            if not (conv.groups == 1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
                    and conv.in_channels == cin and conv.out_channels == cout):
                raise NotImplementedError(f"point-wise branch outside the HIP path: {conv}")
        x = _enter(x, cin)
        bns = ([bn_id] if bn_id is not None else []) + [bn for _, bn in pairs]
        training = bns[0].training
        params = ([bn_id.weight, bn_id.bias] if bn_id is not None else [])
        for conv, bn in pairs:
            params += [conv.weight, bn.weight, bn.bias]
        meta = (bn_id is not None, tuple(_bn_info(bn) for bn in bns), training, act, cin, cout)
        out = PointRepFn.apply(x, self._hcs, meta, *params)
        if self._hcs.last_out_stats is not None:
            out._hc_stats = self._hcs.last_out_stats
            self._hcs.last_out_stats = None
        return out

    def forward(self, x: Tensor) -> Tensor:
        out = self.forward_padded(x)
        cout = _branches(self)[1][0][0].out_channels
        return out if out.shape[1] == cout else out[:, :cout]

    def reparametrize(self) -> nn.Conv2d:
        """One dense 1x1 with bias equal to the eval-mode block (mobileone.py:123-151)."""
        bn_id, pairs = _branches(self)
        conv0 = pairs[0][0]
        conv = nn.Conv2d(conv0.in_channels, conv0.out_channels, 1, bias=True).to(conv0.weight.device)
        conv.weight.data.zero_()
        conv.bias.data.zero_()  # type: ignore[union-attr]
        if bn_id is not None:
            scale = bn_id.weight.data / torch.sqrt(bn_id.running_var + bn_id.eps)
            conv.bias.data += bn_id.bias.data - scale * bn_id.running_mean  # type: ignore[union-attr]
            idx = torch.arange(conv.weight.shape[0], device=conv.weight.device)
            conv.weight.data[idx, idx, 0, 0] += scale
        for c, bn in pairs:
            k, b = fuse_conv_bn(c, bn)
            conv.bias.data += b  # type: ignore[union-attr]
            conv.weight.data += k
        return conv


class MobileOneBlock(nn.Sequential):
    """Depth block -> act -> point block -> act (mobileone.py:154-181)."""

    def __init__(self, in_channels: int, out_channels: int, overparam_factor: int = 1, stride: int = 1,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if act_layer is None:
            act_layer = nn.ReLU(inplace=True)
        super().__init__(
            DepthConvBlock(in_channels, overparam_factor, stride, norm_layer),
            act_layer,
            PointConvBlock(in_channels, out_channels, overparam_factor, norm_layer),
            act_layer,
        )

    def reparametrize(self) -> None:
        """Replace the depth-wise & point-wise blocks by their folded convolutions."""
        self[0] = self[0].reparametrize()
        self[2] = self[2].reparametrize()

    def forward_padded(self, x: Tensor) -> Tensor:
        depth, act1, point, act2 = self[0], self[1], self[2], self[3]
        if isinstance(depth, DepthConvBlock):
            h = depth.forward_padded(x, _relu_code(act1))
        else:                                                   # folded: depth-wise 3x3 + bias
            conv = cast(nn.Conv2d, depth)
            if not (type(conv) is nn.Conv2d and conv.groups == conv.in_channels == conv.out_channels and conv.kernel_size == (3, 3)
                    and conv.padding == (1, 1) and conv.stride[0] in (1, 2) and conv.bias is not None):
                raise NotImplementedError(f"folded depth block outside the HIP path: {conv}")
            if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
                raise NotImplementedError("the re-parametrised MobileOne block is an inference form; run it under torch.no_grad()")
            st = getattr(conv, "_hcs", None)
            if st is None:
                st = conv._hcs = BlockState()
            h = DwBiasActFn.apply(_enter(x, conv.in_channels), conv.weight, conv.bias, st, conv.stride[0], _relu_code(act1))
        if isinstance(point, PointConvBlock):
            return point.forward_padded(h, _relu_code(act2))
        conv = cast(nn.Conv2d, point)
        if not (type(conv) is nn.Conv2d and conv.groups == 1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
                and conv.padding == (0, 0) and conv.bias is not None):
            raise NotImplementedError(f"folded point block outside the HIP path: {conv}")
        if torch.is_grad_enabled() and (h.requires_grad or conv.weight.requires_grad):
            raise NotImplementedError("the re-parametrised MobileOne block is an inference form; run it under torch.no_grad()")
        st = getattr(conv, "_hcp", None)
        if st is None:
            st = conv._hcp = ConvState()
        return PointBiasActFn.apply(_enter(h, conv.in_channels), conv.weight, conv.bias, st, _relu_code(act2))

    def forward(self, x: Tensor) -> Tensor:  # type: ignore[override]
        out = self.forward_padded(x)
        point = self[2]
        cout = point.out_channels if isinstance(point, nn.Conv2d) else _branches(point)[1][0][0].out_channels
        return out if out.shape[1] == cout else out[:, :cout]


class MobileOne(nn.Sequential):
    """MobileOne (mobileone.py:184-236)."""

    def __init__(self, num_blocks: List[int], width_multipliers: List[float], overparam_factor: int = 1, num_classes: int = 10,
                 in_channels: int = 3, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if act_layer is None:
            act_layer = nn.ReLU(inplace=True)
        base_planes = [64, 128, 256, 512]
        planes = [round(mult * chans) for mult, chans in zip(width_multipliers, base_planes)]
        in_planes = min(64, planes[0])
        layers: List[nn.Module] = [MobileOneBlock(in_channels, in_planes, overparam_factor, 2, act_layer, norm_layer)]
        for _num_blocks, _planes in zip(num_blocks, planes):
            stage = [MobileOneBlock(in_planes, _planes, overparam_factor, 2, act_layer, norm_layer)]
            stage.extend([MobileOneBlock(_planes, _planes, overparam_factor, 1, act_layer, norm_layer)
                          for _ in range(_num_blocks - 1)])
            in_planes = _planes
            layers.append(nn.Sequential(*stage))
        super().__init__(OrderedDict([
            ("features", nn.Sequential(*layers)),
            ("pool", GlobalAvgPool2d(flatten=True)),
            ("head", nn.Linear(in_planes, num_classes)),
        ]))
        init.init_module(self, nonlinearity="relu")

    def reparametrize(self) -> None:
        """Fold conv + BN in every branch, then the branches of every block (mobileone.py:228-236)."""
        self.features: nn.Sequential
        self.features[0].reparametrize()
        for stage in self.features[1:]:
            for block in stage:
                block.reparametrize()

    def forward(self, x: Tensor) -> Tensor:  # type: ignore[override]
        _lib.require_gpu(x)
        POOL.begin(x.device)
        try:
            h = self.features[0].forward_padded(x)
            for stage in self.features[1:]:
                for block in stage:
                    h = block.forward_padded(h)
            pooled = self.pool(h)
            if pooled.shape[1] != self.head.in_features:
                pooled = pooled[:, :self.head.in_features]
            return self.head(pooled)
        finally:
            POOL.end()


def _mobileone(pretrained: bool, checkpoint: Any, width_multipliers: List[float], overparam_factor: int, **kwargs: Any) -> MobileOne:
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return MobileOne([2, 8, 10, 1], width_multipliers, overparam_factor, **kwargs)


def mobileone_s0(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> MobileOne:
    """MobileOne-S0 (mobileone.py:271-295)."""
    return _mobileone(pretrained, checkpoint, [0.75, 1.0, 1.0, 2.0], 4, **kwargs)


def mobileone_s1(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> MobileOne:
    """MobileOne-S1 (mobileone.py:319-343)."""
    return _mobileone(pretrained, checkpoint, [1.5, 1.5, 2.0, 2.5], 1, **kwargs)


def mobileone_s2(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> MobileOne:
    """MobileOne-S2 (mobileone.py:367-391)."""
    return _mobileone(pretrained, checkpoint, [1.5, 2.0, 2.5, 4.0], 1, **kwargs)


def mobileone_s3(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> MobileOne:
    """MobileOne-S3 (mobileone.py:415-439)."""
    return _mobileone(pretrained, checkpoint, [2.0, 2.5, 3.0, 4.0], 1, **kwargs)
