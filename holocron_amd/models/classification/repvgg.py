"""RepVGG on the MI355X kernels (reference: holocron/models/classification/repvgg.py).

The module tree, parameter names and ``state_dict`` layout are the reference's
(``features.<stage>.<block>.branches.{0,1}.{0,1}.*``, ``branches.2.*``, ``head.*``), so reference
checkpoints load unchanged; ``forward`` bypasses the torch modules and runs the fused HIP path of
holocron_amd/nn/repblock_op.py.
"""
from collections import OrderedDict
from typing import Any, Callable, List, Optional, Union, cast

import torch
import torch.nn as nn

from ...nn import GlobalAvgPool2d, init
from ... import _lib
from ...nn.repblock_op import POOL, RepState, rep_block_forward
from ...ops import conv as cv
from ..utils import conv_sequence, fuse_conv_bn

__all__ = ["RepBlock", "RepVGG", "repvgg_a0", "repvgg_a1", "repvgg_a2", "repvgg_b0", "repvgg_b1", "repvgg_b2",
           "repvgg_b3"]


class RepBlock(nn.Module):
    """3x3 + 1x1 (+ identity) BatchNorm branches summed, then the activation (repvgg.py:38-73)."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1, identity: bool = True,
                 act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__()
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        self.activation = nn.ReLU(inplace=True) if act_layer is None else act_layer
        branches = [
            nn.Sequential(*conv_sequence(inplanes, planes, None, norm_layer, kernel_size=k, padding=k // 2, stride=stride))
            for k in (3, 1)
        ]
        if identity:
            if inplanes != planes:
                raise ValueError("The number of input and output channels must be identical if identity is used")
            branches.append(norm_layer(planes))
        self.branches: Union[nn.Conv2d, nn.ModuleList] = nn.ModuleList(branches)
        self._hc = RepState(stride, identity)
        self._rep_cache = cv.PackCache()

    def _fusable(self) -> bool:
        b = self.branches
        return (all(isinstance(m, nn.BatchNorm2d) for m in (b[0][1], b[1][1]) + ((b[2],) if len(b) == 3 else ()))
                and b[0][0].groups == 1 and b[0][0].dilation == (1, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if isinstance(self.branches, nn.Conv2d):
            conv = self.branches
            relu = isinstance(self.activation, nn.ReLU)
            if conv.weight.shape[1] % 16 != 0:
                out = cv.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], act=1 if relu else 0)
            else:
                wpk = self._rep_cache.get((conv.weight,), lambda: cv.pack_weight(conv.weight, 0))
                src = cv.to_cl_bf16(x)
                N, Cin, H, W = src.shape
                d = cv.fwd_desc(N, Cin, H, W, conv.weight.shape[0], 3, 3, conv.stride[0], conv.padding[0])
                out = cv.empty_cl(N, conv.weight.shape[0], d.OH, d.OW, src.device)
                cv.launch_conv(d, src, wpk, out, bias=conv.bias.detach(), act=1 if relu else 0)
            return out if relu else self.activation(out)
        if not self._fusable():
            raise NotImplementedError("RepBlock HIP path expects nn.BatchNorm2d branches and dense 3x3/1x1 convs")
        b = self.branches
        relu = isinstance(self.activation, nn.ReLU)
        st = self._hc
        out = rep_block_forward(x, b[0][0].weight, b[1][0].weight, b[0][1], b[1][1], b[2] if len(b) == 3 else None, st, relu)
        return out if relu else self.activation(out)

    def reparametrize(self) -> None:
        """Fold the BNs into their convs and the three branches into one 3x3 conv + bias
        (repvgg.py:75-107)."""
        if not isinstance(self.branches, nn.ModuleList):
            raise AssertionError
        conv3 = cast(nn.Conv2d, self.branches[0][0])
        planes, inplanes = conv3.weight.shape[0], conv3.weight.shape[1]
        k3, b3 = fuse_conv_bn(*self.branches[0])
        k1, b1 = fuse_conv_bn(*self.branches[1])
        rep = nn.Conv2d(inplanes, planes, 3, padding=1, bias=True, stride=conv3.stride).to(k3.device)
        kernel = k3.clone()
        kernel[..., 1:2, 1:2] += k1
        bias = b3 + b1
        if len(self.branches) == 3:
            bn0 = self.branches[2]
            scale = bn0.weight.data / (bn0.running_var + bn0.eps).sqrt()
            idx = torch.arange(planes, device=kernel.device)
            kernel[idx, idx, 1, 1] += scale
            bias = bias + bn0.bias.data - scale * bn0.running_mean
        rep.weight.data = kernel
        rep.bias.data = bias
        self.branches = rep


class RepVGG(nn.Sequential):
    """RepVGG (repvgg.py:110-171): every stage is one stride-2 block followed by ``nb`` stride-1
    identity blocks; global average pool; linear head."""

    def __init__(self, num_blocks: List[int], planes: List[int], width_multiplier: float,
                 final_width_multiplier: float, num_classes: int = 10, in_channels: int = 3,
                 act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        act_layer = nn.ReLU(inplace=True) if act_layer is None else act_layer
        if len(num_blocks) != len(planes):
            raise AssertionError("the length of `num_blocks` and `planes` are expected to be the same")
        widths = [in_channels, int(min(1, width_multiplier) * planes[0])]
        widths += [int(width_multiplier * c) for c in planes[1:-1]]
        widths.append(int(final_width_multiplier * planes[-1]))
        stages: List[nn.Sequential] = []
        for nb, cin, cout in zip(num_blocks, widths[:-1], widths[1:]):
            blocks = [RepBlock(cin, cout, 2, False, act_layer, norm_layer)]
            blocks += [RepBlock(cout, cout, 1, True, act_layer, norm_layer) for _ in range(nb)]
            for prod, cons in zip(blocks[:-1], blocks[1:]):
                prod._hc.emit_stats = cons._hc.identity  # producer emits the consumer's identity-BN statistics
            stages.append(nn.Sequential(*blocks))
        super().__init__(OrderedDict([
            ("features", nn.Sequential(*stages)),
            ("pool", GlobalAvgPool2d(flatten=True)),
            ("head", nn.Linear(widths[-1], num_classes)),
        ]))
        init.init_module(self, nonlinearity="relu")
        self.default_cfg = None

    def reparametrize(self) -> None:
        for stage in self.features:
            for block in stage:
                block.reparametrize()

    # ---- step-level host work: one multi-tensor weight pack + one zero-filled arena per forward ----
    def _pack_all(self) -> None:
        import ctypes as C
        import numpy as np
        blocks = [b for stage in self.features for b in stage if isinstance(b.branches, nn.ModuleList) and b._fusable()]
        stale = []
        for b in blocks:
            w3, w1 = b.branches[0][0].weight, b.branches[1][0].weight
            if not w3.is_cuda:
                return
            if b._hc.packed is None or b._hc.packed_key != RepState.weights_key(w3, w1):
                stale.append((b, w3, w1))
        if not stale:
            return
        items = []
        for b, w3, w1 in stale:
            items += b._hc.pack_items(w3, w1)
        sig = tuple((w.data_ptr(), dst.data_ptr()) for (w, dst, *_r) in items)
        cache = getattr(self, "_hc_pack_table", None)
        if cache is None or cache[0] != sig:
            from ...nn.repblock_op import fill_pack_items
            arr = (_lib.PackItem * len(items))()
            mx = fill_pack_items(arr, items)
            host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
            cache = (sig, host.to(items[0][0].device), len(items), mx)
            self._hc_pack_table = cache
        from ...ops.conv import profiled
        with profiled("weight_pack", 0.0, sum(6.0 * w.numel() for (w, *_r) in items)):     # fp32 master read, bf16 image written
            _lib.check(_lib.load().hc_pack_conv_weights_multi(cache[1].data_ptr(), cache[2], cache[3], _lib.stream()),
                       "hc_pack_conv_weights_multi")
        for b, w3, w1 in stale:
            b._hc.packed_key = RepState.weights_key(w3, w1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            return super().forward(x)   # the blocks raise the "no CPU path" error
        self._pack_all()
        POOL.begin(x.device)
        try:
            return super().forward(x)
        finally:
            POOL.end()


_CFG = {
    "repvgg_a0": ([1, 2, 4, 14, 1], 0.75, 2.5),
    "repvgg_a1": ([1, 2, 4, 14, 1], 1, 2.5),
    "repvgg_a2": ([1, 2, 4, 14, 1], 1.5, 2.75),
    "repvgg_b0": ([1, 4, 6, 16, 1], 1, 2.5),
    "repvgg_b1": ([1, 4, 6, 16, 1], 2, 4),
    "repvgg_b2": ([1, 4, 6, 16, 1], 2.5, 5),
    "repvgg_b3": ([1, 4, 6, 16, 1], 3, 5),
}


def _repvgg(arch: str, pretrained: bool, checkpoint: Any, progress: bool, **kwargs: Any) -> RepVGG:
    if pretrained or checkpoint is not None:
        raise RuntimeError("pretrained checkpoints need network access; load a reference state_dict with "
                           "model.load_state_dict(...) instead (same keys and shapes)")
    num_blocks, a, b = _CFG[arch]
    return RepVGG(num_blocks, [64, 64, 128, 256, 512], a, b, **kwargs)


def repvgg_a0(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-A0 (repvgg.py:206-232)."""
    return _repvgg("repvgg_a0", pretrained, checkpoint, progress, **kwargs)


def repvgg_a1(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-A1 (repvgg.py:256-282)."""
    return _repvgg("repvgg_a1", pretrained, checkpoint, progress, **kwargs)


def repvgg_a2(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-A2 (repvgg.py:306-332)."""
    return _repvgg("repvgg_a2", pretrained, checkpoint, progress, **kwargs)


def repvgg_b0(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-B0 (repvgg.py:356-382)."""
    return _repvgg("repvgg_b0", pretrained, checkpoint, progress, **kwargs)


def repvgg_b1(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-B1 (repvgg.py:406-432)."""
    return _repvgg("repvgg_b1", pretrained, checkpoint, progress, **kwargs)


def repvgg_b2(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-B2 (repvgg.py:456-482)."""
    return _repvgg("repvgg_b2", pretrained, checkpoint, progress, **kwargs)


def repvgg_b3(pretrained: bool = False, checkpoint: Any = None, progress: bool = True, **kwargs: Any) -> RepVGG:
    """RepVGG-B3 (repvgg.py:485-498)."""
    return _repvgg("repvgg_b3", pretrained, checkpoint, progress, **kwargs)
