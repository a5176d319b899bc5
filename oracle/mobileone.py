"""Functional CPU restatement of the reference MobileOne stack over a plain ``state_dict``
(holocron/models/classification/mobileone.py:31-236, models/utils.py:28-86,114-146).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned by tests/golden/mobileone.pt (outputs, gradients and running statistics of reference MobileOneBlocks in training
and eval mode, their re-parametrised form, and one training step + re-parametrised inference of mobileone_s0, generated
from the reference itself).  ``emu=True`` rounds to bf16 exactly where the HIP path stores bf16 (branch planes, the stacked
point-wise conv output, block outputs, packed dense weights) for block-level comparison on the MI355X.
"""
import torch
import torch.nn.functional as F

from .repvgg import _round_weight, bf16r
from .rexnet import _bn, _rnd


def _is_bn(sd, p):
    return (p + ".weight") in sd and sd[p + ".weight"].dim() == 1


def depth_block(x, sd, prefix, stride, training, emu=False):
    """DepthConvBlock.forward (mobileone.py:66-67) or its folded form (one depth-wise 3x3 with bias, :69-98)."""
    if (prefix + ".weight") in sd:
        return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], stride, 1, 1, x.shape[1])
    out, i = 0, 0
    if _is_bn(sd, f"{prefix}.0"):
        out = _bn(x, x, sd, f"{prefix}.0", training)
        i = 1
    while f"{prefix}.{i}.0.weight" in sd:
        w = sd[f"{prefix}.{i}.0.weight"]
        c = F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, w.shape[0])
        out = out + _bn(c, _rnd(c, emu), sd, f"{prefix}.{i}.1", training)
        i += 1
    return out


def point_block(x, sd, prefix, training, emu=False):
    """PointConvBlock.forward (mobileone.py:120-121) or its folded form (one dense 1x1 with bias, :123-151)."""
    if (prefix + ".weight") in sd:
        w = sd[prefix + ".weight"]
        return F.conv2d(x, _round_weight(w) if emu else w, sd[prefix + ".bias"])
    out, i = 0, 0
    if _is_bn(sd, f"{prefix}.0"):
        out = _bn(x, x, sd, f"{prefix}.0", training)
        i = 1
    while f"{prefix}.{i}.0.weight" in sd:
        w = sd[f"{prefix}.{i}.0.weight"]
        c = F.conv2d(x, _round_weight(w) if emu else w)
        out = out + _bn(c, _rnd(c, emu), sd, f"{prefix}.{i}.1", training)
        i += 1
    return out


def block(x, sd, prefix, stride, training, emu=False):
    """MobileOneBlock: depth block -> ReLU -> point block -> ReLU (mobileone.py:154-176)."""
    if emu:
        x = _rnd(x, True)
    h = _rnd(F.relu(depth_block(x, sd, prefix + ".0", stride, training, emu)), emu)
    return _rnd(F.relu(point_block(h, sd, prefix + ".2", training, emu)), emu)


def forward(sd, x, num_blocks=(2, 8, 10, 1), training=False, emu=False):
    """MobileOne.forward (mobileone.py:184-226): stem block (stride 2), four stages (first block stride 2), GAP, Linear."""
    if emu:
        x = bf16r(x)
    h = block(x, sd, "features.0", 2, training, emu)
    for s, n in enumerate(num_blocks):
        for k in range(n):
            h = block(h, sd, f"features.{s + 1}.{k}", 2 if k == 0 else 1, training, emu)
    pooled = h.flatten(2).mean(2)
    return F.linear(pooled, sd["head.weight"], sd["head.bias"])
