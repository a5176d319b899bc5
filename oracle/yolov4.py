"""Functional CPU restatement of the reference YOLOv4 / CSP-Darknet stack over a plain ``state_dict``
(holocron/models/classification/darknetv4.py:37-182, darknetv3.py:23-70, resnet.py:59-87,
holocron/models/detection/yolov4.py:31-229,444-640, models/utils.py:28-86, nn/functional.py:465-500).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned by tests/golden/yolo.pt: in fp32 mode the losses / gradients / running statistics of one training step of a
reduced YOLOv4 reproduce the reference's.  ``emulate_bf16=True`` injects bf16 rounding exactly where the HIP path
stores bf16 (conv outputs, unit outputs, packed weights, their gradients) so that the MI355X result can be compared
layer-for-layer instead of through 70 layers of accumulated storage rounding.
"""
import torch
import torch.nn.functional as F

from . import yolo as oy
from .repvgg import BN_EPS, BN_MOMENTUM, _RoundBoth, _round_weight, bf16r

CSP53 = [(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)]
ANCHORS = torch.tensor([[[12, 16], [19, 36], [40, 28]], [[36, 75], [76, 55], [72, 146]],
                        [[142, 110], [192, 243], [459, 401]]], dtype=torch.float32) / 608
SCALE_XY = (1.2, 1.1, 1.05)


class Cfg:
    """act: "mish" | "leaky" (slope); drop: (p, block_size) of DropBlock2d or None; noise: iterator over the uniform
    draws in call order, or an object with ``draw(shape)`` (None -> torch.rand)."""

    def __init__(self, act="mish", slope=0.01, drop=(0.1, 7), noise=None, training=True, emulate_bf16=False):
        self.act, self.slope, self.drop, self.training, self.emu = act, slope, drop, training, emulate_bf16
        self.noise = noise
        self.step = 2 + (1 if act is not None else 0) + (1 if drop is not None else 0)

    def draw(self, shape):
        if self.noise is None:
            return torch.rand(shape)
        if hasattr(self.noise, "draw"):          # replay object: draw(shape)
            return self.noise.draw(shape)
        if not hasattr(self.noise, "__next__"):
            self.noise = iter(self.noise)
        return next(self.noise)


def _rnd(x, cfg):
    return _RoundBoth.apply(x) if cfg.emu else x


def _act(z, cfg):
    if cfg.act == "mish":
        return F.mish(z)
    if cfg.act == "leaky":
        return F.leaky_relu(z, cfg.slope)
    return z


def _drop_factor(x, cfg):
    """DropBlock2d(p, bs).forward in training mode as a per-pixel factor (functional.py:476-491; the module passes
    p / bs**2 and the functional divides by bs**2 again)."""
    p, bs = cfg.drop
    gamma = (p / bs**2) / bs**2
    noise = cfg.draw((x.shape[0],) + tuple(x.shape[2:]))
    mask = (noise <= gamma).to(x.dtype)
    mask = 1 - F.max_pool2d(mask, kernel_size=(bs, bs), stride=(1, 1), padding=bs // 2)
    ones = mask.sum()
    scale = mask.numel() / ones if ones > 0 else 1.0
    return mask.unsqueeze(1), scale


def unit(x, sd, prefix, idx, stride, pad, cfg, residual=None, drop=True):
    """[Conv2d(bias=False), BatchNorm2d, act, DropBlock2d?] of conv_sequence, residual added after the drop."""
    w = sd[f"{prefix}.{idx}.weight"]
    bn = f"{prefix}.{idx + 1}"
    c = F.conv2d(x, _round_weight(w) if cfg.emu else w, None, stride, pad)
    y = _rnd(c, cfg)
    if cfg.training:
        sd[bn + ".num_batches_tracked"] += 1
        mean, var = c.mean((0, 2, 3)), c.var((0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = c.numel() / c.shape[1]
            sd[bn + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[bn + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
    else:
        mean, var = sd[bn + ".running_mean"], sd[bn + ".running_var"]
    a = sd[bn + ".weight"] * torch.rsqrt(var + BN_EPS)
    z = _act(y * a.view(1, -1, 1, 1) + (sd[bn + ".bias"] - a * mean).view(1, -1, 1, 1), cfg)
    if drop and cfg.drop is not None and cfg.training:
        m, s = _drop_factor(z, cfg)
        z = z * m * s
    if residual is not None:
        z = z + residual
    return _rnd(z, cfg)


def res_block(x, sd, prefix, cfg):
    """darknetv3.ResBlock: 1x1 -> 3x3, += identity, then its own DropBlock2d (p = 0.1, block 7 unless the caller
    changed the modules; ``cfg.drop`` describes all DropBlock2d instances alike) when a drop layer is configured."""
    h = unit(x, sd, prefix + ".conv", 0, 1, 0, cfg)
    out = unit(h, sd, prefix + ".conv", cfg.step, 1, 1, cfg, residual=x)
    if cfg.drop is not None and cfg.training:
        m, s = _drop_factor(out, cfg)
        out = _rnd(_rnd(out * m, cfg) * s, cfg)
    return out


def csp_stage(x, sd, prefix, num_blocks, cfg):  # darknetv4.py:37-115
    x = unit(x, sd, prefix + ".base_layer", 0, 2, 1, cfg)
    x = unit(x, sd, prefix + ".base_layer", cfg.step, 1, 0, cfg)
    x1, x2 = x.chunk(2, dim=1)
    for b in range(num_blocks):
        x2 = res_block(x2, sd, f"{prefix}.main.{b}", cfg)
    x2 = unit(x2, sd, prefix + ".main", num_blocks, 1, 0, cfg)
    return unit(torch.cat([x1, x2], dim=1), sd, prefix + ".transition", 0, 1, 0, cfg)


def backbone(x, sd, layout, cfg, prefix="backbone", num_features=3):  # darknetv4.py:118-182
    x = unit(x, sd, prefix + ".stem", 0, 1, 1, cfg)
    feats = []
    for i, (_, nb) in enumerate(layout):
        x = csp_stage(x, sd, f"{prefix}.stages.{i}", nb, cfg)
        if i >= len(layout) - num_features:
            feats.append(x)
    return feats


def _chain(x, sd, prefix, specs, cfg, start=0):
    idx = start
    for (k, pad) in specs:
        x = unit(x, sd, prefix, idx, 1, pad, cfg)
        idx += cfg.step
    return x, idx


def pan(x, up, sd, prefix, cfg):  # yolov4.py:31-139
    out = unit(x, sd, prefix + ".conv1", 0, 1, 0, cfg)
    a = unit(up, sd, prefix + ".conv2", 0, 1, 0, cfg)
    cat = torch.cat([a, F.interpolate(out, scale_factor=2, mode="nearest")], dim=1)
    y, _ = _chain(cat, sd, prefix + ".convs", [(1, 0), (3, 1), (1, 0), (3, 1), (1, 0)], cfg)
    return y


def neck(feats, sd, cfg, prefix="neck"):  # yolov4.py:142-229
    x, idx = _chain(feats[2], sd, prefix + ".fpn", [(1, 0), (3, 1), (1, 0)], cfg)
    x = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], dim=1)   # SPP, downsample.py:154-167
    x, _ = _chain(x, sd, prefix + ".fpn", [(1, 0), (3, 1), (1, 0)], cfg, start=idx + 1)
    aux1 = pan(x, feats[1], sd, prefix + ".pan1", cfg)
    aux2 = pan(aux1, feats[0], sd, prefix + ".pan2", cfg)
    return aux2, aux1, x


def _out_conv(x, sd, key, cfg):
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    return _rnd(F.conv2d(x, _round_weight(w) if cfg.emu else w, b), cfg)


def head_logits(feats, sd, cfg, prefix="head"):  # yolov4.py:444-625
    nodrop = 2 + (1 if cfg.act is not None else 0)
    h1 = unit(feats[0], sd, prefix + ".head1", 0, 1, 1, cfg, drop=False)
    o1 = _out_conv(h1, sd, f"{prefix}.head1.{nodrop}", cfg)
    h2 = unit(feats[0], sd, prefix + ".pre_head2", 0, 2, 1, cfg)
    h2, _ = _chain(torch.cat([h2, feats[1]], dim=1), sd, prefix + ".head2_1", [(1, 0), (3, 1), (1, 0), (3, 1), (1, 0)], cfg)
    o2 = _out_conv(unit(h2, sd, prefix + ".head2_2", 0, 1, 1, cfg, drop=False), sd, f"{prefix}.head2_2.{nodrop}", cfg)
    h3 = unit(h2, sd, prefix + ".pre_head3", 0, 2, 1, cfg)
    h3, idx = _chain(torch.cat([h3, feats[2]], dim=1), sd, prefix + ".head3", [(1, 0), (3, 1), (1, 0), (3, 1), (1, 0), (3, 1)], cfg)
    o3 = _out_conv(h3, sd, f"{prefix}.head3.{idx}", cfg)
    return o1, o2, o3


def forward_logits(sd, x, layout, cfg):
    if cfg.emu:
        x = bf16r(x)
    feats = backbone(x, sd, layout, cfg)
    return head_logits(neck(feats, sd, cfg), sd, cfg)


def train_losses(sd, x, target, layout, num_classes, cfg, anchors=ANCHORS):
    """YOLOv4.forward in training mode: sum over the three scales of the four losses (yolov4.py:611-640)."""
    logits = forward_logits(sd, x, layout, cfg)
    total = None
    for o, a, s in zip(logits, anchors, SCALE_XY):
        l = oy.compute_losses(o, target, a, num_classes, s)
        total = l if total is None else {k: total[k] + l[k] for k in l}
    return total, logits


def detect(sd, x, layout, num_classes, cfg, anchors=ANCHORS):
    logits = forward_logits(sd, x, layout, cfg)
    per_scale = [oy.post_process(o, a, num_classes, s) for o, a, s in zip(logits, anchors, SCALE_XY)]
    return [{k: torch.cat([d[i][k] for d in per_scale], dim=0) for k in ("boxes", "scores", "labels")}
            for i in range(x.shape[0])]
