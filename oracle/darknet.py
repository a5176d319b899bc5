"""Functional restatement of the reference DarkNet-53 stack (holocron/models/classification/darknetv3.py,
resnet.py:59-87, models/utils.py:28-86) on torch-CPU fp32 over a plain ``state_dict``."""
import torch
import torch.nn.functional as F

from .repvgg import BN_EPS, BN_MOMENTUM, _RoundBoth, _round_weight, bf16r

DARKNET53 = [(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)]   # darknetv3.py:250


def _bn_train_or_eval(c, y, sd, prefix, training):
    """BatchNorm2d with statistics from `c` applied to `y` (c is y except under bf16 emulation)."""
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
        mean = c.mean((0, 2, 3))
        var = c.var((0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = c.numel() / c.shape[1]
            sd[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    a = sd[prefix + ".weight"] * torch.rsqrt(var + BN_EPS)
    return y * a.view(1, -1, 1, 1) + (sd[prefix + ".bias"] - a * mean).view(1, -1, 1, 1)


def conv_bn_act(x, sd, conv_key, bn_prefix, stride, pad, training, slope=0.1, emulate_bf16=False, residual=None):
    """[Conv2d(bias=False), BatchNorm2d, LeakyReLU(slope)] of conv_sequence (models/utils.py:61-84)."""
    w = sd[conv_key]
    if emulate_bf16:
        c = F.conv2d(x, _round_weight(w), None, stride, pad)
        z = _bn_train_or_eval(c, _RoundBoth.apply(c), sd, bn_prefix, training)
    else:
        c = F.conv2d(x, w, None, stride, pad)
        if training:
            sd[bn_prefix + ".num_batches_tracked"] += 1
        z = F.batch_norm(c, sd[bn_prefix + ".running_mean"], sd[bn_prefix + ".running_var"], sd[bn_prefix + ".weight"],
                         sd[bn_prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)
    out = F.leaky_relu(z, slope)
    if residual is not None:
        out = out + residual                     # _ResBlock.forward: out += identity (resnet.py:83)
    return _RoundBoth.apply(out) if emulate_bf16 else out


def res_block(x, sd, prefix, training, emulate_bf16=False):
    """ResBlock: 1x1 -> 3x3 + identity, no activation after the add (darknetv3.py:23-70)."""
    h = conv_bn_act(x, sd, prefix + ".conv.0.weight", prefix + ".conv.1", 1, 0, training, emulate_bf16=emulate_bf16)
    return conv_bn_act(h, sd, prefix + ".conv.3.weight", prefix + ".conv.4", 1, 1, training, emulate_bf16=emulate_bf16,
                       residual=x)


def forward(sd, x, layout, training=False, emulate_bf16=False):
    """DarknetV3.forward: stem -> stages (stride-2 conv + residual blocks) -> GAP -> Linear."""
    if emulate_bf16:
        x = bf16r(x)
    x = conv_bn_act(x, sd, "features.stem.0.weight", "features.stem.1", 1, 1, training, emulate_bf16=emulate_bf16)
    for si, (_, nb) in enumerate(layout):
        p = f"features.layers.{si}"
        x = conv_bn_act(x, sd, p + ".0.weight", p + ".1", 2, 1, training, emulate_bf16=emulate_bf16)
        for bi in range(nb):
            x = res_block(x, sd, f"{p}.{3 + bi}", training, emulate_bf16)
    x = x.view(x.shape[0], x.shape[1], -1).mean(2)
    return F.linear(x, sd["classifier.weight"], sd["classifier.bias"])
