"""Functional CPU restatement of the reference ReXNet stack and FReLU over a plain ``state_dict``
(holocron/models/classification/rexnet.py:38-229, holocron/nn/modules/activation.py:58-82, models/utils.py:28-86).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned by tests/golden/rexnet.pt (outputs, gradients and running statistics of reference ReXBlocks / SEBlock / FReLU
and one training step of rexnet1_0x, generated from the reference itself).  ``emulate_bf16=True`` rounds to bf16 exactly
where the HIP path stores bf16 (conv outputs, unit outputs, the pooled squeeze-excite vector and gate logits, packed
dense weights), for layer-level comparison on the MI355X.
"""
import torch
import torch.nn.functional as F

from .repvgg import BN_EPS, BN_MOMENTUM, _RoundBoth, _round_weight, bf16r


def _rnd(x, emu):
    return _RoundBoth.apply(x) if emu else x


def _bn(c, y, sd, bn, training):
    """BatchNorm2d (momentum 0.1, eps 1e-5) with statistics of ``c`` applied to ``y`` (y is c except under emulation)."""
    if training:
        sd[bn + ".num_batches_tracked"] += 1
        mean, var = c.mean((0, 2, 3)), c.var((0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = c.numel() / c.shape[1]
            sd[bn + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[bn + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
    else:
        mean, var = sd[bn + ".running_mean"], sd[bn + ".running_var"]
    a = sd[bn + ".weight"] * torch.rsqrt(var + BN_EPS)
    return y * a.view(1, -1, 1, 1) + (sd[bn + ".bias"] - a * mean).view(1, -1, 1, 1)


_ACTS = {None: lambda z: z, "silu": F.silu, "relu6": F.relu6, "relu": F.relu}


def unit(x, sd, conv, bn, act, stride, pad, training, emu, groups=1, residual=None):
    """[Conv2d(bias?), BatchNorm2d, act] of conv_sequence; ``residual`` is added to the first channels afterwards."""
    w = sd[conv + ".weight"]
    b = sd.get(conv + ".bias")
    wq = _round_weight(w) if (emu and groups == 1) else w      # depthwise taps stay fp32 in the HIP kernels
    c = F.conv2d(x, wq, b, stride, pad, 1, groups)
    z = _ACTS[act](_bn(c, _rnd(c, emu) if b is None else _rnd(c - b.view(1, -1, 1, 1), emu) + b.view(1, -1, 1, 1), sd, bn, training))
    if residual is not None:
        z = torch.cat([z[:, :residual.shape[1]] + residual, z[:, residual.shape[1]:]], dim=1)   # out[:, :Cin] += x
    return _rnd(z, emu)


def se_gate(z, sd, prefix, training, emu, act="relu6"):
    """act(z * SEBlock-gate(z)) (rexnet.py:63-66 followed by the block activation, rexnet.py:129)."""
    pooled = z.flatten(2).mean(2).view(z.shape[0], -1, 1, 1)
    pooled = _rnd(pooled, emu)
    h = unit(pooled, sd, prefix + ".conv.0", prefix + ".conv.1", "relu6", 1, 0, training, emu)
    w, b = sd[prefix + ".conv.3.weight"], sd[prefix + ".conv.3.bias"]
    logits = _rnd(F.conv2d(h, _round_weight(w) if emu else w, b), emu)
    return _rnd(_ACTS[act](z * torch.sigmoid(logits)), emu)


def rex_block(x, sd, prefix, stride, use_shortcut, training, emu=False):
    """ReXBlock.forward (rexnet.py:69-143); the layout (expansion conv, squeeze-excite) is read off the state_dict."""
    p = prefix + ".conv"
    i = 0
    h = x
    if sd[f"{p}.0.weight"].shape[1] != 1 or sd[f"{p}.0.weight"].shape[2] == 1:      # dense 1x1 expansion + SiLU
        h = unit(h, sd, f"{p}.0", f"{p}.1", "silu", 1, 0, training, emu)
        i = 3
    has_se = f"{p}.{i + 2}.conv.0.weight" in sd
    cdw = sd[f"{p}.{i}.weight"].shape[0]
    h = unit(h, sd, f"{p}.{i}", f"{p}.{i + 1}", None if has_se else "relu6", stride, 1, training, emu, groups=cdw)
    i += 2
    if has_se:
        h = se_gate(h, sd, f"{p}.{i}", training, emu)
        i += 1
    i += 1                                                                            # the shared ReLU6 module
    return unit(h, sd, f"{p}.{i}", f"{p}.{i + 1}", None, 1, 0, training, emu, residual=x if use_shortcut else None)


def rexnet_strides(depth_mult=1.0):
    from math import ceil
    nb = [ceil(e * depth_mult) for e in [1, 2, 2, 3, 3, 5]]
    out = []
    for n, s in zip(nb, [1, 2, 2, 2, 1, 2]):
        out += [s] + [1] * (n - 1)
    return out


def forward(sd, x, training=False, emulate_bf16=False, depth_mult=1.0):
    """ReXNet.forward with dropout disabled: stem -> ReXBlocks -> 1x1 (SiLU) -> GAP -> Linear."""
    emu = emulate_bf16
    if emu:
        x = bf16r(x)
    h = unit(x, sd, "features.0", "features.1", "silu", 2, 1, training, emu)
    strides = rexnet_strides(depth_mult)
    for k, s in enumerate(strides):
        p = f"features.{3 + k}"
        cin = h.shape[1]
        last = max(int(key.split(".")[3]) for key in sd if key.startswith(p + ".conv.") and key.endswith(".weight") and key.count(".") == 4)
        cout = sd[f"{p}.conv.{last - 1}.weight"].shape[0]
        h = rex_block(h, sd, p, s, s == 1 and cin <= cout, training, emu)
    k = 3 + len(strides)
    h = unit(h, sd, f"features.{k}", f"features.{k + 1}", "silu", 1, 0, training, emu)
    pooled = h.flatten(2).mean(2)
    return F.linear(pooled, sd["head.1.weight"], sd["head.1.bias"])


def frelu(x, sd, training, emu=False):
    """FReLU.forward (activation.py:79-82): max(x, bn(depthwise_conv(x) + bias))."""
    c = x.shape[1]
    if emu:
        x = _rnd(x, True)
    t = unit(x, sd, "conv", "bn", None, 1, 1, training, emu, groups=c)
    return _rnd(torch.max(x, t), emu)
