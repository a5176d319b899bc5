"""Restatement of the hot-path functions of holocron/nn/functional.py."""
import torch
import torch.nn.functional as F


def hard_mish(x):  # functional.py:30-41
    return 0.5 * x * (x + 2).clamp(min=0, max=2)


def focal_loss(x, target, weight=None, ignore_index=-100, reduction="mean", gamma=2.0):  # functional.py:59-113
    K = x.shape[1]
    logp = F.log_softmax(x, dim=1)
    flat = logp.transpose(1, 0).flatten(1)                      # [K, N*S]
    tflat = target.reshape(-1)
    logpt = flat.gather(0, tflat[None, :])[0]
    valid = torch.ones_like(tflat, dtype=torch.bool)
    if 0 <= ignore_index < K:
        valid = tflat != ignore_index
    pt = logpt.exp()
    if weight is not None:
        logpt = weight.to(x.dtype).gather(0, tflat) * logpt
    loss = -1 * (1 - pt) ** gamma * logpt
    if reduction == "sum":
        return loss[valid].sum()
    if reduction == "mean":
        return loss[valid].mean()
    return loss.view(*target.shape)


def poly_loss(x, target, eps=2.0, weight=None, ignore_index=-100, reduction="mean"):  # functional.py:540-613
    K = x.shape[1]
    logp = F.log_softmax(x, dim=1)
    hard = target.ndim == x.ndim - 1
    if hard:
        if target.dtype != torch.long:
            raise TypeError("target dtype is expected to be torch.int64")
        tflat = target.reshape(-1)
        l = logp.transpose(1, 0).flatten(1).gather(0, tflat[None, :])[0]          # :571
    else:
        if target.ndim != x.ndim or target.shape[0] != x.shape[0] or target.shape[1] != x.shape[1]:
            raise ValueError("invalid target shape")
        l = logp * target                                                          # :576
    loss = -1 * l + eps * (1 - l.exp())                                            # :579
    if weight is not None:
        w = weight.to(x.dtype)
        loss = w.gather(0, tflat) * loss if hard else w.reshape(1, -1) * loss       # :586-589
    if hard:
        valid = torch.ones_like(tflat, dtype=torch.bool)
        if 0 <= ignore_index < K:
            valid = tflat != ignore_index
        if reduction == "sum":
            return loss[valid].sum()
        if reduction == "mean":
            return loss[valid].mean()
        return loss
    valid = torch.ones((K,), dtype=torch.bool)
    if 0 <= ignore_index < K:
        valid[ignore_index] = False
    if reduction == "sum":
        return loss[:, valid].sum()
    if reduction == "mean":
        return loss[:, valid].sum(1).mean()
    return loss[:, valid].sum(1)


def dice_loss(x, target, weight=None, gamma=1.0, eps=1e-8):  # functional.py:503-537
    inter = gamma * (x * target).flatten(2).sum((0, 2))
    cardinality = (x + gamma * target).flatten(2).sum((0, 2))
    dice = (inter + eps) / (cardinality + eps)
    if weight is None:
        return 1 - (1 + 1 / gamma) * dice.mean()
    w = weight.to(x.dtype)
    return 1 - (1 + 1 / gamma) * (w * dice).sum() / w.sum()


def dropblock2d(x, drop_prob, block_size, noise, training=True):  # functional.py:465-500 with the noise made explicit
    if not training or drop_prob == 0:
        return x
    gamma = drop_prob / block_size**2
    mask = (noise <= gamma).to(x.dtype)
    mask = 1 - F.max_pool2d(mask, kernel_size=(block_size, block_size), stride=(1, 1), padding=block_size // 2)
    one_count = mask.sum()
    out = x * mask.unsqueeze(1)
    if one_count > 0:
        out = out * (mask.numel() / one_count)
    return out


def slim_conv2d(x, sd, stride=1, padding=0, training=True, bn_eps=1e-5, bn_momentum=0.1):
    """SlimConv2d.forward (holocron/nn/modules/conv.py:352-370) over a state_dict (fc1, bn, fc2, conv_top, conv_bot1,
    conv_bot2); training-mode BatchNorm updates the running statistics in ``sd``."""
    z = x.mean(dim=(2, 3), keepdim=True)
    z = F.conv2d(z, sd["fc1.weight"], sd["fc1.bias"])
    if training:
        sd["bn.num_batches_tracked"] += 1
    z = F.batch_norm(z, sd["bn.running_mean"], sd["bn.running_var"], sd["bn.weight"], sd["bn.bias"], training, bn_momentum, bn_eps)
    z = F.conv2d(torch.relu(z), sd["fc2.weight"], sd["fc2.bias"])
    w = torch.sigmoid(z)
    h = x.shape[1] // 2
    xw = x * w
    top = xw[:, :h] + xw[:, h:]
    xw = x * w.flip(dims=(1,))
    bot = xw[:, :h] + xw[:, h:]
    top = F.conv2d(top, sd["conv_top.weight"], sd.get("conv_top.bias"), stride, padding)
    bot = F.conv2d(bot, sd["conv_bot1.weight"], sd["conv_bot1.bias"])
    bot = F.conv2d(bot, sd["conv_bot2.weight"], sd.get("conv_bot2.bias"), stride, padding)
    return torch.cat((top, bot), dim=1)


def norm_conv2d(x, weight, bias=None, stride=1, padding=0, eps=1e-14):
    """norm_conv2d / _xcorr2d with normalize_slices=True (holocron/nn/functional.py:322-413), without the in-place
    updates (so that it is differentiable w.r.t. x as well; the reference is not)."""
    kh, kw = weight.shape[-2:]
    h, w = x.shape[-2:]
    p = F.unfold(x, (kh, kw), dilation=1, padding=padding, stride=stride).transpose(1, 2)        # [N, L, Cin*kh*kw]
    scale = (p.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    p = (p - p.mean(-1, keepdim=True)) * scale
    out = p @ weight.view(weight.size(0), -1).t()
    if bias is not None:
        out = out + bias
    oh = (h + 2 * padding - (kh - 1) - 1) // stride + 1
    ow = (w + 2 * padding - (kw - 1) - 1) // stride + 1
    return out.transpose(1, 2).reshape(-1, weight.shape[0], oh, ow)
