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
