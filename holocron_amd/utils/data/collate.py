"""Mixup on the device (reference: holocron/utils/data/collate.py:15-64).

Same constructor, one-hot conversion, ``alpha == 0`` pass-through and RNG consumption as the reference (``Beta(alpha, alpha)
.sample()`` then ``torch.randperm(batch_size)`` on the host generator), so a seeded run mixes the same pairs with the same
weight.  The reference mixes in place on the host inside the DataLoader collate; here the batch is already in HBM and the mix
is one launch for the images and one for the targets (new tensors are returned: a row is read by two outputs).
"""
from typing import Tuple

import torch
from torch import Tensor
from torch.distributions.beta import Beta

from ... import _lib
from ..._lib import check, ptr, stream

__all__ = ["Mixup"]


class Mixup(torch.nn.Module):
    def __init__(self, num_classes: int, alpha: float = 0.2) -> None:
        super().__init__()
        self.num_classes = num_classes
        if alpha < 0:
            raise ValueError("`alpha` only takes positive values")
        self.alpha = alpha

    def forward(self, inputs: Tensor, targets: Tensor) -> Tuple[Tensor, Tensor]:
        _lib.require_gpu(inputs, targets)
        lib = _lib.load()
        n = inputs.shape[0]
        index_targets = targets.ndim == 1 and self.num_classes > 1 and not targets.is_floating_point()
        if not index_targets:
            if targets.ndim == 1:                       # collate.py:41-46
                if self.num_classes > 1:
                    targets = torch.nn.functional.one_hot(targets.long(), num_classes=self.num_classes)
                elif self.num_classes == 1:
                    targets = targets.unsqueeze(1)
            targets = targets.to(dtype=inputs.dtype)
        if self.alpha == 0:
            if index_targets:
                targets = torch.nn.functional.one_hot(targets, num_classes=self.num_classes).to(dtype=inputs.dtype)
            return inputs, targets
        lam = float(Beta(self.alpha, self.alpha).sample())
        index = torch.randperm(n).to(inputs.device)
        if inputs.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.HipError("Mixup (HIP) mixes fp32 or bf16 batches")
        x = inputs.contiguous()
        out = torch.empty_like(x)
        check(lib.hc_mixup(ptr(x), ptr(index), ptr(out), n, x.numel() // max(n, 1), 0 if x.dtype == torch.float32 else 1, lam, stream()),
              "hc_mixup")
        if index_targets:
            t = targets.contiguous()
            mixed = torch.empty((n, self.num_classes), dtype=torch.float32, device=inputs.device)
            check(lib.hc_mixup_onehot(ptr(t), ptr(index), ptr(mixed), n, self.num_classes, lam, stream()), "hc_mixup_onehot")
            return out, mixed.to(dtype=inputs.dtype)
        t = targets.contiguous()
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        mixed = torch.empty_like(t)
        check(lib.hc_mixup(ptr(t), ptr(index), ptr(mixed), n, t.numel() // max(n, 1), 0 if t.dtype == torch.float32 else 1, lam, stream()),
              "hc_mixup")
        return out, mixed
