"""Validation metrics accumulated on the device (reference: holocron/trainer/classification.py:41-70,
``ClassificationTrainer.evaluate``): the reference synchronises three times per batch (``loss.item()``, two ``.sum().item()``);
here the loss sum, the valid-batch count and the top-1 / top-5 hits stay in HBM until the end of the loader.
"""
from typing import Callable, Dict, Iterable, Tuple

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["TopKAccuracy", "evaluate_classification"]


class TopKAccuracy:
    """``update(logits, target)`` adds the batch's top-1 / top-k hits to a device counter (one launch, no synchronisation);
    ``compute()`` reads it back once.  For fewer than ``k`` classes only top-1 is counted, like the reference."""

    def __init__(self, k: int = 5) -> None:
        self.k = k
        self.counters = None

    def update(self, logits: Tensor, target: Tensor) -> None:
        _lib.require_gpu(logits, target)
        if self.counters is None or self.counters.device != logits.device:
            self.counters = torch.zeros((3,), dtype=torch.float32, device=logits.device)
        lg = logits.detach()
        if lg.dtype != torch.float32 or not lg.is_contiguous():
            lg = lg.float().contiguous()
        tg = target.detach().view(-1).long().contiguous()
        k = self.k if lg.shape[1] >= self.k else 1
        check(_lib.load().hc_topk_hits(ptr(lg), ptr(tg), lg.shape[0], lg.shape[1], k, ptr(self.counters), stream()), "hc_topk_hits")

    def compute(self) -> Tuple[float, float, int]:
        if self.counters is None:
            return 0.0, 0.0, 0
        top1, topk, n = (float(v) for v in self.counters.tolist())
        n = int(n)
        return (top1 / n if n else 0.0), (topk / n if n else 0.0), n


@torch.inference_mode()
def evaluate_classification(model: torch.nn.Module, loader: Iterable, criterion: Callable[[Tensor, Tensor], Tensor], device) -> Dict[str, float]:
    """``{"val_loss", "acc1", "acc5"}`` of ``ClassificationTrainer.evaluate`` (NaN / inf batch losses are skipped in the mean,
    trainer/classification.py:55-58) with one host synchronisation at the end."""
    model.eval()
    acc = TopKAccuracy(5)
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    valid = torch.zeros((), dtype=torch.float32, device=device)
    for x, target in loader:
        x, target = x.to(device, non_blocking=True), target.to(device, non_blocking=True)
        out = model(x)
        loss = criterion(out, target).float()
        ok = torch.isfinite(loss)
        loss_sum += torch.where(ok, loss, torch.zeros_like(loss))
        valid += ok.float()
        acc.update(out, target)
    acc1, acc5, _ = acc.compute()
    nv = float(valid)
    return {"val_loss": float(loss_sum) / nv if nv else float("nan"), "acc1": acc1, "acc5": acc5}
