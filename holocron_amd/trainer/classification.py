"""Classification trainers (reference: holocron/trainer/classification.py).  ``evaluate`` keeps the loss sum, the valid-batch
count and the top-1 / top-5 hits on the device (utils/metrics.py: one launch per batch, one synchronisation per evaluation) where
the reference synchronises three times per batch (classification.py:55-66)."""
import math
from typing import Any, Dict, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from ..utils.metrics import TopKAccuracy
from .core import Trainer

__all__ = ["BinaryClassificationTrainer", "ClassificationTrainer"]


class ClassificationTrainer(Trainer):
    """Image classification trainer (classification.py:21-78); same arguments as ``Trainer``."""

    is_binary: bool = False

    @torch.inference_mode()
    def evaluate(self) -> Dict[str, float]:
        """``{"val_loss", "acc1", "acc5"}``; NaN / inf batch losses are left out of the mean (classification.py:55-58)."""
        self.model.eval()
        dev = next(self.model.parameters()).device
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        valid = torch.zeros((), dtype=torch.float32, device=dev)
        acc = TopKAccuracy(5) if dev.type == "cuda" else None
        top1 = top5 = num_samples = seen_batches = 0
        ncls = 0
        for x, target in self.val_loader:
            x, target = self.to_cuda(x, target)
            loss, out = self._get_loss(x, target, return_logits=True)
            loss = loss.float()
            ncls = out.shape[1]
            ok = torch.isfinite(loss)
            loss_sum += torch.where(ok, loss, torch.zeros_like(loss))
            valid += ok.float()
            if acc is not None and out.is_cuda:
                acc.update(out, target)
            else:                                   # host tensors (no GPU selected): the reference's arithmetic
                pred = out.topk(5, dim=1)[1] if out.shape[1] >= 5 else out.argmax(dim=1, keepdim=True)
                correct = pred.eq(target.view(-1, 1).expand_as(pred))
                top1 += int(correct[:, 0].sum())
                top5 += int(correct.any(dim=1).sum()) if out.shape[1] >= 5 else 0
                num_samples += x.shape[0]           # (the device counters of the branch above carry their own sample count)
            seen_batches += 1
        self._sum_over_ranks(loss_sum, valid)          # every rank evaluates its own shard: the metrics are the whole set's
        nv = float(valid)
        val_loss = float(loss_sum) / nv if nv else float("nan")
        # ONE fixed-shape collective whatever this rank saw: a rank whose validation shard is empty has no device counters, and if it
        # all-reduced a different tensor (other dtype / path) than its peers the collectives would mismatch or deadlock.
        # [top-1 hits, top-5 hits, samples, classes x saw-data, saw-data]
        tot = torch.tensor([top1, top5, num_samples, float(ncls) if seen_batches else 0.0, 1.0 if seen_batches else 0.0],
                           dtype=torch.float64, device=dev)
        if acc is not None and acc.counters is not None:
            tot[:3] += acc.counters.to(torch.float64)
        self._sum_over_ranks(tot)
        top1, top5, num_samples, cls_sum, seen = (float(v) for v in tot.tolist())
        if seen and cls_sum / seen < 5:             # fewer than five classes: the reference never counts a top-5 hit (:64-65)
            top5 = 0.0
        return {"val_loss": val_loss, "acc1": top1 / max(num_samples, 1), "acc5": top5 / max(num_samples, 1)}

    @staticmethod
    def _eval_metrics_str(eval_metrics: Dict[str, float]) -> str:
        return (f"Validation loss: {eval_metrics['val_loss']:.4} "
                f"(Acc@1: {eval_metrics['acc1']:.2%}, Acc@5: {eval_metrics['acc5']:.2%})")

    @torch.inference_mode()
    def plot_top_losses(self, mean: Tuple[float, float, float], std: Tuple[float, float, float],
                        classes: Optional[Sequence[str]] = None, num_samples: int = 12, **kwargs: Any) -> None:
        """The ``num_samples`` training images with the largest loss (classification.py:80-165).  Plotting needs matplotlib."""
        if not self.is_binary and classes is None:
            raise AssertionError("arg 'classes' must be specified for multi-class classification")
        import matplotlib.pyplot as plt
        reduction = self.criterion.reduction
        self.criterion.reduction = "none"
        self.model.eval()
        best = []          # (loss, prob, pred, target, image tensor)
        try:
            for x, target in self.train_loader:
                x, target = self.to_cuda(x, target)
                batch_loss, logits = self._get_loss(x, target, return_logits=True)
                logits = logits.float()
                if self.is_binary:
                    batch_loss = batch_loss.squeeze(1)
                    probs = torch.sigmoid(logits.squeeze(1))
                    preds = (probs >= 0.5).long()
                else:
                    probs, preds = torch.softmax(logits, 1).max(dim=1)
                k = min(num_samples, batch_loss.shape[0])
                vals, idcs = batch_loss.float().topk(k)
                for v, i in zip(vals.tolist(), idcs.tolist()):
                    best.append((v, float(probs[i]), int(preds[i]), float(target[i]) if self.is_binary else int(target[i]),
                                 x[i].float().cpu()))
                best.sort(key=lambda e: -e[0])
                del best[num_samples:]
        finally:
            self.criterion.reduction = reduction
        num_cols = 4
        num_rows = max(1, math.ceil(len(best) / num_cols))
        _, axes = plt.subplots(num_rows, num_cols, figsize=(20, 5), squeeze=False)
        s, m = torch.tensor(std).view(-1, 1, 1), torch.tensor(mean).view(-1, 1, 1)
        for idx, (loss, prob, pred, tgt, img) in enumerate(best):
            ax = axes[idx // num_cols][idx % num_cols]
            ax.imshow((img * s + m).clamp(0, 1).permute(1, 2, 0).numpy())
            ax.title.set_text(f"{loss:.3} / {prob:.2} / {tgt:.2}" if self.is_binary
                              else f"{loss:.3} / {classes[pred]} ({prob:.1%}) / {classes[tgt]}")
            ax.axis("off")
        plt.show(**kwargs)


class BinaryClassificationTrainer(ClassificationTrainer):
    """Binary classification (classification.py:168-232): targets are cast to the logits' dtype and shape."""

    is_binary: bool = True

    def _get_loss(self, x: Tensor, target: Tensor, return_logits: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        out = self.model(x)
        outf = out if out.dtype == torch.float32 else out.float()
        loss = self.criterion(outf, target.to(dtype=outf.dtype).view_as(outf))
        return (loss, out) if return_logits else loss

    @torch.inference_mode()
    def evaluate(self) -> Dict[str, float]:
        """``{"val_loss", "acc"}`` (classification.py:203-228), accumulated on the device."""
        self.model.eval()
        dev = next(self.model.parameters()).device
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        valid = torch.zeros((), dtype=torch.float32, device=dev)
        hits = torch.zeros((), dtype=torch.float32, device=dev)
        num_samples = 0
        for x, target in self.val_loader:
            x, target = self.to_cuda(x, target)
            loss, out = self._get_loss(x, target, return_logits=True)
            loss = loss.float()
            ok = torch.isfinite(loss)
            loss_sum += torch.where(ok, loss, torch.zeros_like(loss))
            valid += ok.float()
            out = out.float()
            hits += ((target.view_as(out) >= 0.5) == (torch.sigmoid(out) >= 0.5)).float().sum() / out[0].numel()
            num_samples += x.shape[0]
        count = torch.full((), float(num_samples), dtype=torch.float32, device=dev)
        self._sum_over_ranks(loss_sum, valid, hits, count)
        nv = float(valid)
        return {"val_loss": float(loss_sum) / nv if nv else float("nan"), "acc": float(hits) / max(float(count), 1.0)}

    @staticmethod
    def _eval_metrics_str(eval_metrics: Dict[str, float]) -> str:
        return f"Validation loss: {eval_metrics['val_loss']:.4} (Acc: {eval_metrics['acc']:.2%})"
