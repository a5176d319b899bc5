"""Segmentation trainer (reference: holocron/trainer/segmentation.py).  The segmentation models are outside the hot path of this
package; the trainer itself is model-agnostic and is kept so that ``holocron.trainer`` exports the reference's names."""
from typing import Any, Dict

import torch

from .core import Trainer

__all__ = ["SegmentationTrainer"]


class SegmentationTrainer(Trainer):
    def __init__(self, *args: Any, num_classes: int = 10, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self.num_classes = num_classes

    @torch.inference_mode()
    def evaluate(self, ignore_index: int = 255) -> Dict[str, float]:
        """Validation loss, global pixel accuracy and mean IoU from a confusion matrix kept on the device (segmentation.py:37-75)."""
        self.model.eval()
        dev = next(self.model.parameters()).device
        nc = self.num_classes
        conf = torch.zeros((nc, nc), dtype=torch.int64, device=dev)
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        valid = torch.zeros((), dtype=torch.float32, device=dev)
        for x, target in self.val_loader:
            x, target = self.to_cuda(x, target)
            loss, out = self._get_loss(x, target, return_logits=True)
            loss = loss.float()
            ok = torch.isfinite(loss)
            loss_sum += torch.where(ok, loss, torch.zeros_like(loss))
            valid += ok.float()
            pred, tgt = out.argmax(dim=1).flatten(), target.flatten()
            keep = (tgt >= 0) & (tgt < nc)
            conf += torch.bincount(nc * tgt[keep].to(torch.int64) + pred[keep], minlength=nc * nc).reshape(nc, nc)
        self._sum_over_ranks(conf, loss_sum, valid)
        diag = torch.diag(conf).double()
        nv = float(valid)
        return {"val_loss": float(loss_sum) / nv if nv else float("nan"),
                "acc_global": float(diag.sum() / conf.sum()),
                "mean_iou": float((diag / (conf.sum(1) + conf.sum(0) - diag)).mean())}

    @staticmethod
    def _eval_metrics_str(eval_metrics: Dict[str, float]) -> str:
        return (f"Validation loss: {eval_metrics['val_loss']:.4} "
                f"(Acc: {eval_metrics['acc_global']:.2%} | Mean IoU: {eval_metrics['mean_iou']:.2%})")
