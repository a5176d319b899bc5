"""Detection trainer (reference: holocron/trainer/detection.py).  IoU matching runs on the HIP box kernel when the boxes are on
the GPU (ops/boxes.py ``box_iou`` = hc_box_pairwise), on torch otherwise."""
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from .core import Trainer

__all__ = ["DetectionTrainer", "assign_iou"]


def _box_iou(b1: Tensor, b2: Tensor) -> Tensor:
    if b1.is_cuda and b2.is_cuda:
        from ..ops.boxes import box_iou
        return box_iou(b1, b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return inter / (a1[:, None] + a2 - inter)


def assign_iou(gt_boxes: Tensor, pred_boxes: Tensor, iou_threshold: float = 0.5) -> Tuple[List[int], List[int]]:
    """Match every ground-truth box with its best prediction (IoU >= threshold); a prediction claimed by several ground truths
    goes to the one with the highest IoU (detection.py:17-33)."""
    best = _box_iou(gt_boxes, pred_boxes).max(dim=1)
    kept = best.values >= iou_threshold
    gt_ids = torch.arange(gt_boxes.shape[0], device=kept.device)[kept]
    pred_ids, vals = best.indices[kept], best.values[kept]
    uniq = torch.unique(pred_ids)
    if pred_ids.shape[0] == uniq.shape[0]:
        return gt_ids.tolist(), pred_ids.tolist()
    gt_out, pred_out = [], []
    for p in uniq.tolist():
        mine = pred_ids == p
        winner = int(vals[mine].argmax())
        gt_out.append(int(gt_ids[mine][winner]))
        pred_out.append(p)
    return gt_out, pred_out


class DetectionTrainer(Trainer):
    """Object detection trainer (detection.py:36-126): the model returns a dict of losses in training mode and a list of
    ``{boxes, scores, labels}`` in eval mode."""

    @staticmethod
    def _to_cuda(x: List[Tensor], target: List[Dict[str, Tensor]]) -> Tuple[List[Tensor], List[Dict[str, Tensor]]]:
        x = [t.cuda(non_blocking=True) for t in x]
        target = [{k: v.cuda(non_blocking=True) for k, v in t.items()} for t in target]
        return x, target

    def _get_loss(self, x: List[Tensor], target: List[Dict[str, Tensor]]) -> Tensor:  # type: ignore[override]
        return sum(self.model(x, target).values())

    @staticmethod
    def _eval_metrics_str(eval_metrics: Dict[str, Optional[float]]) -> str:
        def pct(v):
            return f"{v:.2%}" if isinstance(v, float) else "N/A"
        return (f"Loc error: {pct(eval_metrics['loc_err'])} | Clf error: {pct(eval_metrics['clf_err'])} | "
                f"Det error: {pct(eval_metrics['det_err'])}")

    @torch.inference_mode()
    def evaluate(self, iou_threshold: float = 0.5) -> Dict[str, Optional[float]]:
        """Localisation / classification / detection error over the validation set (detection.py:79-126)."""
        self.model.eval()
        assigned = correct = missed = spurious = num_gt = 0
        for x, target in self.val_loader:
            x, target = self.to_cuda(x, target)
            for dets, t in zip(self.model(x), target):
                n_gt, n_det = t["boxes"].shape[0], dets["boxes"].shape[0]
                gt_idx: List[int] = []
                pr_idx: List[int] = []
                if n_gt > 0 and n_det > 0:
                    gt_idx, pr_idx = assign_iou(t["boxes"].float(), dets["boxes"].float(), iou_threshold)
                if gt_idx:
                    g = torch.as_tensor(gt_idx, device=t["labels"].device)
                    p = torch.as_tensor(pr_idx, device=dets["labels"].device)
                    correct += int((t["labels"][g] == dets["labels"][p].to(t["labels"].device)).sum())
                assigned += len(gt_idx)
                missed += n_gt - len(gt_idx)
                spurious += n_det - len(pr_idx)
                num_gt += n_gt
        if self._world() > 1:                # each rank scored its shard of the validation set
            dev = next(self.model.parameters()).device
            tot = torch.tensor([assigned, correct, missed, spurious, num_gt], dtype=torch.float64, device=dev)
            self._sum_over_ranks(tot)
            assigned, correct, missed, spurious, num_gt = (int(v) for v in tot.tolist())
        num_pred = num_gt - missed + spurious
        denom = num_pred + num_gt
        loc_err = 1 - 2 * assigned / denom if denom > 0 else None
        clf_err = 1 - correct / assigned if assigned > 0 else None
        det_err = 1 - 2 * correct / denom if denom > 0 else None
        return {"loc_err": loc_err, "clf_err": clf_err, "det_err": det_err, "val_loss": loc_err}
