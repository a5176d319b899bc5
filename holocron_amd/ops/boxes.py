"""Pairwise box operators (reference: holocron/ops/boxes.py:16-211) on the HIP kernels.

``ciou_loss`` reproduces the reference numerically, including its quirk that the aspect-ratio
term is added to a temporary copy and therefore never reaches the result.  Like the
reference's torch expressions every operator is differentiable w.r.t. both box sets
(``hc_box_pairwise_bwd``: analytic gradients with autograd's tie / clamp conventions).
"""
import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["batched_nms_sorted", "box_iou", "box_giou", "diou_loss", "ciou_loss", "iou_penalty", "aspect_ratio", "aspect_ratio_consistency",
           "nms"]

_KINDS = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3, "penalty": 4, "arc": 5}


class _PairwiseFn(torch.autograd.Function):
    """out[M, N] = op(boxes1[i], boxes2[j]) with the analytic gradient of the reference's torch expression (same sub-gradient
    conventions: tied max / min split evenly, clamp(min=0) passes the gradient at 0)."""

    @staticmethod
    def forward(ctx, boxes1, boxes2, kind):
        b1 = boxes1.detach().float().contiguous()
        b2 = boxes2.detach().float().contiguous()
        M, N = b1.shape[0], b2.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=b1.device)
        check(_lib.load().hc_box_pairwise(ptr(b1), ptr(b2), ptr(out), M, N, kind, stream()), "hc_box_pairwise")
        ctx.save_for_backward(b1, b2)
        ctx.kind = kind
        ctx.dtypes = (boxes1.dtype, boxes2.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        b1, b2 = ctx.saved_tensors
        M, N = b1.shape[0], b2.shape[0]
        db1, db2 = torch.zeros_like(b1), torch.zeros_like(b2)
        gc = g.float().contiguous()
        check(_lib.load().hc_box_pairwise_bwd(ptr(b1), ptr(b2), ptr(gc), ptr(db1), ptr(db2), M, N, ctx.kind, stream()),
              "hc_box_pairwise_bwd")
        return (db1.to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None,
                db2.to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None, None)


def _pairwise(boxes1: Tensor, boxes2: Tensor, kind: str) -> Tensor:
    _lib.require_gpu(boxes1, boxes2)
    return _PairwiseFn.apply(boxes1, boxes2, _KINDS[kind])


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """IoU matrix [M, N] (torchvision.ops.box_iou as used at holocron/ops/boxes.py:130,202)."""
    return _pairwise(boxes1, boxes2, "iou")


def box_giou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Generalized IoU (holocron/ops/boxes.py:33-66); degenerate boxes raise AssertionError."""
    if torch.any(boxes1[:, 2:] < boxes1[:, :2]) or torch.any(boxes2[:, 2:] < boxes2[:, :2]):
        raise AssertionError("Incorrect coordinate format")
    return _pairwise(boxes1, boxes2, "giou")


def iou_penalty(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Centre-distance penalty of the DIoU loss (holocron/ops/boxes.py:69-103)."""
    return _pairwise(boxes1, boxes2, "penalty")


def diou_loss(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Distance-IoU loss ``1 - IoU + penalty`` (holocron/ops/boxes.py:106-131)."""
    return _pairwise(boxes1, boxes2, "diou")


def aspect_ratio(boxes: Tensor) -> Tensor:
    """atan(w / h) (holocron/ops/boxes.py:134-143)."""
    return torch.atan((boxes[:, 2] - boxes[:, 0]) / (boxes[:, 3] - boxes[:, 1]))


def aspect_ratio_consistency(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """4/pi^2 (atan(w1/h1) - atan(w2/h2))^2 (holocron/ops/boxes.py:146-160)."""
    return _pairwise(boxes1, boxes2, "arc")


def ciou_loss(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Complete-IoU loss as the reference computes it (holocron/ops/boxes.py:163-211)."""
    return _pairwise(boxes1, boxes2, "ciou")


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS with torchvision.ops.nms semantics (call site holocron/models/detection/yolov4.py:329):
    stable descending sort of the scores, suppress j when IoU(i, j) > threshold (strict), returns
    the kept indices (int64) in score order."""
    _lib.require_gpu(boxes, scores)
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes.detach().float()[order].contiguous()
    lib = _lib.load()
    ws = torch.empty((max(int(lib.hc_nms_ws_bytes(n)), 8),), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty((n,), dtype=torch.int32, device=boxes.device)
    nkeep = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    check(lib.hc_nms_sorted(ptr(b), n, float(iou_threshold), ptr(ws), ptr(keep), ptr(nkeep), stream()), "hc_nms_sorted")
    k = int(nkeep.item())
    return order[keep[:k].long()]


def batched_nms_sorted(boxes: Tensor, off: Tensor, counts, iou_threshold: float):
    """Greedy NMS of many independent problems in one launch pair (``hc_nms_sorted_batched``).

    ``boxes`` [T, 4] fp32: the candidates of problem p are rows ``off[p] : off[p + 1]``, already in descending (stable) score
    order; ``off`` int32 device tensor [P + 1]; ``counts``: the same sizes as a host list (they size the scratch).  Returns
    ``(keep, nkeep)``: ``keep[off[p] : off[p] + nkeep[p]]`` are the kept rows of problem p, local to the problem, in score order -
    box for box the decisions of ``nms`` (torchvision.ops.nms semantics, call site holocron/models/detection/yolov4.py:329)."""
    _lib.require_gpu(boxes, off)
    P = len(counts)
    dev = boxes.device
    nkeep = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
    total = int(sum(counts))
    keep = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
    if P == 0:
        return keep[:0], nkeep[:0]
    words, ws_off = 0, []
    for n in counts:
        ws_off.append(words)
        words += int(n) * ((int(n) + 63) // 64)
    ws = torch.empty((max(words, 1),), dtype=torch.int64, device=dev)
    wo = torch.tensor(ws_off, dtype=torch.int64).to(dev, non_blocking=True)
    b = boxes.detach()
    if b.dtype != torch.float32 or not b.is_contiguous():
        b = b.float().contiguous()
    check(_lib.load().hc_nms_sorted_batched(ptr(b), ptr(off), P, int(max(counts)), float(iou_threshold), ptr(ws), ptr(wo), ptr(keep),
                                            ptr(nkeep), stream()), "hc_nms_sorted_batched")
    return keep, nkeep

