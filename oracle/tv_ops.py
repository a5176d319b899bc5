"""Restatement of the torchvision.ops functions the reference calls (torchvision is absent here).

Call sites: holocron/ops/boxes.py:11,18-19,130,202; holocron/models/detection/yolov4.py:12,329,367,379.
Algorithms as published in torchvision (ops/boxes.py and csrc/ops/cpu/nms_kernel.cpp):
  box_area = (x2-x1)*(y2-y1)
  box_iou  = inter / (area1[:,None] + area2 - inter), inter = prod(clamp(min(rb)-max(lt), 0))
  nms      = greedy over a stable descending sort of the scores; box j (later in the order) is
             suppressed when inter/(area_i + area_j - inter) > iou_threshold (strict); returns the
             kept indices in score order as int64; empty input -> empty int64.
"""
import torch


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(b1, b2):
    a1, a2 = box_area(b1), box_area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (a1[:, None] + a2 - inter)


def nms(boxes, scores, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order].float()
    area = box_area(b)
    n = b.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
            yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
            xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
            yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
            w = (xx2 - xx1).clamp(min=0)
            h = (yy2 - yy1).clamp(min=0)
            inter = w * h
            ovr = inter / (area[i] + area[i + 1:] - inter)
            suppressed[i + 1:] |= ovr > iou_threshold
    return order[torch.tensor(keep, dtype=torch.int64)]
