"""Box operators, restating holocron/ops/boxes.py (line numbers of the reference in comments)."""
import math

import torch

from .tv_ops import box_area, box_iou


def _box_iou(b1, b2):  # boxes.py:16-30
    a1, a2 = box_area(b1), box_area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def box_giou(b1, b2):  # boxes.py:33-66
    if torch.any(b1[:, 2:] < b1[:, :2]) or torch.any(b2[:, 2:] < b2[:, :2]):
        raise AssertionError("Incorrect coordinate format")
    iou, union = _box_iou(b1, b2)
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def iou_penalty(b1, b2):  # boxes.py:69-103 (temporaries are always fp32)
    b1, b2 = b1.float(), b2.float()
    cw = torch.max(b1[:, 2, None], b2[None, :, 2]) - torch.min(b1[:, 0, None], b2[None, :, 0])
    ch = torch.max(b1[:, 3, None], b2[None, :, 3]) - torch.min(b1[:, 1, None], b2[None, :, 1])
    c2 = cw.pow(2) + ch.pow(2)
    dx = (b1[:, 0] + b1[:, 2])[:, None] - (b2[:, 0] + b2[:, 2])[None, :]
    dy = (b1[:, 1] + b1[:, 3])[:, None] - (b2[:, 1] + b2[:, 3])[None, :]
    return (dx.pow(2) + dy.pow(2)) / 4 / c2


def diou_loss(b1, b2):  # boxes.py:106-131
    return 1 - box_iou(b1, b2) + iou_penalty(b1, b2)


def aspect_ratio(b):  # boxes.py:134-143
    return torch.atan((b[:, 2] - b[:, 0]) / (b[:, 3] - b[:, 1]))


def aspect_ratio_consistency(b1, b2):  # boxes.py:146-160
    v = aspect_ratio(b1)[:, None] - aspect_ratio(b2)[None, :]
    return v.pow(2) * (4 / math.pi ** 2)


def ciou_loss(b1, b2):  # boxes.py:163-211
    # the reference adds alpha*v into `ciou_loss[filter_]`, a boolean-mask COPY, so the term is lost
    # (boxes.py:208-209): the returned tensor equals the DIoU loss.
    return 1 - box_iou(b1, b2) + iou_penalty(b1, b2)
