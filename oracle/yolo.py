"""CPU restatement of the YOLOv4 detection layer (holocron/models/detection/yolov4.py:269-420).  TEST INFRASTRUCTURE
ONLY (see oracle/__init__.py).  Plain torch-CPU fp32, differentiable through autograd exactly where the reference is.

Pinned by tests/golden/yolo.pt (generated from the reference itself: decode, the four losses, the gradient w.r.t. the
logits and the eval-time detections on random logits).  ``torchvision.ops.{box_iou,nms}`` are third-party and absent
from /root/reference: oracle/tv_ops.py restates them.
"""
import torch
import torch.nn.functional as F

from .boxes import ciou_loss
from .tv_ops import box_iou, nms


def format_outputs(output, anchors, num_classes, scale_xy):  # yolov4.py:269-300
    b, _, h, w = output.shape
    A = anchors.shape[0]
    output = output.reshape(b, A, 5 + num_classes, h, w).permute(0, 3, 4, 1, 2)
    c_x = torch.arange(w, dtype=torch.float32).reshape(1, 1, -1, 1)
    c_y = torch.arange(h, dtype=torch.float32).reshape(1, -1, 1, 1)
    b_xy = scale_xy * torch.sigmoid(output[..., :2]) - 0.5 * (scale_xy - 1)
    b_x = (b_xy[..., 0] + c_x) / w
    b_y = (b_xy[..., 1] + c_y) / h
    b_xy = torch.stack((b_x, b_y), dim=-1)
    b_wh = (torch.exp(output[..., 2:4]) * anchors.view(1, 1, 1, -1, 2)).clamp(0, 2)
    top_left = b_xy - 0.5 * b_wh
    bot_right = top_left + b_wh
    return torch.cat((top_left, bot_right), dim=-1), output[..., 4], output[..., 5:]


def build_targets(pred_boxes, b_o, target, anchors, num_classes):  # yolov4.py:338-392
    b, h, w, A = b_o.shape
    target_o = torch.zeros((b, h, w, A))
    target_scores = torch.zeros((b, h, w, A, num_classes))
    obj_mask = torch.zeros((b, h, w, A), dtype=torch.bool)
    noobj_mask = torch.ones((b, h, w, A), dtype=torch.bool)
    gt_boxes = [t["boxes"] for t in target]
    gt_labels = [t["labels"] for t in target]
    boxes = torch.cat(gt_boxes, dim=0)
    centers = boxes[..., [0, 2, 1, 3]].reshape(-1, 2, 2).mean(dim=-1)
    cx = (centers[:, 0] * w).to(torch.long)
    cy = (centers[:, 1] * h).to(torch.long)
    sel = torch.tensor([i for i, bx in enumerate(gt_boxes) for _ in range(bx.shape[0])], dtype=torch.long)
    if sel.shape[0] > 0:
        gt_wh = boxes[:, 2:] - boxes[:, :2]
        anchor_idxs = box_iou(torch.cat((-gt_wh, gt_wh), dim=-1), torch.cat((-anchors, anchors), dim=-1)).argmax(dim=1)
        obj_mask[sel, cy, cx, anchor_idxs] = True
        noobj_mask[sel, cy, cx, :] = False
        for idx in range(b):
            if gt_boxes[idx].shape[0] > 0:
                gt_ious, gt_idxs = box_iou(pred_boxes[idx, obj_mask[idx]], gt_boxes[idx]).max(dim=1)
                target_o[idx, obj_mask[idx]] = gt_ious                      # NOT detached in the reference (:381)
                target_scores[idx, obj_mask[idx], gt_labels[idx][gt_idxs]] = 1.0
                # yolov4.py:385-386 writes into a temporary (double advanced indexing): noobj_mask is unchanged
    return target_o, target_scores, obj_mask, noobj_mask


def compute_losses(x, target, anchors, num_classes, scale_xy, lambda_obj=1.0, lambda_noobj=0.001, lambda_class=0.1,
                   lambda_coords=1.0):  # yolov4.py:394-420
    pred_boxes, b_o, b_scores = format_outputs(x, anchors, num_classes, scale_xy)
    target_o, target_scores, obj_mask, noobj_mask = build_targets(pred_boxes, b_o, target, anchors, num_classes)
    bbox_loss = torch.zeros(1)
    for idx, t in enumerate(target):
        if t["boxes"].shape[0] > 0 and torch.any(obj_mask[idx]):
            bbox_loss = bbox_loss + ciou_loss(pred_boxes[idx, obj_mask[idx]], t["boxes"]).min(dim=1).values.sum()
    so = torch.sigmoid(b_o)
    n = so.shape[0]
    return {
        "obj_loss": lambda_obj * F.mse_loss(so[obj_mask], target_o[obj_mask], reduction="sum") / n,
        "noobj_loss": lambda_noobj * so[noobj_mask].pow(2).sum() / n,
        "bbox_loss": lambda_coords * bbox_loss / n,
        "clf_loss": lambda_class * F.binary_cross_entropy_with_logits(b_scores[obj_mask], target_scores[obj_mask],
                                                                     reduction="none").mean(1).sum(0) / n,
    }


def post_process(x, anchors, num_classes, scale_xy, rpn_nms_thresh=0.7, box_score_thresh=0.05):  # yolov4.py:302-336
    boxes, b_o, b_scores = format_outputs(x, anchors, num_classes, scale_xy)
    b_o = torch.sigmoid(b_o)
    b_scores = torch.sigmoid(b_scores)
    boxes = boxes.clamp(0, 1)
    dets = []
    for idx in range(b_o.shape[0]):
        coords = torch.zeros((0, 4))
        scores = torch.zeros(0)
        labels = torch.zeros(0, dtype=torch.long)
        m = b_o[idx] >= 0.5
        if torch.any(m):
            coords = boxes[idx, m]
            scores, labels = b_scores[idx, m].max(dim=-1)
            scores = scores * b_o[idx, m]
            k = scores >= box_score_thresh
            coords, labels, scores = coords[k].clamp(0, 1), labels[k], scores[k]
            kept = nms(coords, scores, rpn_nms_thresh)
            coords, scores, labels = coords[kept], scores[kept], labels[kept]
        dets.append({"boxes": coords, "scores": scores, "labels": labels})
    return dets
