"""CPU restatement of the YOLOv1 / YOLOv2 loss, box conversion and post-processing
(holocron/models/detection/yolo.py:48-215, yolov2.py:157-200).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned by tests/golden/yolo_v1.pt: the reference's own known-answer cases (tests/test_models_detection.py:95-233) and random
predictions / targets run through the reference `_YOLO._compute_losses` (losses and gradients) and `post_process`.
`box_iou` / `nms` are torchvision's (absent here): restated in oracle/tv_ops.py.
"""
import torch
import torch.nn.functional as F

from .tv_ops import box_iou, nms


def to_isoboxes(b_coords, grid_shape, clamp=False, cell_relative=True):
    """YOLOv1.to_isoboxes (yolo.py:140-163) / YOLOv2.to_isoboxes (yolov2.py:157-173)."""
    if cell_relative:
        c_x = torch.arange(grid_shape[1], dtype=torch.float)
        c_y = torch.arange(grid_shape[0], dtype=torch.float)
        b_x = (b_coords[..., 0] + c_x.reshape(1, 1, -1, 1)) / grid_shape[1]
        b_y = (b_coords[..., 1] + c_y.reshape(1, -1, 1, 1)) / grid_shape[0]
        xy = torch.stack((b_x, b_y), dim=-1)
    else:
        xy = b_coords[..., :2]
    wh = b_coords[..., 2:]
    out = torch.cat((xy - wh / 2, xy + wh / 2), dim=-1).reshape(*b_coords.shape)
    return out.clamp(0, 1) if clamp else out


def compute_losses(pred_boxes, pred_o, pred_scores, target, lambdas=(1.0, 0.5, 5.0, 1.0), ignore_high_iou=False, cell_relative=True):
    """_YOLO._compute_losses (yolo.py:48-138); lambdas = (obj, noobj, coords, class)."""
    gt_boxes = [t["boxes"] for t in target]
    gt_labels = [t["labels"] for t in target]
    if not all(torch.all(b >= 0) and torch.all(b <= 1) for b in gt_boxes):
        raise ValueError("Ground truth boxes are expected to have values between 0 and 1.")
    b, h, w, _, _ = pred_scores.shape
    pred_xyxy = to_isoboxes(pred_boxes, (h, w), False, cell_relative)
    pred_xy = (pred_xyxy[..., [0, 1]] + pred_xyxy[..., [2, 3]]) / 2
    obj_loss, noobj_loss, bbox_loss, clf_loss = (torch.zeros(1) for _ in range(4))
    is_noobj = torch.ones_like(pred_o, dtype=torch.bool)
    for idx in range(b):
        gt_xy = (gt_boxes[idx][:, :2] + gt_boxes[idx][:, 2:]) / 2
        gt_wh = gt_boxes[idx][:, 2:] - gt_boxes[idx][:, :2]
        gt_centers = torch.stack((gt_boxes[idx][:, [0, 2]].mean(dim=-1) * w, gt_boxes[idx][:, [1, 3]].mean(dim=-1) * h), dim=1)
        gt_idcs = gt_centers.to(dtype=torch.long)
        for k in range(gt_boxes[idx].shape[0]):
            cy, cx = gt_idcs[k, 1], gt_idcs[k, 0]
            iou_ = box_iou(gt_boxes[idx][k].unsqueeze(0), pred_xyxy[idx, cy, cx])
            iou, a = iou_.squeeze(0).max(dim=0)
            is_noobj[idx, cy, cx, a] = False
            gt_scores = torch.zeros_like(pred_scores[idx, cy, cx])
            gt_scores[:, gt_labels[idx][k]] = 1
            clf_loss = clf_loss + (gt_scores - pred_scores[idx, cy, cx]).pow(2).sum()
            obj_loss = obj_loss + (iou - pred_o[idx, cy, cx, a]).pow(2)
            bbox_loss = bbox_loss + (gt_xy[k] - pred_xy[idx, cy, cx, a, :2]).pow(2).sum()
            bbox_loss = bbox_loss + (gt_wh.sqrt() - pred_boxes[idx, cy, cx, a, 2:].sqrt()).pow(2).sum()   # every box of the image
        if ignore_high_iou and gt_boxes[idx].shape[0] > 0:
            iou_ = box_iou(pred_xyxy[idx].reshape(-1, 4), gt_boxes[idx]).max(dim=-1).values.reshape(h, w, -1)
            is_noobj[idx, iou_ >= 0.5] = False
    noobj_loss = noobj_loss + pred_o[is_noobj].pow(2).sum()
    n = pred_boxes.shape[0]
    return {"obj_loss": lambdas[0] * obj_loss / n, "noobj_loss": lambdas[1] * noobj_loss / n,
            "bbox_loss": lambdas[2] * bbox_loss / n, "clf_loss": lambdas[3] * clf_loss / n}


def post_process(b_coords, b_o, b_scores, grid_shape, num_anchors, rpn_nms_thresh=0.7, box_score_thresh=0.05, cell_relative=True):
    """_YOLO.post_process (yolo.py:165-215)."""
    pred_xyxy = to_isoboxes(b_coords.reshape(-1, *grid_shape, num_anchors, 4), grid_shape, True, cell_relative).reshape(b_o.shape[0], -1, 4)
    detections = []
    for idx in range(b_coords.shape[0]):
        coords = torch.zeros((0, 4), dtype=b_o.dtype)
        scores = torch.zeros(0, dtype=b_o.dtype)
        labels = torch.zeros(0, dtype=torch.long)
        obj_mask = b_o[idx] >= 0.5
        if torch.any(obj_mask):
            coords = pred_xyxy[idx, obj_mask]
            scores, labels = b_scores[idx, obj_mask].max(dim=-1)
            scores = scores * b_o[idx, obj_mask]
            keep = scores >= box_score_thresh
            coords, labels, scores = coords[keep], labels[keep], scores[keep]
            kept = nms(coords, scores, rpn_nms_thresh)
            coords, scores, labels = coords[kept], scores[kept], labels[kept]
        detections.append({"boxes": coords, "scores": scores, "labels": labels})
    return detections


def format_outputs_v1(x, num_anchors, num_classes):
    """YOLOv1._format_outputs (yolo.py:314-334)."""
    b = x.shape[0]
    x = x.reshape(b, 7, 7, num_anchors * 5 + num_classes)
    b_scores = F.softmax(x[..., -num_classes:].unsqueeze(3), dim=-1)
    x = torch.sigmoid(x[..., :num_anchors * 5].reshape(b, 7, 7, num_anchors, 5))
    return x[..., :4], x[..., 4], b_scores


def format_outputs_v2(x, anchors, num_classes):
    """YOLOv2._format_outputs (yolov2.py:175-200)."""
    b, _, h, w = x.shape
    A = anchors.shape[0]
    x = x.reshape(b, A, 5 + num_classes, h, w).permute(0, 3, 4, 1, 2)
    b_scores = F.softmax(x[..., -num_classes:], dim=-1)
    c_x = torch.arange(w, dtype=torch.float)
    c_y = torch.arange(h, dtype=torch.float)
    b_x = (torch.sigmoid(x[..., 0]) + c_x.reshape(1, 1, -1, 1)) / w
    b_y = (torch.sigmoid(x[..., 1]) + c_y.reshape(1, -1, 1, 1)) / h
    b_w = anchors[:, 0].reshape(1, 1, 1, -1) * torch.exp(x[..., 2])
    b_h = anchors[:, 1].reshape(1, 1, 1, -1) * torch.exp(x[..., 3])
    return torch.stack((b_x, b_y, b_w, b_h), dim=4), torch.sigmoid(x[..., 4]), b_scores
