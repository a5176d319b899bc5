"""YOLOv1 on the MI355X kernels (reference: holocron/models/detection/yolo.py).

``_YOLO`` keeps the reference's method contract: ``_compute_losses(pred_boxes, pred_o, pred_scores, target, ignore_high_iou)``
on the formatted predictions, ``to_isoboxes`` and ``post_process``.  The per-image / per-box Python loops of the reference
(yolo.py:84-128: a handful of tiny kernels and several host synchronisations per ground-truth box) are three launches for the
whole batch, forward and backward each (``hc_yolo1_loss_fwd`` / ``_bwd``); post-processing decodes and scores every
prediction in one launch (``hc_yolo1_decode``) and runs the greedy NMS of ``holocron_amd.ops.boxes``.
"""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ..._lib import check, ptr, stream
from ...nn.init import init_module
from ...ops.boxes import nms
from ..utils import conv_sequence

__all__ = ["YOLOv1", "yolov1"]


def _pack_targets(target: List[Dict[str, Tensor]], device):
    counts = [int(t["boxes"].shape[0]) for t in target]
    if sum(counts) > 0:
        gt_boxes = torch.cat([t["boxes"].reshape(-1, 4) for t in target], 0).to(device=device, dtype=torch.float32).contiguous()
        gt_labels = torch.cat([t["labels"].reshape(-1) for t in target], 0).to(device=device, dtype=torch.int64).contiguous()
    else:
        gt_boxes = torch.zeros((0, 4), dtype=torch.float32, device=device)
        gt_labels = torch.zeros((0,), dtype=torch.int64, device=device)
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    img = [i for i, c in enumerate(counts) for _ in range(c)]
    gt_off = torch.tensor(off, dtype=torch.int32).to(device, non_blocking=True)
    gt_img = torch.tensor(img, dtype=torch.int32).to(device, non_blocking=True)
    return gt_boxes, gt_labels, gt_img, gt_off


def _f32c(t: Tensor) -> Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _FormatFn(torch.autograd.Function):
    """``_format_outputs`` of YOLOv1 (yolo.py:314-334) / YOLOv2 (yolov2.py:175-200) as one launch each way
    (``hc_yolo_format_fwd`` / ``_bwd``): raw head output -> boxes [N, H, W, A, 4], objectness [N, H, W, A], class distribution
    [N, H, W, As, C].  ``layout`` holds the element strides of the logits (include/holocron_hip.h), ``dx_layout_of`` derives the
    same for the gradient buffer, which is ``empty_like(x)``: every logit is written exactly once."""

    @staticmethod
    def forward(ctx, x, dims, layout_of, anchors):
        import ctypes as C
        N, H, W, A, As, nc = dims
        v2 = anchors is not None
        dev = x.device
        boxes = torch.empty((N, H, W, A, 4), dtype=torch.float32, device=dev)
        obj = torch.empty((N, H, W, A), dtype=torch.float32, device=dev)
        scores = torch.empty((N, H, W, As, nc), dtype=torch.float32, device=dev)
        lay = (C.c_int64 * 8)(*layout_of(x))
        check(_lib.load().hc_yolo_format_fwd(ptr(x), lay, N, H, W, A, As, nc, int(v2), ptr(anchors) if v2 else None, ptr(boxes), ptr(obj),
                                             ptr(scores), stream()), "hc_yolo_format_fwd")
        ctx.save_for_backward(x, scores, anchors)
        ctx.meta = (dims, layout_of)
        return boxes, obj, scores

    @staticmethod
    def backward(ctx, g_boxes, g_obj, g_scores):
        import ctypes as C
        x, scores, anchors = ctx.saved_tensors
        (N, H, W, A, As, nc), layout_of = ctx.meta
        dx = torch.empty_like(x)
        gb, go, gs = (None if g is None else _f32c(g) for g in (g_boxes, g_obj, g_scores))
        check(_lib.load().hc_yolo_format_bwd(ptr(x), (C.c_int64 * 8)(*layout_of(x)), (C.c_int64 * 8)(*layout_of(dx)), N, H, W, A, As, nc,
                                             int(anchors is not None), ptr(anchors) if anchors is not None else None, ptr(scores),
                                             ptr(gb) if gb is not None else None, ptr(go) if go is not None else None,
                                             ptr(gs) if gs is not None else None, ptr(dx), stream()), "hc_yolo_format_bwd")
        return dx, None, None, None


class _Yolo1LossFn(torch.autograd.Function):
    """sums[4] = obj, noobj, bbox, clf of _YOLO._compute_losses before the lambda / N scaling."""

    @staticmethod
    def forward(ctx, pb, po, ps, packed, cell_rel, ignore):
        gt_boxes, gt_labels, gt_img, gt_off = packed
        N, H, W, A, _ = pb.shape
        As, nc = ps.shape[3], ps.shape[4]
        G = gt_boxes.shape[0]
        dev = pb.device
        pb, po, ps = _f32c(pb), _f32c(po), _f32c(ps)
        assign = torch.empty((max(2 * G, 1),), dtype=torch.int32, device=dev)
        mark = torch.empty((N, H, W, A), dtype=torch.uint8, device=dev)
        sums = torch.empty((4,), dtype=torch.float32, device=dev)
        check(_lib.load().hc_yolo1_loss_fwd(ptr(pb), ptr(po), ptr(ps), N, H, W, A, As, nc, int(cell_rel), int(ignore), ptr(gt_boxes),
                                            ptr(gt_labels), ptr(gt_img), ptr(gt_off), G, ptr(assign), ptr(mark), ptr(sums), stream()),
              "hc_yolo1_loss_fwd")
        ctx.save_for_backward(pb, po, ps, gt_boxes, gt_labels, gt_img, gt_off, assign, mark)
        ctx.meta = (N, H, W, A, As, nc, G, int(cell_rel), int(ignore))
        return sums

    @staticmethod
    def backward(ctx, gsums):
        pb, po, ps, gt_boxes, gt_labels, gt_img, gt_off, assign, mark = ctx.saved_tensors
        N, H, W, A, As, nc, G, cell_rel, ignore = ctx.meta
        gc = gsums.float().contiguous()
        dpb, dpo, dps = torch.empty_like(pb), torch.empty_like(po), torch.empty_like(ps)
        check(_lib.load().hc_yolo1_loss_bwd(ptr(pb), ptr(po), ptr(ps), N, H, W, A, As, nc, cell_rel, ignore, ptr(gt_boxes), ptr(gt_labels),
                                            ptr(gt_img), ptr(gt_off), G, ptr(assign), ptr(mark), ptr(gc), ptr(dpb), ptr(dpo), ptr(dps),
                                            stream()), "hc_yolo1_loss_bwd")
        return dpb, dpo, dps, None, None, None


class _YOLO(nn.Module):
    """Loss, box conversion and post-processing shared by YOLOv1 and YOLOv2 (yolo.py:28-215)."""

    _cell_relative = True     # YOLOv1.to_isoboxes adds the cell offset (yolo.py:153-158); YOLOv2's coordinates are absolute

    def __init__(self, num_classes: int = 20, rpn_nms_thresh: float = 0.7, box_score_thresh: float = 0.05, lambda_obj: float = 1,
                 lambda_noobj: float = 0.5, lambda_class: float = 1, lambda_coords: float = 5) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.rpn_nms_thresh = rpn_nms_thresh
        self.box_score_thresh = box_score_thresh
        self.lambda_obj = lambda_obj
        self.lambda_noobj = lambda_noobj
        self.lambda_class = lambda_class
        self.lambda_coords = lambda_coords

    def _compute_losses(self, pred_boxes: Tensor, pred_o: Tensor, pred_scores: Tensor, target: List[Dict[str, Tensor]],
                        ignore_high_iou: bool = False) -> Dict[str, Tensor]:
        """yolo.py:48-138.  pred_boxes [N, H, W, A, 4] (xc, yc, w, h), pred_o [N, H, W, A], pred_scores [N, H, W, A | 1, C]."""
        _lib.require_gpu(pred_boxes)
        gt_boxes = [t["boxes"] for t in target]
        if not all(bool(torch.all(b >= 0)) and bool(torch.all(b <= 1)) for b in gt_boxes):
            raise ValueError("Ground truth boxes are expected to have values between 0 and 1.")
        packed = _pack_targets(target, pred_boxes.device)
        sums = _Yolo1LossFn.apply(pred_boxes, pred_o, pred_scores, packed, self._cell_relative, ignore_high_iou)
        n = pred_boxes.shape[0]
        return {
            "obj_loss": (self.lambda_obj * sums[0] / n).reshape(1),
            "noobj_loss": (self.lambda_noobj * sums[1] / n).reshape(1),
            "bbox_loss": (self.lambda_coords * sums[2] / n).reshape(1),
            "clf_loss": (self.lambda_class * sums[3] / n).reshape(1),
        }

    @classmethod
    def to_isoboxes(cls, b_coords: Tensor, grid_shape: Tuple[int, int], clamp: bool = False) -> Tensor:
        """(xc, yc, w, h) -> (xmin, ymin, xmax, ymax), relative coordinates (yolo.py:140-163); b_coords [..., H, W, A, 4]."""
        _lib.require_gpu(b_coords)
        h, w = int(grid_shape[0]), int(grid_shape[1])
        if b_coords.dim() < 4 or tuple(b_coords.shape[-4:-2]) != (h, w):
            raise ValueError("to_isoboxes expects a tensor of shape (..., H, W, num_anchors, 4)")
        A = b_coords.shape[-2]
        bc = _f32c(b_coords)
        n = bc.numel() // (4 * h * w * A)
        boxes = torch.empty_like(bc)
        check(_lib.load().hc_yolo1_decode(ptr(bc), None, None, n, h, w, A, 0, int(cls._cell_relative), int(clamp), ptr(boxes), None, None,
                                          stream()), "hc_yolo1_decode")
        return boxes

    def post_process(self, b_coords: Tensor, b_o: Tensor, b_scores: Tensor, grid_shape: Tuple[int, int], rpn_nms_thresh: float = 0.7,
                     box_score_thresh: float = 0.05) -> List[Dict[str, Tensor]]:
        """yolo.py:165-215.  b_coords [N, H*W*A, 4], b_o [N, H*W*A], b_scores [N, H*W*A, C]."""
        _lib.require_gpu(b_coords)
        h, w = int(grid_shape[0]), int(grid_shape[1])
        N = b_coords.shape[0]
        A = self.num_anchors
        bc, bo, bs = _f32c(b_coords), _f32c(b_o), _f32c(b_scores)
        nc = bs.shape[-1]
        dev = bc.device
        boxes = torch.empty((N, h * w * A, 4), dtype=torch.float32, device=dev)
        score = torch.empty((N, h * w * A), dtype=torch.float32, device=dev)
        label = torch.empty((N, h * w * A), dtype=torch.int64, device=dev)
        check(_lib.load().hc_yolo1_decode(ptr(bc), ptr(bo), ptr(bs), N, h, w, A, nc, int(self._cell_relative), 1, ptr(boxes), ptr(score),
                                          ptr(label), stream()), "hc_yolo1_decode")
        keep = (bo >= 0.5) & (score >= box_score_thresh)
        detections = []
        for idx in range(N):
            sel = keep[idx].nonzero().squeeze(1)
            coords, scores, labels = boxes[idx, sel], score[idx, sel], label[idx, sel]
            if sel.numel() > 0:
                kept = nms(coords, scores, rpn_nms_thresh)
                coords, scores, labels = coords[kept], scores[kept], labels[kept]
            detections.append({"boxes": coords, "scores": scores, "labels": labels})
        return detections


class YOLOv1(_YOLO):
    """YOLOv1 (yolo.py:218-370): DarkNet-24 body, four 3x3 convolutions (one with stride 2), a two-layer classifier."""

    def __init__(self, layout: List[List[int]], num_classes: int = 20, in_channels: int = 3, stem_channels: int = 64, num_anchors: int = 2,
                 lambda_obj: float = 1, lambda_noobj: float = 0.5, lambda_class: float = 1, lambda_coords: float = 5.0,
                 rpn_nms_thresh: float = 0.7, box_score_thresh: float = 0.05, head_hidden_nodes: int = 512,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None, conv_layer: Optional[Callable[..., nn.Module]] = None,
                 backbone_norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__(num_classes, rpn_nms_thresh, box_score_thresh, lambda_obj, lambda_noobj, lambda_class, lambda_coords)
        from ..classification.darknet import DarknetBodyV1, _FusedSequential
        if act_layer is None:
            act_layer = nn.LeakyReLU(0.1, inplace=True)
        if backbone_norm_layer is None and norm_layer is not None:
            backbone_norm_layer = norm_layer
        self.backbone = DarknetBodyV1(layout, in_channels, stem_channels, act_layer, backbone_norm_layer)
        self.block4 = _FusedSequential(
            *conv_sequence(1024, 1024, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1, bias=(norm_layer is None)),
            *conv_sequence(1024, 1024, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1, stride=2,
                           bias=(norm_layer is None)),
            *conv_sequence(1024, 1024, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1, bias=(norm_layer is None)),
            *conv_sequence(1024, 1024, act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1, bias=(norm_layer is None)),
        )
        self.classifier = nn.Sequential(
            nn.Flatten(),
            nn.Linear(1024 * 7 ** 2, head_hidden_nodes),
            act_layer,
            nn.Dropout(0.5),
            nn.Linear(head_hidden_nodes, 7 ** 2 * (num_anchors * 5 + num_classes)),
        )
        self.num_anchors = num_anchors
        init_module(self.block4, "leaky_relu")
        init_module(self.classifier, "leaky_relu")

    def _format_outputs(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """[N, 7*7*(A*5 + C)] -> boxes [N, 7, 7, A, 4], objectness [N, 7, 7, A], class distribution [N, 7, 7, 1, C]
        (yolo.py:314-334): one launch (``_FormatFn``), one more for its gradient."""
        _lib.require_gpu(x)
        b, _ = x.shape
        h, w = 7, 7
        A, nc = self.num_anchors, self.num_classes
        D = A * 5 + nc
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()

        def layout_of(t):      # [N, 7 * 7 * (A * 5 + C)] contiguous: a cell's A * 5 box logits, then its C class logits
            return (h * w * D, w * D, D, 5, 1, A * 5, 0, 1)
        return _FormatFn.apply(x, (b, h, w, A, 1, nc), layout_of, None)

    def _forward(self, x: Tensor) -> Tensor:
        from ...nn.convbn_op import prepack_model_convs
        from ...nn.repblock_op import POOL
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            out = self.backbone(x)
            out = self.block4(out)
        finally:
            POOL.end()
        # the flatten order of the classifier is (C, H, W): the logical view of the NHWC bf16 activation
        return self.classifier(out.float())

    def forward(self, x: Tensor, target: Optional[List[Dict[str, Tensor]]] = None) -> Union[Dict[str, Tensor], List[Dict[str, Tensor]]]:
        if self.training and target is None:
            raise ValueError("`target` needs to be specified in training mode")
        if isinstance(x, (list, tuple)):
            x = torch.stack(x, dim=0)
        out = self._forward(x)
        b_coords, b_o, b_scores = self._format_outputs(out)
        if self.training:
            return self._compute_losses(b_coords, b_o, b_scores, target)  # type: ignore[arg-type]
        b_coords = b_coords.reshape(b_coords.shape[0], -1, 4)
        b_o = b_o.reshape(b_o.shape[0], -1)
        b_scores = b_scores.repeat_interleave(self.num_anchors, dim=3)
        b_scores = b_scores.contiguous().reshape(b_scores.shape[0], -1, self.num_classes)
        return self.post_process(b_coords, b_o, b_scores, (7, 7), self.rpn_nms_thresh, self.box_score_thresh)


def yolov1(pretrained: bool = False, progress: bool = True, pretrained_backbone: bool = True, **kwargs: Any) -> YOLOv1:
    """YOLOv1 with the DarkNet-24 layout (yolo.py:397-478).  Checkpoints need network access: ``pretrained`` raises,
    ``pretrained_backbone`` is accepted for signature parity and ignored."""
    if pretrained:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return YOLOv1([[192], [128, 256, 256, 512], [*([256, 512] * 4), 512, 1024], [512, 1024] * 2], **kwargs)
