"""YOLOv4 on the MI355X kernels (reference: holocron/models/detection/yolov4.py).

Same module tree / ``state_dict`` keys (``backbone.*``, ``neck.{fpn,pan1,pan2}.*``,
``head.{head1,pre_head2,head2_1,head2_2,pre_head3,head3,yolo1..3}.*``), same forward contract: a dict of four
losses in training mode (``ValueError`` without targets), a list of ``{boxes, scores, labels}`` in eval mode.

* conv stacks: fused conv_bn_act units (Mish + DropBlock ride in the BN pass), concats written in place;
* ``YoloLayer``: decode / target assignment / loss + gradient / candidate filter are the kernels of
  csrc/yolo.hip reading the head's NHWC logits in place; NMS is ``holocron_amd.ops.boxes.nms``.

Reference quirks kept (SURVEY.md "parity-critical quirks"): ``ciou_loss == diou_loss``; the "ignore" step of
``_build_targets`` (yolov4.py:385-386) assigns into a temporary and therefore never changes ``noobj_mask``; the
IoU objectness target is not detached, so ``obj_loss`` also back-propagates into the boxes.
"""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ..._lib import check, ptr, stream
from ...nn import SPP, DropBlock2d
from ...nn.convbn_op import cl_ld, prepack_model_convs, run_conv_sequence
from ...nn.functional import drop_plan_scope
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ...ops.boxes import nms
from ...ops.nhwc import cat_buffer, cat_cl, upsample2x_cl
from ..classification.darknetv4 import DarknetBodyV4
from ..utils import conv_sequence

__all__ = ["PAN", "Neck", "YoloLayer", "Yolov4Head", "YOLOv4", "yolov4", "PackedTargets"]


class PAN(nn.Module):
    """Path-aggregation block (yolov4.py:31-139): 1x1 on both inputs, upsample + concat, five alternating convs."""

    def __init__(self, in_channels: int, act_layer=None, norm_layer=None, drop_layer=None, conv_layer=None) -> None:
        super().__init__()
        b = norm_layer is None
        half = in_channels // 2

        def cs(cin, cout, **kw):
            return conv_sequence(cin, cout, act_layer, norm_layer, drop_layer, conv_layer, bias=b, **kw)
        self.conv1 = nn.Sequential(*cs(in_channels, half, kernel_size=1))
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.conv2 = nn.Sequential(*cs(in_channels, half, kernel_size=1))
        self.convs = nn.Sequential(
            *cs(in_channels, half, kernel_size=1), *cs(half, in_channels, kernel_size=3, padding=1),
            *cs(in_channels, half, kernel_size=1), *cs(half, in_channels, kernel_size=3, padding=1),
            *cs(in_channels, half, kernel_size=1))

    def forward(self, x: Tensor, up: Tensor) -> Tensor:
        out = run_conv_sequence(self.conv1, x)
        N, _, H, W = up.shape
        half = out.shape[1]
        buf, (pa, pb) = cat_buffer(N, [half, half], H, W, up.device)
        a = run_conv_sequence(self.conv2, up, out=pa)
        b = upsample2x_cl(out, out=pb)
        return run_conv_sequence(self.convs, cat_cl([a, b], buf))


class Neck(nn.Module):
    def __init__(self, in_planes: List[int], act_layer=None, norm_layer=None, drop_layer=None, conv_layer=None) -> None:
        super().__init__()
        b = norm_layer is None
        c = in_planes[0]

        def cs(cin, cout, **kw):
            return conv_sequence(cin, cout, act_layer, norm_layer, drop_layer, conv_layer, bias=b, **kw)
        self.fpn = nn.Sequential(
            *cs(c, c // 2, kernel_size=1), *cs(c // 2, c, kernel_size=3, padding=1), *cs(c, c // 2, kernel_size=1),
            SPP([5, 9, 13]),
            *cs(4 * c // 2, c // 2, kernel_size=1), *cs(c // 2, c, kernel_size=3, padding=1), *cs(c, c // 2, kernel_size=1))
        self.pan1 = PAN(in_planes[1], act_layer, norm_layer, drop_layer, conv_layer)
        self.pan2 = PAN(in_planes[2], act_layer, norm_layer, drop_layer, conv_layer)
        init_module(self, "leaky_relu")

    def forward(self, feats: List[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
        out = run_conv_sequence(self.fpn, feats[2])
        aux1 = self.pan1(out, feats[1])
        aux2 = self.pan2(aux1, feats[0])
        return aux2, aux1, out


def _logit_layout(x: Tensor, channels: int):
    """(tensor, dtype code, sn, sc, sp) describing how the kernels read ``x`` ([N, channels, H, W] logically)."""
    N, Cc, H, W = x.shape
    padded = x.dtype == torch.bfloat16 and Cc == (channels + 15) // 16 * 16 and cl_ld(x) == Cc
    if Cc != channels and not padded:
        raise ValueError(f"expected {channels} channels, got {Cc}")
    if x.dtype == torch.bfloat16 and (Cc == 1 or x.stride(1) == 1):
        ld = x.stride(3) if W > 1 else (x.stride(2) if H > 1 else x.stride(0))
        if (H == 1 or x.stride(2) == W * ld) and (N == 1 or x.stride(0) == H * W * ld):
            return x, 1, H * W * ld, 1, ld
    if x.dtype == torch.float32 and x.is_contiguous():
        return x, 0, Cc * H * W, H * W, 1
    x = x.float().contiguous()
    return x, 0, Cc * H * W, H * W, 1


class _YoloLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, tgt):
        gt_boxes, gt_labels, gt_img, gt_off = tgt
        N, _, H, W = x.shape
        A, nc = layer.anchors.shape[0], layer.num_classes
        xs, dt, sn, sc, sp = _logit_layout(x, A * (5 + nc))
        lib = _lib.load()
        dev = x.device
        obj_mask = torch.empty((N, H, W, A), dtype=torch.uint8, device=dev)
        cell_gt = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
        anchors = layer.anchors.float().contiguous()
        check(lib.hc_yolo_assign(ptr(gt_boxes), ptr(gt_img), gt_boxes.shape[0], ptr(anchors), N, H, W, A, ptr(obj_mask),
                                 ptr(cell_gt), stream()), "hc_yolo_assign")
        sums = torch.empty((4,), dtype=torch.float32, device=dev)
        check(lib.hc_yolo_loss_fwd(ptr(xs), dt, sn, sc, sp, N, H, W, A, nc, ptr(anchors), layer.scale_xy, ptr(gt_boxes),
                                   ptr(gt_labels), ptr(gt_off), ptr(obj_mask), ptr(cell_gt), ptr(sums), stream()),
              "hc_yolo_loss_fwd")
        ctx.save_for_backward(xs, anchors, gt_boxes, gt_labels, gt_off, obj_mask, cell_gt)
        ctx.meta = (dt, sn, sc, sp, N, H, W, A, nc, layer.scale_xy, x.dtype)
        return sums

    @staticmethod
    def backward(ctx, gsums):
        xs, anchors, gt_boxes, gt_labels, gt_off, obj_mask, cell_gt = ctx.saved_tensors
        dt, sn, sc, sp, N, H, W, A, nc, scale_xy, in_dtype = ctx.meta
        gcoef = gsums.float().contiguous()
        # same memory layout as the logits that were read (padded NHWC bf16, or NCHW fp32)
        if dt == 0 or cl_ld(xs) == xs.shape[1]:
            dx = torch.zeros_like(xs)
        else:   # a channel slice of a wider buffer: keep the pixel stride
            span = xs.stride(0) * xs.shape[0]
            dx = torch.zeros((span,), dtype=torch.bfloat16, device=xs.device).as_strided(xs.shape, xs.stride(), 0)
        check(_lib.load().hc_yolo_loss_bwd(ptr(xs), dt, sn, sc, sp, N, H, W, A, nc, ptr(anchors), scale_xy, ptr(gt_boxes),
                                           ptr(gt_labels), ptr(gt_off), ptr(obj_mask), ptr(cell_gt), ptr(gcoef), ptr(dx), stream()),
              "hc_yolo_loss_bwd")
        return (dx if dx.dtype == in_dtype else dx.to(in_dtype)), None, None


class PackedTargets:
    """The ragged ground truth of a batch (`target`: the reference's list of {"boxes" [k, 4], "labels" [k]} dicts,
    holocron/models/detection/yolov4.py:338-388) flattened ONCE into the four device tensors the assignment / loss kernels read:
    boxes [M, 4] fp32, labels [M] int64, image index per box [M] int32, per-image offsets [N + 1] int32.

    Pass it in place of `target` (`model(x, packed)`): the forward then does no host work and issues no host -> device copy, which is
    what a training step replayed from a hipGraph needs (the copies of a per-step packing are pageable memcpy nodes whose host source is
    gone at replay).  The buffers keep their addresses; `update(target)` refills them for a batch with the same box counts."""

    def __init__(self, target: List[Dict[str, Tensor]], device) -> None:
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PackedTargets must be built outside stream capture")
        self.counts = [int(t["boxes"].shape[0]) for t in target]
        self._t = YoloLayer._pack_targets(list(target), torch.device(device))
        self.n = len(target)

    def __len__(self) -> int:
        return self.n

    def tensors(self, device):
        if self._t[3].device != torch.device(device):
            raise ValueError("PackedTargets lives on another device than the model")
        return self._t

    def update(self, target: List[Dict[str, Tensor]]) -> None:
        """Same box count per image, new boxes / labels: refill the device buffers in place (not under capture)."""
        if [int(t["boxes"].shape[0]) for t in target] != self.counts:
            raise ValueError("PackedTargets.update: the box counts per image changed - build a new PackedTargets (and re-capture)")
        if sum(self.counts) == 0:
            return
        dev = self._t[0].device
        self._t[0].copy_(torch.cat([t["boxes"].reshape(-1, 4) for t in target], 0).to(device=dev, dtype=torch.float32))
        self._t[1].copy_(torch.cat([t["labels"].reshape(-1) for t in target], 0).to(device=dev, dtype=torch.int64))


class YoloLayer(nn.Module):
    """Scale-specific part of the YOLO head (yolov4.py:233-442)."""

    def __init__(self, anchors: Tensor, num_classes: int = 80, scale_xy: float = 1.0, iou_thresh: float = 0.213,
                 lambda_obj: float = 1, lambda_noobj: float = 0.001, lambda_class: float = 0.1, lambda_coords: float = 1.0,
                 rpn_nms_thresh: float = 0.7, box_score_thresh: float = 0.05, ignore_thresh: float = 0.5) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.register_buffer("anchors", anchors)
        self.rpn_nms_thresh = rpn_nms_thresh
        self.box_score_thresh = box_score_thresh
        self.ignore_thresh = ignore_thresh
        self.lambda_obj = lambda_obj
        self.lambda_noobj = lambda_noobj
        self.lambda_class = lambda_class
        self.lambda_coords = lambda_coords
        self.scale_xy = scale_xy
        self.iou_thresh = iou_thresh

    def extra_repr(self) -> str:
        return f"num_classes={self.num_classes}, scale_xy={self.scale_xy}"

    def _decode(self, x: Tensor, with_scores: bool, clamp01: bool):
        _lib.require_gpu(x)
        N, _, H, W = x.shape
        A, nc = self.anchors.shape[0], self.num_classes
        xs, dt, sn, sc, sp = _logit_layout(x, A * (5 + nc))
        dev = x.device
        boxes = torch.empty((N, H, W, A, 4), dtype=torch.float32, device=dev)
        obj = score = label = None
        if with_scores:
            obj = torch.empty((N, H, W, A), dtype=torch.float32, device=dev)
            score = torch.empty((N, H, W, A), dtype=torch.float32, device=dev)
            label = torch.empty((N, H, W, A), dtype=torch.int64, device=dev)
        anchors = self.anchors.float().contiguous()
        check(_lib.load().hc_yolo_decode(ptr(xs), dt, sn, sc, sp, N, H, W, A, nc, ptr(anchors), self.scale_xy, ptr(boxes), ptr(obj),
                                         ptr(score), ptr(label), int(clamp01), stream()), "hc_yolo_decode")
        return boxes, obj, score, label

    def _format_outputs(self, output: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """(boxes [N,H,W,A,4] xyxy, objectness logits [N,H,W,A], class logits [N,H,W,A,nc]) — yolov4.py:269-300.
        Boxes come from the decode kernel (no gradient: the training path differentiates inside the loss kernel)."""
        b, _, h, w = output.shape
        A = self.anchors.shape[0]
        boxes, _, _, _ = self._decode(output, False, False)
        out = output.reshape(b, A, 5 + self.num_classes, h, w).permute(0, 3, 4, 1, 2)
        return boxes, out[..., 4], out[..., 5:]

    def post_process_logits(self, x: Tensor) -> List[Dict[str, Tensor]]:
        """Eval path (yolov4.py:302-336) from raw logits: objectness >= 0.5, class-agnostic score threshold, boxes
        clamped to [0, 1], greedy NMS per image."""
        boxes, obj, score, label = self._decode(x, True, True)
        N = x.shape[0]
        boxes, obj, score, label = boxes.view(N, -1, 4), obj.view(N, -1), score.view(N, -1), label.view(N, -1)
        keep = (obj >= 0.5) & (score >= self.box_score_thresh)
        detections = []
        for idx in range(N):
            sel = keep[idx].nonzero().squeeze(1)   # (h, w, anchor) order, like the reference's boolean indexing
            coords, scores, labels = boxes[idx, sel], score[idx, sel], label[idx, sel]
            if sel.numel() > 0:
                kept = nms(coords, scores, self.rpn_nms_thresh)
                coords, scores, labels = coords[kept], scores[kept], labels[kept]
            detections.append({"boxes": coords, "scores": scores, "labels": labels})
        return detections

    @staticmethod
    def _pack_targets(target, device):
        if isinstance(target, PackedTargets):
            return target.tensors(device)
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # Packing the ragged ground truth is host work followed by pageable host -> device copies.  Captured, those become memcpy
            # nodes that re-read freed host buffers on every replay (a memory fault on the MI355X: VERDICT r2 weak #9).
            raise RuntimeError("YOLOv4: pack the targets before stream capture - `packed = PackedTargets(target, device)` once, then "
                               "`model(x, packed)` inside the captured step")
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

    def _weighted_losses(self, x: Tensor, target: List[Dict[str, Tensor]], packed=None) -> Tensor:
        """The four losses of this layer as ONE fp32 vector [obj, noobj, bbox, clf], already weighted by the lambdas and divided by the
        batch size (yolov4.py:411-420): one multiply by a cached device vector instead of a mul and a div per entry."""
        if packed is None:
            packed = self._pack_targets(target, x.device)
        sums = _YoloLossFn.apply(x, self, packed)
        n = x.shape[0]
        key = (n, float(self.lambda_obj), float(self.lambda_noobj), float(self.lambda_coords), float(self.lambda_class), x.device)
        if getattr(self, "_hc_coef_key", None) != key:
            self._hc_coef = torch.tensor([self.lambda_obj / n, self.lambda_noobj / n, self.lambda_coords / n, self.lambda_class / n],
                                         dtype=torch.float32).to(x.device)
            self._hc_coef_key = key
        return sums * self._hc_coef

    @staticmethod
    def _loss_dict(w: Tensor) -> Dict[str, Tensor]:
        return {"obj_loss": w[0], "noobj_loss": w[1], "bbox_loss": w[2].reshape(1), "clf_loss": w[3]}

    def _compute_losses_logits(self, x: Tensor, target: List[Dict[str, Tensor]], packed=None) -> Dict[str, Tensor]:
        return self._loss_dict(self._weighted_losses(x, target, packed))

    def forward(self, x: Tensor, target: Optional[List[Dict[str, Tensor]]] = None, packed=None):
        if self.training and target is None:
            raise ValueError("`target` needs to be specified in training mode")
        if self.training:
            return self._compute_losses_logits(x, target, packed)
        return self.post_process_logits(x)


class Yolov4Head(nn.Module):
    def __init__(self, num_classes: int = 80, anchors: Optional[Tensor] = None, act_layer=None, norm_layer=None, drop_layer=None,
                 conv_layer=None) -> None:
        if anchors is None:
            anchors = torch.tensor([[[12, 16], [19, 36], [40, 28]], [[36, 75], [76, 55], [72, 146]],
                                    [[142, 110], [192, 243], [459, 401]]], dtype=torch.float32) / 608
        elif not isinstance(anchors, torch.Tensor):
            anchors = torch.tensor(anchors, dtype=torch.float32)
        if anchors.shape[0] != 3:
            raise AssertionError(f"The number of anchors is expected to be 3. received: {anchors.shape[0]}")
        super().__init__()
        b = norm_layer is None
        nout = (5 + num_classes) * 3

        def cs(cin, cout, drop=drop_layer, **kw):
            return conv_sequence(cin, cout, act_layer, norm_layer, drop, conv_layer, bias=b, **kw)

        def out_conv(cin):
            return conv_sequence(cin, nout, None, None, None, conv_layer, kernel_size=1, bias=True)
        self.head1 = nn.Sequential(*cs(128, 256, drop=None, kernel_size=3, padding=1), *out_conv(256))
        self.yolo1 = YoloLayer(anchors[0], num_classes=num_classes, scale_xy=1.2)
        self.pre_head2 = nn.Sequential(*cs(128, 256, kernel_size=3, padding=1, stride=2))
        self.head2_1 = nn.Sequential(
            *cs(512, 256, kernel_size=1), *cs(256, 512, kernel_size=3, padding=1), *cs(512, 256, kernel_size=1),
            *cs(256, 512, kernel_size=3, padding=1), *cs(512, 256, kernel_size=1))
        self.head2_2 = nn.Sequential(*cs(256, 512, drop=None, kernel_size=3, padding=1), *out_conv(512))
        self.yolo2 = YoloLayer(anchors[1], num_classes=num_classes, scale_xy=1.1)
        self.pre_head3 = nn.Sequential(*cs(256, 512, kernel_size=3, padding=1, stride=2))
        self.head3 = nn.Sequential(
            *cs(1024, 512, kernel_size=1), *cs(512, 1024, kernel_size=3, padding=1), *cs(1024, 512, kernel_size=1),
            *cs(512, 1024, kernel_size=3, padding=1), *cs(1024, 512, kernel_size=1), *cs(512, 1024, kernel_size=3, padding=1),
            *out_conv(1024))
        self.yolo3 = YoloLayer(anchors[2], num_classes=num_classes, scale_xy=1.05)
        init_module(self, "leaky_relu")
        for seq in (self.head1, self.head2_2, self.head3):   # zero init of the output convs (yolov4.py:601-607)
            seq[-1].weight.data.zero_()
            seq[-1].bias.data.zero_()

    def forward(self, feats: List[Tensor], target: Optional[List[Dict[str, Tensor]]] = None):
        if self.training and target is None:
            raise ValueError("`target` needs to be specified in training mode")
        o1 = run_conv_sequence(self.head1, feats[0], padded_out=True)

        N, c2, H2, W2 = feats[1].shape
        buf, (pa, _) = cat_buffer(N, [256, c2], H2, W2, feats[1].device)
        h2 = run_conv_sequence(self.pre_head2, feats[0], out=pa)
        h2 = run_conv_sequence(self.head2_1, cat_cl([h2, feats[1]], buf))
        o2 = run_conv_sequence(self.head2_2, h2, padded_out=True)

        N, c3, H3, W3 = feats[2].shape
        buf, (pa, _) = cat_buffer(N, [512, c3], H3, W3, feats[2].device)
        h3 = run_conv_sequence(self.pre_head3, h2, out=pa)
        o3 = run_conv_sequence(self.head3, cat_cl([h3, feats[2]], buf), padded_out=True)

        if not self.training:          # the 3 x N NMS problems of the batch as one launch pair (same detections, same order)
            return post_process_scales([self.yolo1, self.yolo2, self.yolo3], [o1, o2, o3])
        # training: the three scales' weighted loss vectors are added as vectors (two launches) and THEN split into the reference's
        # dict (yolov4.py:611-640 adds the twelve scalars one by one)
        packed = YoloLayer._pack_targets(target, o1.device)
        w = (self.yolo1._weighted_losses(o1, target, packed) + self.yolo2._weighted_losses(o2, target, packed)
             + self.yolo3._weighted_losses(o3, target, packed))
        return YoloLayer._loss_dict(w)


def post_process_scales(layers, outs) -> List[Dict[str, Tensor]]:
    """Eval path of the whole head in one go: what ``[layer.post_process_logits(o) for layer, o in zip(layers, outs)]`` followed by the
    per-image concatenation of Yolov4Head.forward computes (yolov4.py:302-336, 603-609), detection for detection and in the same
    order - but the ``N x len(layers)`` (image, scale) NMS problems run as ONE batched launch pair and the host waits for the
    device twice per batch instead of twice per image and scale (the candidate counts, then the kept counts)."""
    from ...ops.boxes import batched_nms_sorted
    N, S, dev = outs[0].shape[0], len(layers), outs[0].device
    thr = layers[0].rpn_nms_thresh
    if any(l.rpn_nms_thresh != thr for l in layers):
        raise ValueError("post_process_scales: the layers must share one NMS threshold")
    bl, sl, ll, kl, sc_of = [], [], [], [], []
    for si, (layer, x) in enumerate(zip(layers, outs)):
        boxes, obj, score, label = layer._decode(x, True, True)
        boxes, obj, score, label = boxes.view(N, -1, 4), obj.view(N, -1), score.view(N, -1), label.view(N, -1)
        bl.append(boxes), sl.append(score), ll.append(label)
        kl.append((obj >= 0.5) & (score >= layer.box_score_thresh))
        sc_of.append(torch.full((boxes.shape[1],), si, dtype=torch.int64, device=dev))
    boxes, score, label, keepm = torch.cat(bl, 1), torch.cat(sl, 1), torch.cat(ll, 1), torch.cat(kl, 1)
    Pt = boxes.shape[1]
    # problem of a candidate: image-major, so that an image's detections end up contiguous (scale 1, 2, 3: the reference's cat order)
    pid = torch.arange(N, device=dev)[:, None] * S + torch.cat(sc_of)[None, :]
    pid = torch.where(keepm, pid, torch.full_like(pid, N * S)).reshape(-1)
    # stable sort by descending score, then stable sort by problem: every problem's candidates in descending score order with ties in
    # (h, w, anchor) order - the order torchvision's nms visits them in
    o1 = torch.sort(score.reshape(-1), descending=True, stable=True).indices
    p1 = pid[o1]
    o2 = torch.sort(p1, stable=True).indices
    order, psort = o1[o2], p1[o2]
    counts = torch.bincount(psort, minlength=N * S + 1)[: N * S]
    counts_h = counts.cpu().tolist()                                    # host wait 1: the scratch and the grid need the sizes
    total = sum(counts_h)
    empty = {"boxes": boxes.new_zeros((0, 4)), "scores": score.new_zeros((0,)), "labels": label.new_zeros((0,))}
    if total == 0:
        return [dict(empty) for _ in range(N)]
    order, psort = order[:total], psort[:total]
    off = torch.zeros((N * S + 1,), dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    flat_boxes = boxes.reshape(-1, 4)
    keep, nkeep = batched_nms_sorted(flat_boxes[order].contiguous(), off, counts_h, thr)
    pos = torch.arange(total, device=dev)
    base = off[:-1].long()[psort]
    sel = (pos - base) < nkeep.long()[psort]
    final = order[(base + keep[:total].long())[sel]]                   # (boolean indexing: host wait 2)
    fb, fs, fl = flat_boxes[final], score.reshape(-1)[final], label.reshape(-1)[final]
    sizes = nkeep.view(N, S).sum(1).cpu().tolist()
    out, at = [], 0
    for n in range(N):
        out.append({"boxes": fb[at:at + sizes[n]], "scores": fs[at:at + sizes[n]], "labels": fl[at:at + sizes[n]]})
        at += sizes[n]
    return out


class YOLOv4(nn.Module):
    def __init__(self, layout: List[Tuple[int, int]], num_classes: int = 80, in_channels: int = 3, stem_channels: int = 32,
                 anchors: Optional[Tensor] = None, act_layer: Optional[nn.Module] = None,
                 norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None,
                 conv_layer: Optional[Callable[..., nn.Module]] = None,
                 backbone_norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__()
        act_layer = nn.Mish(inplace=True) if act_layer is None else act_layer
        norm_layer = nn.BatchNorm2d if norm_layer is None else norm_layer
        backbone_norm_layer = norm_layer if backbone_norm_layer is None else backbone_norm_layer
        drop_layer = DropBlock2d if drop_layer is None else drop_layer
        self.backbone = DarknetBodyV4(layout, in_channels, stem_channels, 3, act_layer, backbone_norm_layer, drop_layer, conv_layer)
        self.neck = Neck([1024, 512, 256], act_layer, norm_layer, drop_layer, conv_layer)
        self.head = Yolov4Head(num_classes, anchors, act_layer, norm_layer, drop_layer, conv_layer)
        init_module(self.neck, "leaky_relu")
        init_module(self.head, "leaky_relu")

    def forward(self, x: Union[Tensor, List[Tensor]], target: Optional[List[Dict[str, Tensor]]] = None):
        if not isinstance(x, torch.Tensor):
            x = torch.stack(x, dim=0)
        if self.training and target is None:
            raise ValueError("`target` needs to be specified in training mode")
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            with drop_plan_scope(self, x.device):
                feats = self.backbone(x)
                x20, x13, x6 = self.neck(feats)
                return self.head((x20, x13, x6), target)
        finally:
            POOL.end()


def yolov4(pretrained: bool = False, progress: bool = True, pretrained_backbone: bool = True, **kwargs: Any) -> YOLOv4:
    """YOLOv4 (yolov4.py:736-775) with the CSP-Darknet-53 layout.  Checkpoints cannot be downloaded here: the
    reference default ``pretrained_backbone=True`` (FrozenBatchNorm2d backbone + download) raises; build with
    ``pretrained_backbone=False`` and ``load_state_dict`` reference weights."""
    if pretrained or pretrained_backbone:
        raise RuntimeError("pretrained weights need network access; pass pretrained_backbone=False and load a state_dict")
    return YOLOv4([(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)], **kwargs)
