"""YOLOv2 on the MI355X kernels (reference: holocron/models/detection/yolov2.py).

DarkNet-19 body with the stride-16 passthrough (1x1 conv -> ``concat_downsample2d``), two 3x3 blocks, the fused concat, a 3x3
block and the 1x1 head; losses / post-processing are ``_YOLO``'s kernels with absolute box centres (yolov2.py:157-173).
"""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import _lib
from ...nn import ConcatDownsample2d
from ...nn.convbn_op import prepack_model_convs, run_conv_sequence
from ...nn.init import init_module
from ...nn.repblock_op import POOL
from ...ops.nhwc import cat_cl
from ..classification.darknet import _FusedSequential
from ..classification.darknetv2 import DarknetBodyV2
from ..utils import conv_sequence
from .yolo import _YOLO, _FormatFn

__all__ = ["YOLOv2", "yolov2"]


class YOLOv2(_YOLO):
    _cell_relative = False

    def __init__(self, layout: List[Tuple[int, int]], num_classes: int = 20, in_channels: int = 3, stem_chanels: int = 32,
                 anchors: Optional[Tensor] = None, passthrough_ratio: int = 8, lambda_obj: float = 1, lambda_noobj: float = 0.5,
                 lambda_class: float = 1, lambda_coords: float = 5, rpn_nms_thresh: float = 0.7, box_score_thresh: float = 0.05,
                 act_layer: Optional[nn.Module] = None, norm_layer: Optional[Callable[[int], nn.Module]] = None,
                 drop_layer: Optional[Callable[..., nn.Module]] = None, conv_layer: Optional[Callable[..., nn.Module]] = None,
                 backbone_norm_layer: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__(num_classes, rpn_nms_thresh, box_score_thresh, lambda_obj, lambda_noobj, lambda_class, lambda_coords)
        if act_layer is None:
            act_layer = nn.LeakyReLU(0.1, inplace=True)
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if backbone_norm_layer is None:
            backbone_norm_layer = norm_layer
        if anchors is None:
            anchors = torch.tensor([[1.3221, 1.73145], [3.19275, 4.00944], [5.05587, 8.09892], [9.47112, 4.84053],
                                    [11.2364, 10.0071]]) / 13
        self.backbone = DarknetBodyV2(layout, in_channels, stem_chanels, True, act_layer, backbone_norm_layer, drop_layer, conv_layer)
        self.block5 = _FusedSequential(
            *conv_sequence(layout[-1][0], layout[-1][0], act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1,
                           bias=(norm_layer is None)),
            *conv_sequence(layout[-1][0], layout[-1][0], act_layer, norm_layer, drop_layer, conv_layer, kernel_size=3, padding=1,
                           bias=(norm_layer is None)),
        )
        self.passthrough_layer = _FusedSequential(
            *conv_sequence(layout[-2][0], layout[-2][0] // passthrough_ratio, act_layer, norm_layer, drop_layer, conv_layer,
                           kernel_size=1, bias=(norm_layer is None)),
            ConcatDownsample2d(scale_factor=2),
        )
        self.block6 = _FusedSequential(
            *conv_sequence(layout[-1][0] + layout[-2][0] // passthrough_ratio * 2 ** 2, layout[-1][0], act_layer, norm_layer, drop_layer,
                           conv_layer, kernel_size=3, padding=1, bias=(norm_layer is None))
        )
        self.head = nn.Conv2d(layout[-1][0], anchors.shape[0] * (5 + num_classes), 1)
        self.register_buffer("anchors", anchors)
        init_module(self.block5, "leaky_relu")
        init_module(self.passthrough_layer, "leaky_relu")
        init_module(self.block6, "leaky_relu")
        if self.head.bias is not None:
            self.head.bias.data.zero_()

    @property
    def num_anchors(self) -> int:
        return self.anchors.shape[0]

    def _format_outputs(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """[N, A*(5 + C), H, W] -> boxes [N, H, W, A, 4] (absolute xc, yc, w, h), objectness, class probabilities
        (yolov2.py:175-200): one launch (``yolo._FormatFn``: sigmoid / exp / softmax of every predictor), one more for its gradient."""
        _lib.require_gpu(x)
        b, _, h, w = x.shape
        A, nc = self.num_anchors, self.num_classes
        if x.dtype != torch.float32:
            x = x.float()

        def layout_of(t):      # [N, A * (5 + C), H, W] with whatever strides the head left (channels-last after the padded conv)
            sn, sc, sh, sw = t.stride()
            return (sn, sh, sw, (5 + nc) * sc, sc, 5 * sc, (5 + nc) * sc, sc)
        anchors = self.anchors.to(device=x.device, dtype=torch.float32).contiguous()
        return _FormatFn.apply(x, (b, h, w, A, A, nc), layout_of, anchors)

    def _forward(self, x: Tensor) -> Tensor:
        _lib.require_gpu(x)
        prepack_model_convs(self)
        POOL.begin(x.device)
        try:
            out, passthrough = self.backbone(x)
            passthrough = self.passthrough_layer(passthrough)
            out = self.block5(out)
            out = cat_cl([passthrough, out])
            out = self.block6(out)
            out = run_conv_sequence([self.head], out, padded_out=True)
        finally:
            POOL.end()
        return out[:, :self.head.out_channels].float()

    def forward(self, x: Union[Tensor, List[Tensor], Tuple[Tensor, ...]], target: Optional[List[Dict[str, Tensor]]] = None
                ) -> Union[Dict[str, Tensor], List[Dict[str, Tensor]]]:
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
        b_scores = b_scores.reshape(b_scores.shape[0], -1, self.num_classes)
        return self.post_process(b_coords, b_o, b_scores, out.shape[-2:], self.rpn_nms_thresh, self.box_score_thresh)  # type: ignore[arg-type]


def yolov2(pretrained: bool = False, progress: bool = True, pretrained_backbone: bool = True, **kwargs: Any) -> YOLOv2:
    """YOLOv2 with the DarkNet-19 layout (yolov2.py:272-321)."""
    if pretrained:
        raise RuntimeError("pretrained checkpoints need network access; use load_state_dict with a reference state_dict")
    return YOLOv2([(64, 0), (128, 1), (256, 1), (512, 2), (1024, 2)], **kwargs)
