"""fp8 (OCP e4m3) inference executor for a re-parametrised RepVGG — BASELINE.json config C5
("repvgg_a2 fp8 MFMA inference (reparametrised 3x3), synthetic 224^2, bs=1024").

The reference has no fp8 path; its inference form is ``RepVGG.reparametrize()`` (holocron/models/classification/
repvgg.py:75-107,168-171): one 3x3 conv + bias + ReLU per block.  This executor runs exactly that graph with

* weights quantised per output channel:  Wq = fp8(W / sw[co]),  sw[co] = max|W[co]| / 448,
* activations quantised per tensor with static scales from a calibration batch:  Xq = fp8(X / sx),
* convolutions on ``v_mfma_scale_f32_32x32x64_f8f6f4`` (unit block scales) with fp32 accumulation, and the epilogue
  ``out_q = fp8(relu(acc * (sw[co] * sx_in / sx_out) + bias[co] / sx_out))`` (hc_conv_gather, ``ch_mult`` mode),
* channel counts padded to multiples of 64 in the fp8 layout (zeros), the stem (3 input channels) through an im2col
  that quantises straight to a 64-wide fp8 k-step,
* global average pool on the fp8 tensor, fp32 linear head.
"""
from typing import List

import torch
from torch import nn

from ... import _lib
from ..._lib import check, ptr, stream
from ...ops import conv as cv
from .repvgg import RepBlock, RepVGG

FP8_MAX = 448.0


def _ceil64(c):
    return (c + 63) // 64 * 64


def quantize_weight_fp8(w: torch.Tensor, cin_pad: int, cout_pad: int):
    """fp32 OIHW -> (uint8 e4m3 [cout_pad][KH*KW][cin_pad], per-output-channel scale fp32 [cout_pad])."""
    Cout, Cin, KH, KW = w.shape
    amax = w.detach().abs().amax(dim=(1, 2, 3)).clamp(min=1e-12)
    sw = (amax / FP8_MAX).float()
    q = (w.detach().float() / sw.view(-1, 1, 1, 1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    packed = torch.zeros((cout_pad, KH * KW, cin_pad), dtype=torch.uint8, device=w.device)
    packed[:Cout, :, :Cin] = q.permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin).contiguous().view(torch.uint8)
    swp = torch.ones((cout_pad,), dtype=torch.float32, device=w.device)
    swp[:Cout] = sw
    return packed, swp


class _Layer:
    __slots__ = ("wq", "mult", "badd", "stride", "cin_p", "cout", "cout_p", "sx_out", "desc", "im2col_k")


class Fp8RepVGG(nn.Module):
    """Build from a re-parametrised, eval-mode RepVGG on the GPU and a calibration batch (logical NCHW fp32)."""

    def __init__(self, model: RepVGG, calibration: torch.Tensor) -> None:
        super().__init__()
        _lib.require_gpu(calibration)
        blocks: List[RepBlock] = [b for stage in model.features for b in stage]
        if any(not isinstance(b.branches, nn.Conv2d) for b in blocks):
            raise ValueError("Fp8RepVGG needs a re-parametrised model: call model.reparametrize() first")
        if any(not isinstance(b.activation, nn.ReLU) for b in blocks):
            raise NotImplementedError("the fp8 epilogue fuses ReLU only")
        self.head = model.head
        dev = calibration.device
        # ---- calibration: per-tensor amax of the input and of every block output on the bf16 inference path -----------
        with torch.no_grad():
            amax = [float(calibration.abs().amax())]
            h = calibration
            for b in blocks:
                h = b(h)
                amax.append(float(h.float().abs().amax()))
        scales = [max(a, 1e-6) / FP8_MAX for a in amax]
        self.input_scale = scales[0]
        self.act_scales = scales[1:]
        self.layers: List[_Layer] = []
        for i, b in enumerate(blocks):
            conv = b.branches
            Cout, Cin, KH, KW = conv.weight.shape
            L = _Layer()
            L.stride, L.cout, L.cout_p = conv.stride[0], Cout, _ceil64(Cout)
            sx_in, sx_out = scales[i], scales[i + 1]
            if i == 0:
                # stem: im2col over (kh, kw, ci) -> K = 27, one 64-byte fp8 k-step; the conv becomes a 1x1 over it
                K = Cin * KH * KW
                L.im2col_k = (K + 15) // 16 * 16
                L.cin_p = 64
                w2 = conv.weight.detach().permute(0, 2, 3, 1).reshape(Cout, K, 1, 1)     # k = (kh*KW + kw)*Cin + ci
                wq, sw = quantize_weight_fp8(w2, L.cin_p, L.cout_p)
            else:
                L.im2col_k = 0
                L.cin_p = _ceil64(Cin)
                wq, sw = quantize_weight_fp8(conv.weight, L.cin_p, L.cout_p)
            L.wq = wq
            L.mult = (sw * (sx_in / sx_out)).contiguous()
            L.badd = torch.zeros((L.cout_p,), dtype=torch.float32, device=dev)
            L.badd[:Cout] = conv.bias.detach().float() / sx_out
            L.sx_out = sx_out
            L.desc = {}
            self.layers.append(L)
        self._conv_meta = [(b.branches.kernel_size[0], b.branches.padding[0]) for b in blocks]

    def _conv(self, L: _Layer, i: int, src: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
        """src: uint8 [N, H, W, cin_p]; returns uint8 [N, OH, OW, cout_p]."""
        k, pad = self._conv_meta[i]
        if L.im2col_k:
            k, pad, stride = 1, 0, 1
        else:
            stride = L.stride
        key = (N, H, W)
        if key not in L.desc:
            L.desc[key] = cv.fwd_desc(N, L.cin_p, H, W, L.cout_p, k, k, stride, pad)
        d = L.desc[key]
        out = torch.empty((N, d.OH, d.OW, L.cout_p), dtype=torch.uint8, device=src.device)
        d.ch_mult = ptr(L.mult)
        try:
            cv.launch_conv(d, src, L.wq, out, bias=L.badd, act=1, flops=2.0 * N * d.OH * d.OW * L.cout * (L.wq.shape[1] * L.wq.shape[2]))
        finally:
            d.ch_mult = None
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(x)
        lib = _lib.load()
        N = x.shape[0]
        L0 = self.layers[0]
        k, pad = self._conv_meta[0]
        xc = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        H0, W0 = xc.shape[2], xc.shape[3]
        OH, OW = cv.conv_out_size(H0, k, L0.stride, pad), cv.conv_out_size(W0, k, L0.stride, pad)
        h = torch.empty((N, OH, OW, 64), dtype=torch.uint8, device=x.device)
        check(lib.hc_im2col_small_fp8(ptr(xc), ptr(h), N, xc.shape[1], H0, W0, OH, OW, k, k, L0.stride, pad, 64, 1.0 / self.input_scale,
                                      stream()), "hc_im2col_small_fp8")
        H, W = OH, OW
        for i, L in enumerate(self.layers):
            h = self._conv(L, i, h, N, H, W)
            H, W = h.shape[1], h.shape[2]
        Llast = self.layers[-1]
        pooled = torch.empty((N, Llast.cout), dtype=torch.float32, device=x.device)
        check(lib.hc_gap_fp8(ptr(h), ptr(pooled), N, H * W, Llast.cout_p, Llast.cout, Llast.sx_out, stream()), "hc_gap_fp8")
        return self.head(pooled)
