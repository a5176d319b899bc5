"""NormConv2d on the HIP kernels (reference: holocron/nn/modules/conv.py:55-147, holocron/nn/functional.py:322-413).

The reference unfolds the input, normalises every patch (mean / biased variance over its Cin*KH*KW entries) and multiplies
by the flattened filters.  Here nothing is unfolded:
    out[p][co] = rstd_p * (conv(x, W)[p][co] - mean_p * sum_k W[co][k]) + bias[co]
so the MFMA gather-conv runs on x itself, hc_patch_stats supplies (mean_p, rstd_p) and the conv epilogue applies the
affine to its fp32 accumulators.  Backward: dW = wgrad(x, g * rstd) - sum_p g rstd mean (the same for every tap),
db = sum_p g.  The reference normalises the unfolded patches IN PLACE, so autograd cannot back-propagate into an input
that requires grad (RuntimeError in torch >= 1.5); the same holds here.
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib
from .._lib import check, ptr, stream
from ..ops import conv as cv
from .convbn_op import ConvState, cl_ld
from .mbconv_op import _pack_padded, ceil16, pad_channels


class NormConv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, st, meta):
        stride, pad, eps = meta
        lib = _lib.load()
        Cout, Cin, KH, KW = w.shape
        Cin_p, Cout_p = ceil16(Cin), ceil16(Cout)
        xp = pad_channels(x.detach(), Cin_p)
        N, _, H, W = xp.shape
        dev = xp.device
        wf, _ = _pack_padded(st, w, Cin_p, Cout_p, False)
        key = ("fn", N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
        if key not in st.desc:
            st.desc[key] = cv.fwd_desc(N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
        fd = st.desc[key]
        npix = N * fd.OH * fd.OW
        mean = torch.empty((npix,), dtype=torch.float32, device=dev)
        rstd = torch.empty((npix,), dtype=torch.float32, device=dev)
        check(lib.hc_patch_stats(ptr(xp), Cin_p, ptr(mean), ptr(rstd), N, H, W, Cin, KH, KW, stride, pad, eps, stream()), "hc_patch_stats")
        # sum of the bf16-rounded filters (what the MFMA conv multiplies by)
        wsum = torch.zeros((Cout_p,), dtype=torch.float32, device=dev)
        wsum[:Cout] = w.detach().to(torch.bfloat16).float().sum(dim=(1, 2, 3))
        bp = torch.zeros((Cout_p,), dtype=torch.float32, device=dev)
        if bias is not None:
            bp[:Cout] = bias.detach().float()
        y = cv.empty_cl(N, Cout_p, fd.OH, fd.OW, dev)
        fd.pix_scale, fd.pix_shift, fd.ch_coef = ptr(rstd), ptr(mean), ptr(wsum)
        try:
            cv.launch_conv(fd, xp, wf, y, bias=bp, act=0, flops=2.0 * npix * Cout * Cin * KH * KW)
        finally:
            fd.pix_scale = fd.pix_shift = fd.ch_coef = None
        ctx.save_for_backward(xp, mean, rstd, w)
        ctx.geom = (N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, fd.OH, fd.OW, stride, pad, bias is not None)
        out = y if Cout_p == Cout else y[:, :Cout]
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("NormConv2d: the reference normalises the unfolded input in place (functional.py:347-349) and "
                               "cannot back-propagate into an input that requires grad; neither does this path")
        xp, mean, rstd, w = ctx.saved_tensors
        N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, OH, OW, stride, pad, has_bias = ctx.geom
        lib = _lib.load()
        dev = g.device
        gp = pad_channels(g, Cout_p)
        gs = torch.empty_like(gp)
        red = torch.zeros((_lib.stat_replicas(), 2, Cout_p), dtype=torch.float32, device=dev)
        check(lib.hc_normconv_bwd_scale(ptr(gp), ptr(mean), ptr(rstd), ptr(gs), ptr(red), N * OH * OW, Cout_p, stream()),
              "hc_normconv_bwd_scale")
        sums = red.sum(0)
        dwp = cv.conv_wgrad(xp, gs, Cin_p, Cout_p, KH, KW, stride, pad, flops=2.0 * N * OH * OW * Cout * Cin * KH * KW)
        dw = dwp[:Cout, :Cin] - sums[1, :Cout].view(-1, 1, 1, 1)
        db = sums[0, :Cout].clone() if has_bias else None
        return None, dw.contiguous(), db, None, None


def norm_conv2d_module(x, module):
    """Forward of a NormConv2d-like module (weight, bias, stride, padding, dilation, groups, padding_mode, eps)."""
    _lib.require_gpu(x)
    if module.dilation != (1, 1) or module.stride[0] != module.stride[1] or module.padding[0] != module.padding[1] \
            or module.kernel_size[0] * module.kernel_size[1] > _lib.HC_MAX_TAPS:
        raise NotImplementedError("NormConv2d on the HIP path: square stride / padding, dilation 1, at most 12 taps")
    # `groups` is accepted and ignored, exactly like the reference's _xcorr2d (functional.py:322-363): only groups == 1
    # gives a weight whose flattened width matches the patches
    if module.groups != 1:
        raise RuntimeError("NormConv2d: the reference ignores `groups` and fails on the matmul shapes for groups != 1")
    st = getattr(module, "_hcp", None)
    if st is None:
        st = module._hcp = ConvState()
    pad = module.padding[0]
    if module.padding_mode != "zeros":
        x = torch.nn.functional.pad(x, module._reversed_padding_repeated_twice, mode=module.padding_mode)   # conv.py:127-137
        pad = 0
    return NormConv2dFn.apply(x, module.weight, module.bias, st, (module.stride[0], pad, float(module.eps)))
