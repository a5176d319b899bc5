"""Fused conv -> BatchNorm2d -> activation [+ residual] on the HIP kernels.

This is the unit ``conv_sequence`` builds (reference: holocron/models/utils.py:61-84) and what the
DarkNet / CSP / YOLO stacks are made of (darknetv3.py:23-70).  Kernel sequence:
  forward : gather-conv (sum/sumsq epilogue) -> rep_bn_finalize (one branch) -> bn_act_apply
  backward: bn_act_bwd_reduce -> rep_bn_bwd_finalize -> bn_act_bwd_apply (dy) -> gather-conv dgrad, wgrad
The pre-activation is recomputed from the stored conv output in backward instead of being kept.
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib
from .._lib import RepBnBwdDesc, RepBnDesc, check, ptr, stream
from ..ops import conv as cv
from .repblock_op import POOL


def act_code(act):
    """(code, slope) of an activation module for the fused kernels, or None if it is not fusable."""
    from .modules import HardMish
    if act is None:
        return 0, 0.0
    if isinstance(act, nn.ReLU):
        return 1, 0.0
    if isinstance(act, HardMish):
        return 2, 0.0
    if isinstance(act, nn.LeakyReLU):
        return 3, float(act.negative_slope)
    if isinstance(act, nn.Mish):
        return 4, 0.0
    if isinstance(act, nn.SiLU):
        return 5, 0.0
    if isinstance(act, nn.ReLU6):
        return 6, 0.0
    return None


class ConvState:
    """Host state of one conv+bn pair: packed weights and geometry descriptors."""

    def __init__(self):
        self.fwd_cache = cv.PackCache()
        self.bwd_cache = cv.PackCache()
        self.desc = {}


class ConvBnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, res, st, meta):
        lib = _lib.load()
        stride, pad, act, slope, bnbuf, eps, momentum, training = meta
        Cout, Cin, KH, KW = w.shape
        N, _, H, W = x.shape
        dev = x.device
        im2col = (Cin % 16) != 0
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise RuntimeError("conv_bn_act (HIP) expects contiguous fp32 conv weights")
        if im2col:
            K = Cin * KH * KW
            Kpad = (K + 15) // 16 * 16
            src = cv.im2col_small(x, KH, KW, stride, pad, Kpad)
            wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight_im2col(w, Kpad))
            key = ("c", N, src.shape[2], src.shape[3], Kpad, Cout)
            if key not in st.desc:
                st.desc[key] = cv.fwd_desc(N, Kpad, src.shape[2], src.shape[3], Cout, 1, 1, 1, 0)
            fd = st.desc[key]
            flops = 2.0 * N * fd.OH * fd.OW * Cout * K
        else:
            src = cv.to_cl_bf16(x)
            wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight(w, 0))
            key = ("f", N, Cin, H, W, Cout, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.fwd_desc(N, Cin, H, W, Cout, KH, KW, stride, pad)
            fd = st.desc[key]
            flops = None
        OH, OW = fd.OH, fd.OW
        npix = N * OH * OW
        y = cv.empty_cl(N, Cout, OH, OW, dev)
        stats = POOL.take((_lib.HC_STAT_REPLICAS, 2, Cout), dev) if training else None
        cv.launch_conv(fd, src, wpk, y, stats=stats, flops=flops)

        coef = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        save = torch.empty((6, Cout), dtype=torch.float32, device=dev)
        d = RepBnDesc()
        for b in range(3):
            d.stats[b] = d.gamma[b] = d.beta[b] = None
            d.running_mean[b] = d.running_var[b] = d.num_batches_tracked[b] = None
        rm, rv, nbt = bnbuf
        d.stats[0], d.gamma[0], d.beta[0] = ptr(stats), ptr(gamma), ptr(beta)
        d.running_mean[0], d.running_var[0], d.num_batches_tracked[0] = ptr(rm), ptr(rv), ptr(nbt)
        d.coef, d.save, d.C, d.count = ptr(coef), ptr(save), Cout, npix
        d.eps, d.momentum, d.training = eps, momentum, 1 if training else 0
        check(lib.hc_rep_bn_finalize(C.byref(d), stream()), "hc_rep_bn_finalize")

        resc = None if res is None else cv.to_cl_bf16(res)
        out = cv.empty_cl(N, Cout, OH, OW, dev)
        check(lib.hc_bn_act_apply(ptr(y), ptr(coef), ptr(resc), ptr(out), npix, Cout, act, slope, stream()), "hc_bn_act_apply")
        ctx.st, ctx.meta2 = st, (stride, pad, act, slope, im2col, training)
        ctx.geom = (N, Cin, H, W, Cout, KH, KW, OH, OW)
        ctx.red = POOL.take((_lib.HC_STAT_REPLICAS, 4, Cout), dev) if training else None
        ctx.has_res = res is not None
        ctx.save_for_backward(src, y, coef, save, gamma, w)
        return out

    @staticmethod
    def backward(ctx, g):
        stride, pad, act, slope, im2col, training = ctx.meta2
        if not training:
            raise NotImplementedError("conv_bn_act backward in eval mode (running statistics) is not implemented")
        lib = _lib.load()
        st = ctx.st
        src, y, coef, save, gamma, w = ctx.saved_tensors
        N, Cin, H, W, Cout, KH, KW, OH, OW = ctx.geom
        dev = g.device
        g = cv.to_cl_bf16(g)
        npix = N * OH * OW
        red = ctx.red
        ctx.red = None
        if red is None:
            red = torch.zeros((_lib.HC_STAT_REPLICAS, 4, Cout), dtype=torch.float32, device=dev)
        check(lib.hc_bn_act_bwd_reduce(ptr(g), ptr(y), ptr(coef), ptr(red), npix, Cout, act, slope, stream()),
              "hc_bn_act_bwd_reduce")
        dgam = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbet = torch.empty((Cout,), dtype=torch.float32, device=dev)
        bcoef = torch.empty((9, Cout), dtype=torch.float32, device=dev)
        d = RepBnBwdDesc()
        d.red, d.save, d.bcoef = ptr(red), ptr(save), ptr(bcoef)
        for b in range(3):
            d.gamma[b] = d.dgamma[b] = d.dbeta[b] = None
        d.gamma[0], d.dgamma[0], d.dbeta[0] = ptr(gamma), ptr(dgam), ptr(dbet)
        d.C, d.count, d.has_identity, d.accumulate = Cout, npix, 0, 0
        check(lib.hc_rep_bn_bwd_finalize(C.byref(d), stream()), "hc_rep_bn_bwd_finalize")
        dy = torch.empty_like(y)
        check(lib.hc_bn_act_bwd_apply(ptr(g), ptr(y), ptr(coef), ptr(bcoef), ptr(dy), npix, Cout, act, slope, stream()),
              "hc_bn_act_bwd_apply")

        dx = None
        if ctx.needs_input_grad[0]:
            if im2col:
                raise NotImplementedError("input gradient of the im2col (Cin % 16 != 0) path")
            key = ("d", N, Cin, H, W, Cout, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.dgrad_desc(N, Cin, H, W, Cout, [(KH, KW, pad, 0, 0)], stride)
            wpd = st.bwd_cache.get((w,), lambda: cv.pack_weight(w, 1))
            dx = cv.empty_cl(N, Cin, H, W, dev)
            cv.launch_conv(st.desc[key], dy, wpd, dx)
        if im2col:
            Kpad = src.shape[1]
            dwc = cv.conv_wgrad(src, dy, Kpad, Cout, 1, 1, 1, 0, flops=2.0 * npix * Cout * Cin * KH * KW)
            dw = torch.empty_like(w, dtype=torch.float32)
            check(lib.hc_unpack_im2col_grad(ptr(dwc), ptr(dw), Cout, Cin, KH, KW, Kpad, 0, stream()), "hc_unpack_im2col_grad")
        else:
            dw = cv.conv_wgrad(src, dy, Cin, Cout, KH, KW, stride, pad)
        return dx, dw, dgam, dbet, (g if ctx.has_res else None), None, None


def fusable(conv, bn, act):
    return (isinstance(conv, nn.Conv2d) and type(conv) is nn.Conv2d and isinstance(bn, nn.BatchNorm2d)
            and conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and conv.stride[0] in (1, 2)
            and conv.kernel_size[0] * conv.kernel_size[1] <= _lib.HC_MAX_TAPS and conv.padding_mode == "zeros"
            and conv.out_channels % 8 == 0 and act_code(act) is not None)


def conv_bn_act(x, conv, bn, act=None, residual=None):
    """out = act(bn(conv(x))) [+ residual] on the fused HIP path."""
    st = getattr(conv, "_hc", None)
    if st is None:
        st = conv._hc = ConvState()
    code, slope = act_code(act)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    meta = (conv.stride[0], conv.padding[0], code, slope,
            (bn.running_mean, bn.running_var, bn.num_batches_tracked), bn.eps, momentum, bn.training)
    return ConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, residual, st, meta)


def run_conv_sequence(seq, x, residual=None):
    """Execute the modules of a ``conv_sequence`` list / nn.Sequential, fusing every
    [Conv2d, BatchNorm2d, activation?] run into one conv_bn_act call.  ``residual`` is added to the
    output of the LAST fused unit (DarkNet ResBlock: ``out = conv(x); out += identity``)."""
    mods = list(seq)
    i, n = 0, len(mods)
    units = []
    while i < n:
        m = mods[i]
        if i + 1 < n and isinstance(m, nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d):
            act = mods[i + 2] if (i + 2 < n and act_code(mods[i + 2]) is not None and mods[i + 2] is not None
                                  and not isinstance(mods[i + 2], (nn.Conv2d, nn.BatchNorm2d))) else None
            if fusable(m, mods[i + 1], act):
                units.append(("fused", m, mods[i + 1], act))
                i += 3 if act is not None else 2
                continue
        units.append(("module", m))
        i += 1
    last_fused = max((k for k, u in enumerate(units) if u[0] == "fused"), default=-1)
    for k, u in enumerate(units):
        if u[0] == "fused":
            x = conv_bn_act(x, u[1], u[2], u[3], residual if (k == last_fused and k == len(units) - 1) else None)
        else:
            x = u[1](x)
    if residual is not None and not (last_fused == len(units) - 1 and last_fused >= 0):
        x = x + residual
    return x
