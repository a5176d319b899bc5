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


def cl_ld(t):
    """Channels per pixel of the buffer a bf16 NHWC tensor (possibly a channel-slice view of a wider concat
    buffer) lives in, or None when the tensor is not such a view."""
    if t.dtype != torch.bfloat16 or t.dim() != 4:
        return None
    N, Cc, H, W = t.shape
    ld = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else (t.stride(0) if N > 1 else Cc))
    if Cc > 1 and t.stride(1) != 1:
        return None
    if ld < Cc or ld % 8 or Cc % 8 or t.data_ptr() % 16:
        return None
    if (W > 1 and t.stride(3) != ld) or (H > 1 and t.stride(2) != W * ld) or (N > 1 and t.stride(0) != H * W * ld):
        return None
    return ld


def as_cl_view(t):
    """(tensor, ld): the tensor itself when the kernels can read it in place, else a dense NHWC bf16 copy."""
    ld = cl_ld(t)
    if ld is not None:
        return t, ld
    t = cv.to_cl_bf16(t)
    return t, t.shape[1]


def dropblock_keep(N, H, W, drop_prob, block_size, device):
    """keep map + count of one DropBlock call (batched per step when a DropPlan is active, see nn/functional.py)."""
    from . import functional as Fh
    return Fh.dropblock_keep(N, H, W, drop_prob, block_size, device)


class ConvState:
    """Host state of one conv+bn pair: packed weights and geometry descriptors."""

    def __init__(self):
        self.fwd_cache = cv.PackCache()
        self.bwd_cache = cv.PackCache()
        self.desc = {}


class ConvBnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, res, st, meta, out_holder=None, link=None):
        lib = _lib.load()
        cv._WCONV.note_forward()       # a weight-gradient queue still armed by a backward pass that died is emptied here
        out = None if out_holder is None else out_holder[0]
        # ``link`` couples the two units of a residual block whose shortcut is the block INPUT (DarkNet ResBlock): the unit that adds
        # the residual ("sink") parks the residual's gradient in link["grad"] instead of returning it, and the unit that consumes the
        # block input ("source", whose backward runs later) hands it to its data-gradient launch as `resid` - dx arrives complete and
        # autograd has no second gradient of x to add (one elementwise launch per block and step)
        ctx.link = link
        stride, pad, act, slope, bnbuf, eps, momentum, training, drop = meta[:9]
        drop2 = meta[9] if len(meta) > 9 else None       # DropBlock BEHIND the residual add (DarkNet ResBlock), same pass
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
        stats = POOL.take((_lib.stat_replicas(), 2, Cout), dev) if training else None
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
        keep = count = None
        if drop is not None and training and drop[0] > 0:
            keep, count = dropblock_keep(N, OH, OW, drop[0], drop[1], dev)
        keep2 = count2 = None
        if drop2 is not None and drop2[0] > 0:           # drawn AFTER the unit's own mask: the order the separate modules drew in
            keep2, count2 = dropblock_keep(N, OH, OW, drop2[0], drop2[1], dev)
        if out is None:
            out = cv.empty_cl(N, Cout, OH, OW, dev)
        out_ld = cl_ld(out)
        if out_ld is None or tuple(out.shape) != (N, Cout, OH, OW):
            raise _lib.HipError("conv_bn_act: `out` must be an NHWC bf16 view of shape %s" % ((N, Cout, OH, OW),))
        with cv.profiled("bn_elementwise", 0.0, npix * Cout * 2.0 * (3 if resc is not None else 2)):
            check(lib.hc_bn_act_apply_post(ptr(y), ptr(coef), ptr(resc), Cout if resc is not None else 0, ptr(keep), ptr(count),
                                           ptr(keep2), ptr(count2), ptr(out), out_ld, npix, Cout, act, slope, stream()),
                  "hc_bn_act_apply_post")
        ctx.drop = (keep, count)
        ctx.drop2 = (keep2, count2)
        ctx.st, ctx.meta2 = st, (stride, pad, act, slope, im2col, training)
        ctx.geom = (N, Cin, H, W, Cout, KH, KW, OH, OW)
        ctx.red, ctx.red_gen = POOL.take_for_backward((_lib.stat_replicas(), 4, Cout), dev) if training else (None, -1)
        ctx.has_res = res is not None
        ctx.save_for_backward(src, y, coef, save, gamma, w)
        return out

    @staticmethod
    def backward(ctx, g):
        stride, pad, act, slope, im2col, training = ctx.meta2
        lib = _lib.load()
        st = ctx.st
        src, y, coef, save, gamma, w = ctx.saved_tensors
        N, Cin, H, W, Cout, KH, KW, OH, OW = ctx.geom
        dev = g.device
        g, g_ld = as_cl_view(g)
        keep, count = ctx.drop
        keep2, count2 = ctx.drop2
        npix = N * OH * OW
        red = POOL.claim(ctx.red, ctx.red_gen, (_lib.stat_replicas(), 4, Cout), dev)   # stale after another forward's POOL.begin()
        ctx.red = None
        with cv.profiled("bn_elementwise", 0.0, npix * Cout * 2.0 * 2):
            check(lib.hc_bn_act_bwd_reduce_post(ptr(g), g_ld, ptr(y), ptr(coef), ptr(keep), ptr(count), ptr(keep2), ptr(count2), ptr(red),
                                                npix, Cout, act, slope, stream()), "hc_bn_act_bwd_reduce_post")
        dgam = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbet = torch.empty((Cout,), dtype=torch.float32, device=dev)
        bcoef = torch.empty((9, Cout), dtype=torch.float32, device=dev)
        d = RepBnBwdDesc()
        d.red, d.save, d.bcoef = ptr(red), ptr(save), ptr(bcoef)
        for b in range(3):
            d.gamma[b] = d.dgamma[b] = d.dbeta[b] = None
        d.gamma[0], d.dgamma[0], d.dbeta[0] = ptr(gamma), ptr(dgam), ptr(dbet)
        d.C, d.count, d.has_identity, d.accumulate = Cout, npix, 0, 0
        d.frozen = 0 if training else 1               # eval mode / freeze_bn: running statistics, dy = a * dz
        check(lib.hc_rep_bn_bwd_finalize(C.byref(d), stream()), "hc_rep_bn_bwd_finalize")
        dy = torch.empty_like(y)
        # the residual input's gradient: the incoming one, or - behind a fused post-residual DropBlock - its masked form
        gres = torch.empty_like(y) if (keep2 is not None and ctx.has_res) else None
        with cv.profiled("bn_elementwise", 0.0, npix * Cout * 2.0 * (3 if gres is None else 4)):
            check(lib.hc_bn_act_bwd_apply_post(ptr(g), g_ld, ptr(y), ptr(coef), ptr(bcoef), ptr(keep), ptr(count), ptr(keep2), ptr(count2),
                                               ptr(gres), ptr(dy), npix, Cout, act, slope, stream()), "hc_bn_act_bwd_apply_post")

        dx = None
        if ctx.needs_input_grad[0]:
            if im2col:
                raise NotImplementedError("input gradient of the im2col (Cin % 16 != 0) path")
            key = ("d", N, Cin, H, W, Cout, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.dgrad_desc(N, Cin, H, W, Cout, [(KH, KW, pad, 0, 0)], stride)
            wpd = st.bwd_cache.get((w,), lambda: cv.pack_weight(w, 1))
            dx = cv.empty_cl(N, Cin, H, W, dev)
            extra = None
            if ctx.link is not None and ctx.link.get("role") == "source":
                extra = ctx.link.pop("grad", None)          # parked by the block's last unit (its backward ran first)
                if extra is not None:
                    if tuple(extra.shape) != (N, Cin, H, W):
                        raise RuntimeError("linked residual gradient does not have the block input's shape")
                    if cl_ld(extra) != Cin:
                        extra = extra.contiguous(memory_format=torch.channels_last)
            cv.launch_conv(st.desc[key], dy, wpd, dx, resid=extra)
        with cv.side_stream_for_wgrad((w,), (src, dy)) as side:
            if im2col:
                Kpad = src.shape[1]
                dwc = cv.conv_wgrad(src, dy, Kpad, Cout, 1, 1, 1, 0, flops=2.0 * npix * Cout * Cin * KH * KW)
                dw = torch.empty_like(w, dtype=torch.float32)
                check(lib.hc_unpack_im2col_grad(ptr(dwc), ptr(dw), Cout, Cin, KH, KW, Kpad, 0, stream()), "hc_unpack_im2col_grad")
            else:
                dw = cv.conv_wgrad_unit(src, dy, w, Cin, Cout, KH, KW, stride, pad)
            side.produced(dw)
        gr = (gres if gres is not None else g) if ctx.has_res else None
        if gr is not None and ctx.link is not None and ctx.link.get("role") == "sink" and ctx.link.get("armed"):
            peer = ctx.link["peer"]
            peer["grad"] = gr if g_ld == Cout or gres is not None else gr.contiguous(memory_format=torch.channels_last)
            gr = None                                       # delivered through the source unit's dx
        return dx, dw, dgam, dbet, gr, None, None, None, None


class ConvBiasFn(torch.autograd.Function):
    """Plain conv + bias, no normalisation / activation (the YOLO head outputs, yolov4.py:480,536,598).  The output
    keeps ``ceil16(Cout)`` channels per pixel (zero weights / bias in the pad rows); callers slice the logical view."""

    @staticmethod
    def forward(ctx, x, w, bias, st, meta):
        stride, pad = meta
        Cout, Cin, KH, KW = w.shape
        N, _, H, W = x.shape
        dev = x.device
        if Cin % 16:
            raise NotImplementedError("conv+bias on the HIP path needs Cin % 16 == 0")
        Cp = (Cout + 15) // 16 * 16   # the data-gradient conv reads dy with Cp channels per pixel: % 16
        src = cv.to_cl_bf16(x)

        def padded():
            if Cp == Cout:
                return w
            return torch.cat([w, w.new_zeros((Cp - Cout, Cin, KH, KW))], 0)
        wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight(padded(), 0))
        bp = torch.zeros((Cp,), dtype=torch.float32, device=dev)
        if bias is not None:
            bp[:Cout] = bias.detach().float()
        key = ("fb", N, Cin, H, W, Cp, KH, KW, stride, pad)
        if key not in st.desc:
            st.desc[key] = cv.fwd_desc(N, Cin, H, W, Cp, KH, KW, stride, pad)
        fd = st.desc[key]
        y = cv.empty_cl(N, Cp, fd.OH, fd.OW, dev)
        cv.launch_conv(fd, src, wpk, y, bias=bp, act=0)
        ctx.st, ctx.meta2 = st, (stride, pad, Cp, bias is not None)
        ctx.geom = (N, Cin, H, W, Cout, KH, KW, fd.OH, fd.OW)
        ctx.save_for_backward(src, w)
        return y

    @staticmethod
    def backward(ctx, g):
        src, w = ctx.saved_tensors
        st = ctx.st
        stride, pad, Cp, has_bias = ctx.meta2
        N, Cin, H, W, Cout, KH, KW, OH, OW = ctx.geom
        dev = g.device
        dy = cv.to_cl_bf16(g)
        lib = _lib.load()
        db = None
        if has_bias:
            stats = torch.zeros((_lib.stat_replicas(), 2, Cp), dtype=torch.float32, device=dev)
            check(lib.hc_channel_stats(ptr(dy), ptr(stats), N * OH * OW, Cp, stream()), "hc_channel_stats")
            db = stats[:, 0].sum(0)[:Cout]
        dx = None
        if ctx.needs_input_grad[0]:
            key = ("db", N, Cin, H, W, Cp, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.dgrad_desc(N, Cin, H, W, Cp, [(KH, KW, pad, 0, 0)], stride)

            def padded():
                if Cp == Cout:
                    return w
                return torch.cat([w, w.new_zeros((Cp - Cout, Cin, KH, KW))], 0)
            wpd = st.bwd_cache.get((w,), lambda: cv.pack_weight(padded(), 1))
            dx = cv.empty_cl(N, Cin, H, W, dev)
            cv.launch_conv(st.desc[key], dy, wpd, dx)
        dw = cv.conv_wgrad(src, dy, Cin, Cp, KH, KW, stride, pad)[:Cout]
        return dx, dw, db, None, None


class ConvBiasActFn(torch.autograd.Function):
    """act(conv(x) + bias) with act in {none, ReLU, LeakyReLU}, no normalisation: the conv_sequence unit of a model built with
    ``norm_layer=None`` (YOLOv1's default, yolo.py:233-296).  Bias and activation ride in the gather-conv epilogue; the backward
    pass rebuilds the activation mask from the stored output.  Cin % 16 != 0 (a 3-channel stem) goes through the im2col
    column tensor like the BatchNorm units."""

    @staticmethod
    def forward(ctx, x, w, bias, st, meta):
        stride, pad, act, slope = meta
        Cout, Cin, KH, KW = w.shape
        N, _, H, W = x.shape
        dev = x.device
        if Cout % 16:
            raise NotImplementedError("conv+bias+act on the HIP path needs Cout % 16 == 0")
        if act == 3 and abs(slope - 0.1) > 1e-9:
            raise NotImplementedError("the gather-conv epilogue fuses LeakyReLU(0.1) only")
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise RuntimeError("conv_bias_act (HIP) expects contiguous fp32 conv weights")
        im2col = (Cin % 16) != 0
        if im2col:
            K = Cin * KH * KW
            Kpad = (K + 15) // 16 * 16
            src = cv.im2col_small(x, KH, KW, stride, pad, Kpad)
            wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight_im2col(w, Kpad))
            key = ("cb", N, src.shape[2], src.shape[3], Kpad, Cout)
            if key not in st.desc:
                st.desc[key] = cv.fwd_desc(N, Kpad, src.shape[2], src.shape[3], Cout, 1, 1, 1, 0)
        else:
            src = cv.to_cl_bf16(x)
            wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight(w, 0))
            key = ("fba", N, Cin, H, W, Cout, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.fwd_desc(N, Cin, H, W, Cout, KH, KW, stride, pad)
        fd = st.desc[key]
        bkey = None if bias is None else (bias.data_ptr(), bias._version, cv.weights_epoch())
        if getattr(st, "bias_key", ()) != bkey or getattr(st, "bias_f", None) is None or st.bias_f.device != dev:
            st.bias_f = torch.zeros((Cout,), dtype=torch.float32, device=dev) if bias is None else bias.detach().float().contiguous()
            st.bias_key = bkey
        out = cv.empty_cl(N, Cout, fd.OH, fd.OW, dev)
        cv.launch_conv(fd, src, wpk, out, bias=st.bias_f, act=act, flops=2.0 * N * fd.OH * fd.OW * Cout * Cin * KH * KW)
        ctx.st, ctx.meta2 = st, (stride, pad, act, slope, im2col, bias is not None)
        ctx.geom = (N, Cin, H, W, Cout, KH, KW, fd.OH, fd.OW)
        ctx.save_for_backward(src, w, out)
        return out

    @staticmethod
    def backward(ctx, g):
        src, w, out = ctx.saved_tensors
        st = ctx.st
        stride, pad, act, slope, im2col, has_bias = ctx.meta2
        N, Cin, H, W, Cout, KH, KW, OH, OW = ctx.geom
        dev = g.device
        lib = _lib.load()
        g, g_ld = as_cl_view(g)
        npix = N * OH * OW
        if act == 0:
            dy = g if g_ld == Cout else g.contiguous(memory_format=torch.channels_last)
        else:
            dy = torch.empty_like(out)
            check(lib.hc_leaky_bwd(ptr(g), g_ld, ptr(out), ptr(dy), npix, Cout, slope if act == 3 else 0.0, stream()), "hc_leaky_bwd")
        db = None
        if has_bias:
            stats = torch.zeros((_lib.stat_replicas(), 2, Cout), dtype=torch.float32, device=dev)
            check(lib.hc_channel_stats(ptr(dy), ptr(stats), npix, Cout, stream()), "hc_channel_stats")
            db = stats[:, 0].sum(0)
        dx = None
        if ctx.needs_input_grad[0]:
            if im2col:
                raise NotImplementedError("input gradient of the im2col (Cin % 16 != 0) path")
            key = ("dba", N, Cin, H, W, Cout, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.dgrad_desc(N, Cin, H, W, Cout, [(KH, KW, pad, 0, 0)], stride)
            wpd = st.bwd_cache.get((w,), lambda: cv.pack_weight(w, 1))
            dx = cv.empty_cl(N, Cin, H, W, dev)
            cv.launch_conv(st.desc[key], dy, wpd, dx)
        with cv.side_stream_for_wgrad((w,), (src, dy)) as side:
            if im2col:
                Kpad = src.shape[1]
                dwc = cv.conv_wgrad(src, dy, Kpad, Cout, 1, 1, 1, 0, flops=2.0 * npix * Cout * Cin * KH * KW)
                dw = torch.empty_like(w, dtype=torch.float32)
                check(lib.hc_unpack_im2col_grad(ptr(dwc), ptr(dw), Cout, Cin, KH, KW, Kpad, 0, stream()), "hc_unpack_im2col_grad")
            else:
                dw = cv.conv_wgrad_unit(src, dy, w, Cin, Cout, KH, KW, stride, pad)
            side.produced(dw)
        return dx, dw, db, None, None


def conv_bias_act(x, conv, act):
    """``act(conv(x))`` for a bias-carrying nn.Conv2d followed directly by ReLU / LeakyReLU(0.1) (no BatchNorm)."""
    st = getattr(conv, "_hc", None)
    if st is None:
        st = conv._hc = ConvState()
    code = act_code(act)
    if not (type(conv) is nn.Conv2d and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == "zeros"
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.padding[0] == conv.padding[1] and code is not None and code[0] in (0, 1, 3)
            and (conv.in_channels % 16 != 0 or conv.kernel_size[0] * conv.kernel_size[1] <= _lib.HC_MAX_TAPS)):
        raise NotImplementedError(f"conv+bias+act unit outside the HIP path: {conv}, {act}")
    return ConvBiasActFn.apply(x, conv.weight, conv.bias, st, (conv.stride[0], conv.padding[0], code[0], code[1]))


def conv_bias(x, conv):
    """``conv(x)`` for a bias-carrying nn.Conv2d with no BN / activation after it, as an NHWC bf16 tensor whose
    channel count is rounded up to a multiple of 16 (the extra channels are exact zeros)."""
    st = getattr(conv, "_hc", None)
    if st is None:
        st = conv._hc = ConvState()
    if not (type(conv) is nn.Conv2d and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == "zeros"
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1]):
        raise NotImplementedError("conv_bias: unsupported convolution geometry for the HIP path")
    return ConvBiasFn.apply(x, conv.weight, conv.bias, st, (conv.stride[0], conv.padding[0]))


def fusable(conv, bn, act):
    return (isinstance(conv, nn.Conv2d) and type(conv) is nn.Conv2d and isinstance(bn, nn.BatchNorm2d)
            and conv.groups == 1 and conv.dilation == (1, 1) and conv.bias is None
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and conv.stride[0] in (1, 2)
            and (conv.in_channels % 16 != 0 or conv.kernel_size[0] * conv.kernel_size[1] <= _lib.HC_MAX_TAPS) and conv.padding_mode == "zeros"
            and conv.out_channels % 16 == 0 and act_code(act) is not None)


def conv_bn_act(x, conv, bn, act=None, residual=None, drop=None, out=None, post_drop=None, link=None):
    """out = dropblock(act(bn(conv(x)))) [+ residual] on the fused HIP path.  ``drop``: a DropBlock2d-like module
    (attributes ``drop_prob``, ``block_size``) applied after the activation, or None.  ``out``: NHWC bf16 slice of a
    concat buffer to write the result into.  ``post_drop``: a second DropBlock2d applied to the SUM (DarkNet's ResBlock,
    darknetv3.py:59-61: ``dropblock(x + conv(x))``) in the same passes."""
    st = getattr(conv, "_hc", None)
    if st is None:
        st = conv._hc = ConvState()
    code, slope = act_code(act)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    dp = None
    if drop is not None and drop.training and drop.drop_prob > 0:
        dp = (float(drop.drop_prob), int(drop.block_size))
    dp2 = None
    if post_drop is not None and post_drop.training and post_drop.drop_prob > 0:
        dp2 = (float(post_drop.drop_prob), int(post_drop.block_size))
    if (not bn.training and not torch.is_grad_enabled() and dp is None and dp2 is None and bn.running_mean is not None
            and _INFER_FUSED and (out is None or cl_ld(out) == conv.out_channels)):
        return _conv_bn_act_infer(x, conv, bn, code, slope, residual, st, out)
    meta = (conv.stride[0], conv.padding[0], code, slope,
            (bn.running_mean, bn.running_var, bn.num_batches_tracked), bn.eps, momentum, bn.training, dp, dp2)
    return ConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, residual, st, meta, None if out is None else [out], link)


# HC_INFER_FUSED=0: eval-mode units run conv -> finalize -> apply like the training path (A/B)
_INFER_FUSED = __import__("os").environ.get("HC_INFER_FUSED", "1") != "0"


def _conv_bn_act_infer(x, conv, bn, code, slope, residual, st, out):
    """INFERENCE form of the unit (eval mode, no autograd, no DropBlock): ONE gather-conv launch whose epilogue applies the BatchNorm
    of the running statistics (per-channel scale and shift, fp32, on the fp32 accumulator), the activation and the residual
    (models/utils.py:73-84 in eval mode; north_star's "fused conv+BN+activation epilogues").  The training path needs batch statistics
    between the conv and the normalisation and keeps its three launches; here the unnormalised conv output is never stored and the
    BatchNorm launches (finalize + apply, a third of an eval pass's launches in YOLOv4) are gone.  Scale / shift are cached on the
    unit, keyed on the versions of the four BatchNorm tensors and the optimizer epoch."""
    w = conv.weight
    Cout, Cin, KH, KW = w.shape
    N, _, H, W = x.shape
    dev = x.device
    stride, pad = conv.stride[0], conv.padding[0]
    key = (bn.weight.data_ptr(), bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, cv.weights_epoch(),
           float(bn.eps), str(dev))
    if getattr(st, "infer_key", None) != key:
        with torch.no_grad():
            a = (bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
            st.infer_scale = a
            st.infer_shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * a).contiguous()
        st.infer_key = key
    if w.dtype != torch.float32 or not w.is_contiguous():
        raise RuntimeError("conv_bn_act (HIP) expects contiguous fp32 conv weights")
    if (Cin % 16) != 0:
        K = Cin * KH * KW
        Kpad = (K + 15) // 16 * 16
        src = cv.im2col_small(x, KH, KW, stride, pad, Kpad)
        wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight_im2col(w, Kpad))
        dkey = ("c", N, src.shape[2], src.shape[3], Kpad, Cout)
        if dkey not in st.desc:
            st.desc[dkey] = cv.fwd_desc(N, Kpad, src.shape[2], src.shape[3], Cout, 1, 1, 1, 0)
        flops = 2.0 * N * st.desc[dkey].OH * st.desc[dkey].OW * Cout * K
    else:
        src = cv.to_cl_bf16(x)
        wpk = st.fwd_cache.get((w,), lambda: cv.pack_weight(w, 0))
        dkey = ("f", N, Cin, H, W, Cout, KH, KW, stride, pad)
        if dkey not in st.desc:
            st.desc[dkey] = cv.fwd_desc(N, Cin, H, W, Cout, KH, KW, stride, pad)
        flops = None
    fd = st.desc[dkey]
    if out is None:
        out = cv.empty_cl(N, Cout, fd.OH, fd.OW, dev)
    elif tuple(out.shape) != (N, Cout, fd.OH, fd.OW):
        raise _lib.HipError("conv_bn_act: `out` must be an NHWC bf16 view of shape %s" % ((N, Cout, fd.OH, fd.OW),))
    resc = None if residual is None else cv.to_cl_bf16(residual)
    if resc is not None and cl_ld(resc) != Cout:
        resc = resc.contiguous(memory_format=torch.channels_last)
    cv.launch_conv(fd, src, wpk, out, resid=resc, bias=st.infer_shift, act=code, flops=flops, ch_scale=st.infer_scale,
                   act_slope=slope, resid_after_act=True)
    return out


def _is_act(m):
    return m is not None and not isinstance(m, (nn.Conv2d, nn.BatchNorm2d)) and act_code(m) is not None


def plan_conv_sequence(seq):
    """Group the modules of a ``conv_sequence`` list / nn.Sequential into execution units:
    ("fused", conv, bn, act, drop) | ("convbias", conv) | ("spp", m) | ("drop", m) | ("block", m)."""
    from .modules import DropBlock2d, SPP
    mods = list(seq)
    i, n = 0, len(mods)
    units = []
    while i < n:
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            if i + 1 < n and isinstance(mods[i + 1], nn.BatchNorm2d):
                j = i + 2
                act = None
                if j < n and _is_act(mods[j]):
                    act = mods[j]
                    j += 1
                drop = None
                if j < n and isinstance(mods[j], DropBlock2d):
                    drop = mods[j]
                    j += 1
                if not fusable(m, mods[i + 1], act):
                    raise NotImplementedError(f"conv/bn/act unit outside the HIP path: {m}, {mods[i + 1]}, {act}")
                units.append(("fused", m, mods[i + 1], act, drop))
                i = j
                continue
            if m.bias is not None:
                if i + 1 < n and _is_act(mods[i + 1]) and m.out_channels % 16 == 0:
                    units.append(("convbiasact", m, mods[i + 1]))
                    i += 2
                    continue
                units.append(("convbias", m))
                i += 1
                continue
            raise NotImplementedError(f"bias-free convolution without BatchNorm is outside the HIP path: {m}")
        if isinstance(m, nn.MaxPool2d):
            if not (m.kernel_size in (2, (2, 2)) and m.stride in (2, (2, 2)) and m.padding in (0, (0, 0)) and not m.ceil_mode
                    and m.dilation in (1, (1, 1))):
                raise NotImplementedError(f"only MaxPool2d(2) has a HIP path: {m}")
            units.append(("maxpool2", m))
        elif isinstance(m, SPP):
            units.append(("spp", m))
        elif isinstance(m, DropBlock2d):
            units.append(("drop", m))
        elif hasattr(m, "forward_hip"):
            units.append(("block", m))
        else:
            raise NotImplementedError(f"{type(m).__name__} has no HIP execution path in run_conv_sequence")
        i += 1
    return units


def run_conv_sequence(seq, x, residual=None, out=None, padded_out=False, post_drop=None):
    """Execute the modules of a ``conv_sequence`` list / nn.Sequential on the HIP path, fusing every
    [Conv2d, BatchNorm2d, activation?, DropBlock2d?] run into one conv_bn_act call.  ``residual`` is added to the
    output of the LAST unit (DarkNet ResBlock: ``out = conv(x); out += identity``) and ``out`` (a concat slice)
    receives it; both need the last unit to be a fused one.  ``padded_out``: leave the channel padding of a final
    bias conv in place (the YOLO layer kernels read the padded NHWC logits directly)."""
    units = getattr(seq, "_hc_plan", None)
    if units is None or getattr(seq, "_hc_plan_len", -1) != len(seq):
        units = plan_conv_sequence(seq)
        try:
            seq._hc_plan, seq._hc_plan_len = units, len(seq)
        except AttributeError:
            pass
    last = len(units) - 1
    if (residual is not None or out is not None or post_drop is not None) and (last < 0 or units[last][0] != "fused"):
        raise NotImplementedError("residual / out / post_drop need the sequence to end with a fused conv unit")
    # residual == the sequence's own input and both ends are fused units of a stride-1 chain: couple them (ConvBnActFn `link`)
    src_link = sink_link = None
    if (residual is not None and residual is x and last >= 1 and units[0][0] == "fused" and x.requires_grad and torch.is_grad_enabled()
            and units[0][1].in_channels % 16 == 0 and all(u[0] == "fused" for u in units)):
        src_link = {"role": "source"}
        sink_link = {"role": "sink", "peer": src_link, "armed": True}
    for k, u in enumerate(units):
        kind = u[0]
        if kind == "fused":
            x = conv_bn_act(x, u[1], u[2], u[3], residual if k == last else None, u[4], out if k == last else None,
                            post_drop if k == last else None, sink_link if k == last else (src_link if k == 0 else None))
        elif kind == "convbias":
            x = conv_bias(x, u[1])
            if x.shape[1] != u[1].out_channels and not (padded_out and k == last):
                x = x[:, :u[1].out_channels]
        elif kind == "convbiasact":
            x = conv_bias_act(x, u[1], u[2])
        elif kind == "maxpool2":
            from ..ops.nhwc import maxpool2_cl
            x = maxpool2_cl(x)
        elif kind == "spp":
            x = u[1](x)
        elif kind == "drop":
            x = u[1](x)
        else:
            x = u[1].forward_hip(x)
    return x


def prepack_model_convs(model):
    """Step-level host work for models built from conv_sequence units: repack the bf16 forward / data-gradient weight
    images of every stale bias-free convolution with ONE hc_pack_conv_weights_multi launch (instead of two launches
    per convolution) and hand them to the per-conv caches."""
    import numpy as np
    convs = getattr(model, "_hc_convs", None)
    if convs is None:
        convs = [m for m in model.modules() if type(m) is nn.Conv2d and m.groups == 1 and m.bias is None
                 and m.in_channels % 16 == 0 and m.out_channels % 16 == 0 and m.weight.dtype == torch.float32]
        model._hc_convs = convs
    epoch = cv.weights_epoch()
    items, touched = [], []
    for conv in convs:
        w = conv.weight
        if not w.is_cuda:
            return
        st = getattr(conv, "_hc", None)
        if st is None:
            st = conv._hc = ConvState()
        key = ((w.data_ptr(), w._version), epoch)
        if getattr(st, "packed_key", None) == key and st.fwd_cache._key == key:
            continue
        Cout, Cin, KH, KW = w.shape
        T = KH * KW
        if getattr(st, "wf", None) is None or st.wf.device != w.device:
            st.wf = torch.empty((Cout, T, Cin), dtype=torch.bfloat16, device=w.device)
            st.wb = torch.empty((Cin, T, Cout), dtype=torch.bfloat16, device=w.device)
        items.append((w, st.wf, Cout, Cin, KH, KW, 0, 0, T))
        items.append((w, st.wb, Cout, Cin, KH, KW, 1, 0, T))
        touched.append((st, key))
    if not items:
        return
    sig = tuple((w.data_ptr(), dst.data_ptr()) for (w, dst, *_r) in items)
    cache = getattr(model, "_hc_pack_table", None)
    if cache is None or cache[0] != sig:
        arr = (_lib.PackItem * len(items))()
        mx = 0
        for a, (w, dst, Cout, Cin, KH, KW, mode, tap0, T) in zip(arr, items):
            a.w, a.dst, a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T = w.data_ptr(), dst.data_ptr(), Cout, Cin, KH, KW, mode, tap0, T
            mx = max(mx, w.numel())
        host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
        cache = (sig, host.to(items[0][0].device), len(items), mx)
        model._hc_pack_table = cache
    check(_lib.load().hc_pack_conv_weights_multi(cache[1].data_ptr(), cache[2], cache[3], stream()), "hc_pack_conv_weights_multi")
    for st, key in touched:
        st.packed_key = key
        st.fwd_cache._key, st.fwd_cache._val = key, st.wf
        st.bwd_cache._key, st.bwd_cache._val = key, st.wb
