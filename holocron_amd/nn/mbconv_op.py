"""Channel-padded conv -> BatchNorm2d -> activation units for the inverted-bottleneck stacks (ReXNet:
holocron/models/classification/rexnet.py:38-143) and the depthwise conv inside FReLU
(holocron/nn/modules/activation.py:58-82).

ReXNet's widths (27, 38, 50, ... 162, 228, ...) are not multiples of the 16 channels the MFMA gather-conv needs per
k-step, so activations travel between these units as NHWC bf16 with ``ceil16(C)`` channels per pixel; the pad channels
are exact zeros end to end (zero weight rows / columns in the packed weights, ``c_valid`` in the BatchNorm finalize
kernels gives them a = shift = 0, and SiLU / ReLU6 / identity map 0 to 0).  Parameters keep the reference's shapes.

  dense (1x1, or any kxk):  hc_conv_gather (+stats) -> hc_rep_bn_finalize -> hc_bn_act_apply (+ partial-width residual)
  depthwise 3x3:            hc_dw3x3_fwd   (+stats) -> same BN passes; backward hc_dw3x3_dgrad / hc_dw3x3_wgrad
  squeeze-excite:           hc_gap_fwd -> two tiny convs -> hc_se_scale_fwd (gate * z, ReLU6 fused)
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib
from .._lib import RepBnBwdDesc, RepBnDesc, check, ptr, stream
from ..ops import conv as cv
from .convbn_op import ConvState, act_code, as_cl_view, cl_ld
from .repblock_op import POOL


# Wide layers travel with their channel count padded to a multiple of 64 instead of 16 (from 160 channels on;
# 0 = always 16).  The gather-conv's k-step is the largest of 64 / 32 / 16 channels that divides the padded count: ReXNet's late
# widths (228, 300, 366, 432, 840, 906, 972 -> 240, 304, 368, 432, 848, 912, 976) are all = 16 mod 32, so every 1 x 1 convolution
# over them ran 15-61 sixteen-channel steps, each a DMA round trip for a quarter of the MFMA work of a 64-channel step.  Padding those
# tensors costs 2-7 % more bytes on layers that are 7 x 7 / 14 x 14 / 28 x 28 maps; rexnet1_0x step 22.45 -> 21.92 ms (same box;
# 32-channel padding 22.09, padding from 96 channels on 22.84: the 56 x 56 stages are HBM-bound and pay for their padding).  With the
# LDS-tiled depthwise kernels (64-channel slices: csrc/dwconv.hip) the threshold moved from 200 to 160: 162 -> 192 channels is three
# whole slices (22.36 -> 22.22 ms same-box).
_PAD_WIDE_FROM = 160
_PAD_WIDE_TO = 64


def ceil16(c):
    """Channels per pixel an activation of ``c`` logical channels travels with: a multiple of 16 (the smallest k-step of the MFMA
    gather-conv; the name is from when that was the only rule), of 64 from 160 channels on."""
    if _PAD_WIDE_FROM and c >= _PAD_WIDE_FROM:
        return (c + _PAD_WIDE_TO - 1) // _PAD_WIDE_TO * _PAD_WIDE_TO
    return (c + 15) // 16 * 16


def pad_channels(x, Cp):
    """Logical NCHW tensor -> NHWC bf16 with Cp >= C channels per pixel (zeros in the padding)."""
    _lib.require_gpu(x)
    N, Cc, H, W = x.shape
    if Cc == Cp:
        return cv.to_cl_bf16(x)
    if Cc > Cp:
        raise _lib.HipError("pad_channels: tensor has more channels than the padded width")
    out = torch.zeros((N, Cp, H, W), dtype=torch.bfloat16, device=x.device).contiguous(memory_format=torch.channels_last)
    if x.dtype == torch.float32 and x.is_contiguous():
        check(_lib.load().hc_nchw_to_nhwc_bf16(ptr(x), ptr(out), N, Cc, H, W, Cp, stream()), "hc_nchw_to_nhwc_bf16")
        return out
    out[:, :Cc] = x.to(torch.bfloat16)
    return out


class _PadChannelsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Cp):
        ctx.c = x.shape[1]
        ctx.dtype = x.dtype
        return pad_channels(x, Cp)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c].to(ctx.dtype), None


_REGISTRY = None      # the unit list of the model whose forward is running (ReXNet.forward -> padded_model_scope)


class padded_model_scope:
    """Around a model's forward: remembers every channel-padded conv unit that packs its weights inside (first step), and from the
    second step on repacks ALL of them - the optimizer has moved every weight - with one multi-tensor launch for the dense kernels and
    one for the depthwise ones instead of two launches per convolution (rexnet1_0x: 116 launches per step -> 2)."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        global _REGISTRY
        units = getattr(self.model, "_hcp_units", None)
        if units is None:
            units = self.model._hcp_units = {}
        self.prev, _REGISTRY = _REGISTRY, units
        if units:
            try:
                _prepack_units(self.model, units)
            except BaseException:
                _REGISTRY = self.prev          # a failed repack must not leave later models registering into this one's dict
                raise
        return self

    def __exit__(self, *exc):
        global _REGISTRY
        _REGISTRY = self.prev
        return False


def _prepack_units(model, units):
    import numpy as np
    epoch = cv.weights_epoch()
    stale = [(st, w, cinp, coutp, dwf) for (st, w, cinp, coutp, dwf) in units.values()
             if getattr(st, "pkey", None) != ((w.data_ptr(), w._version), epoch) and w.is_cuda and getattr(st, "pw", None) is not None
             and st.pw[0].device == w.device]
    if not stale:
        return
    lib = _lib.load()
    dense = [u for u in stale if not u[4]]
    dwise = [u for u in stale if u[4]]
    cache = getattr(model, "_hcp_tables", None)
    if cache is None:
        cache = model._hcp_tables = {}

    def upload(arr, dev):
        # the item table goes up by a pageable host-to-device copy: inside a stream capture that either fails or is recorded as a
        # memcpy node whose host buffer is gone at replay (the pack kernel would then read garbage pointers) - same rule as
        # repblock_op.launch_pack_items; a model's SECOND forward is the first to come here, so one eager step is not enough
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("padded conv weight images must be packed once outside stream capture (run two eager steps before "
                               "GraphedStep.capture(): the first registers the units, the second uploads their item tables)")
        return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    if dense:
        sig = tuple((w.data_ptr(), st.pw[0].data_ptr(), st.pw[1].data_ptr()) for (st, w, _a, _b, _c) in dense)
        ent = cache.get("dense")
        if ent is None or ent[0] != sig:
            arr = (_lib.PackItem * (2 * len(dense)))()
            mx = 0
            for i, (st, w, cinp, coutp, _d) in enumerate(dense):
                Cout, Cin_g, KH, KW = w.shape
                for a, (dst, mode, ld) in zip((arr[2 * i], arr[2 * i + 1]), ((st.pw[0], 0, cinp), (st.pw[1], 1, coutp))):
                    a.w, a.dst, a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T, a.ld = (w.data_ptr(), dst.data_ptr(), Cout, Cin_g, KH, KW, mode,
                                                                                        0, KH * KW, ld)
                mx = max(mx, w.numel())
            tab = upload(arr, dense[0][1].device)
            ent = cache["dense"] = (sig, tab, 2 * len(dense), mx)
        check(lib.hc_pack_conv_weights_multi(ent[1].data_ptr(), ent[2], ent[3], stream()), "hc_pack_conv_weights_multi")
    if dwise:
        sig = tuple((w.data_ptr(), st.pw[0].data_ptr(), st.pw[1].data_ptr()) for (st, w, _a, _b, _c) in dwise)
        ent = cache.get("dw")
        if ent is None or ent[0] != sig:
            arr = (_lib.DwPackItem * (2 * len(dwise)))()
            mx = 0
            for i, (st, w, _cinp, coutp, _d) in enumerate(dwise):
                for a, (dst, flip) in zip((arr[2 * i], arr[2 * i + 1]), ((st.pw[0], 0), (st.pw[1], 1))):
                    a.w, a.out, a.C, a.Cpad, a.flip = w.data_ptr(), dst.data_ptr(), w.shape[0], coutp, flip
                mx = max(mx, coutp)
            tab = upload(arr, dwise[0][1].device)
            ent = cache["dw"] = (sig, tab, 2 * len(dwise), mx)
        check(lib.hc_dw3x3_pack_multi(ent[1].data_ptr(), ent[2], ent[3], stream()), "hc_dw3x3_pack_multi")
    for (st, w, _a, _b, _c) in stale:
        st.pkey = ((w.data_ptr(), w._version), epoch)


def _pack_padded(st, w, Cin_p, Cout_p, depthwise):
    """Packed weights of a channel-padded conv, refreshed when the parameter changed."""
    key = ((w.data_ptr(), w._version), cv.weights_epoch())
    if _REGISTRY is not None and id(st) not in _REGISTRY:
        _REGISTRY[id(st)] = (st, w, Cin_p, Cout_p, depthwise)
    if getattr(st, "pkey", None) == key and st.pw[0].device == w.device:
        return st.pw
    lib = _lib.load()
    Cout, Cin_g, KH, KW = w.shape
    dev = w.device
    wc = w.detach()
    if depthwise:
        if getattr(st, "pw", None) is None or st.pw[0].device != dev:
            st.pw = (torch.empty((9, Cout_p), dtype=torch.float32, device=dev), torch.empty((9, Cout_p), dtype=torch.float32, device=dev))
        check(lib.hc_dw3x3_pack(ptr(wc), ptr(st.pw[0]), Cout, Cout_p, 0, stream()), "hc_dw3x3_pack")
        check(lib.hc_dw3x3_pack(ptr(wc), ptr(st.pw[1]), Cout, Cout_p, 1, stream()), "hc_dw3x3_pack")
    else:
        T = KH * KW
        if getattr(st, "pw", None) is None or st.pw[0].device != dev:
            st.pw = (torch.zeros((Cout_p, T, Cin_p), dtype=torch.bfloat16, device=dev),
                     torch.zeros((Cin_p, T, Cout_p), dtype=torch.bfloat16, device=dev))
            arr = (_lib.PackItem * 2)()
            for a, (dst, mode, ld) in zip(arr, ((st.pw[0], 0, Cin_p), (st.pw[1], 1, Cout_p))):
                a.w, a.dst, a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T, a.ld = wc.data_ptr(), dst.data_ptr(), Cout, Cin_g, KH, KW, mode, 0, T, ld
            import numpy as np
            st.ptable = (torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev), wc.data_ptr())
        if st.ptable[1] != wc.data_ptr():
            st.pw = None
            return _pack_padded(st, w, Cin_p, Cout_p, depthwise)
        check(lib.hc_pack_conv_weights_multi(st.ptable[0].data_ptr(), 2, w.numel(), stream()), "hc_pack_conv_weights_multi")
    st.pkey = key
    return st.pw


class PadConvBnActFn(torch.autograd.Function):
    """y = act(bn(conv(x))) [+ residual on the first res_C channels] on channel-padded NHWC bf16 activations."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, res, cbias, st, meta):
        lib = _lib.load()
        stride, pad, act, slope, bnbuf, eps, momentum, training, depthwise = meta
        Cout, Cin_g, KH, KW = w.shape
        Cin = Cout if depthwise else Cin_g
        Cin_p, Cout_p = ceil16(Cin), ceil16(Cout)
        N, Cx, H, W = x.shape
        dev = x.device
        if Cx != Cin_p or cl_ld(x) != Cin_p:
            raise _lib.HipError(f"padded conv unit expects a dense NHWC bf16 input with {Cin_p} channels, got {tuple(x.shape)}")
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise RuntimeError("conv_bn_act (HIP) expects contiguous fp32 conv weights")
        wf, wb = _pack_padded(st, w, Cin_p, Cout_p, depthwise)
        OH, OW = cv.conv_out_size(H, KH, stride, pad), cv.conv_out_size(W, KW, stride, pad)
        npix = N * OH * OW
        y = cv.empty_cl(N, Cout_p, OH, OW, dev)
        stats = POOL.take((_lib.stat_replicas(), 2, Cout_p), dev) if training else None
        if depthwise:
            with cv.profiled("dwconv", 18.0 * y.numel(), (x.numel() + y.numel()) * 2.0):
                check(lib.hc_dw3x3_fwd(ptr(x), ptr(wf), ptr(y), ptr(stats), N, H, W, Cout_p, stride, stream()), "hc_dw3x3_fwd")
        else:
            key = ("f", N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.fwd_desc(N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
            cv.launch_conv(st.desc[key], x, wf, y, stats=stats, flops=2.0 * npix * Cout * Cin * KH * KW)
        coef = torch.empty((4, Cout_p), dtype=torch.float32, device=dev)
        save = torch.empty((6, Cout_p), dtype=torch.float32, device=dev)
        d = RepBnDesc()
        rm, rv, nbt = bnbuf
        d.stats[0], d.gamma[0], d.beta[0] = ptr(stats), ptr(gamma), ptr(beta)
        d.running_mean[0], d.running_var[0], d.num_batches_tracked[0] = ptr(rm), ptr(rv), ptr(nbt)
        d.coef, d.save, d.C, d.count = ptr(coef), ptr(save), Cout_p, npix
        d.eps, d.momentum, d.training, d.c_valid = eps, momentum, 1 if training else 0, Cout
        check(lib.hc_rep_bn_finalize(C.byref(d), stream()), "hc_rep_bn_finalize")
        if cbias is not None:
            # a conv bias in front of BatchNorm (FReLU, activation.py:73): batch statistics cancel it exactly, it only
            # moves the running mean; with running statistics it shifts the affine
            if training:
                if rm is not None:
                    rm.add_(cbias.detach(), alpha=momentum)
            else:
                coef[3, :Cout] += coef[0, :Cout] * cbias.detach()
        res_C = 0
        if res is not None:
            res_C = res.shape[1]
            if cl_ld(res) != res_C or res_C > Cout_p or tuple(res.shape[2:]) != (OH, OW):
                raise _lib.HipError("residual must be a dense NHWC bf16 tensor with at most the output's channels")
        out = cv.empty_cl(N, Cout_p, OH, OW, dev)
        with cv.profiled("bn_elementwise", 0.0, npix * Cout_p * 2.0 * (3 if res is not None else 2)):
            check(lib.hc_bn_act_apply(ptr(y), ptr(coef), ptr(res), res_C, None, None, ptr(out), Cout_p, npix, Cout_p, act, slope, stream()),
                  "hc_bn_act_apply")
        ctx.st, ctx.meta2 = st, (stride, pad, act, slope, training, depthwise, res_C)
        ctx.geom = (N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, OH, OW)
        ctx.red, ctx.red_gen = POOL.take_for_backward((_lib.stat_replicas(), 4, Cout_p), dev) if training else (None, -1)
        ctx.has_cbias = cbias is not None
        ctx.save_for_backward(x, y, coef, save, gamma, w)
        return out

    @staticmethod
    def backward(ctx, g):
        stride, pad, act, slope, training, depthwise, res_C = ctx.meta2
        lib = _lib.load()
        st = ctx.st
        x, y, coef, save, gamma, w = ctx.saved_tensors
        N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, OH, OW = ctx.geom
        dev = g.device
        g, g_ld = as_cl_view(g)
        npix = N * OH * OW
        red = POOL.claim(ctx.red, ctx.red_gen, (_lib.stat_replicas(), 4, Cout_p), dev)   # stale after another forward's POOL.begin()
        ctx.red = None
        with cv.profiled("bn_elementwise", 0.0, npix * Cout_p * 2.0 * 2):
            check(lib.hc_bn_act_bwd_reduce(ptr(g), g_ld, ptr(y), ptr(coef), None, None, ptr(red), npix, Cout_p, act, slope, stream()),
                  "hc_bn_act_bwd_reduce")
        dgam = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbet = torch.empty((Cout,), dtype=torch.float32, device=dev)
        bcoef = torch.empty((9, Cout_p), dtype=torch.float32, device=dev)
        d = RepBnBwdDesc()
        d.red, d.save, d.bcoef = ptr(red), ptr(save), ptr(bcoef)
        d.gamma[0], d.dgamma[0], d.dbeta[0] = ptr(gamma), ptr(dgam), ptr(dbet)
        d.C, d.count, d.has_identity, d.accumulate, d.c_valid = Cout_p, npix, 0, 0, Cout
        d.frozen = 0 if training else 1               # eval mode / freeze_bn: running statistics, dy = a * dz
        check(lib.hc_rep_bn_bwd_finalize(C.byref(d), stream()), "hc_rep_bn_bwd_finalize")
        dy = torch.empty_like(y)
        with cv.profiled("bn_elementwise", 0.0, npix * Cout_p * 2.0 * 3):
            check(lib.hc_bn_act_bwd_apply(ptr(g), g_ld, ptr(y), ptr(coef), ptr(bcoef), None, None, ptr(dy), npix, Cout_p, act, slope,
                                          stream()), "hc_bn_act_bwd_apply")
        wf, wb = st.pw
        dx = None
        if depthwise:
            if ctx.needs_input_grad[0]:
                dx = cv.empty_cl(N, Cin_p, H, W, dev)
                with cv.profiled("dwconv", 18.0 * dy.numel(), (dx.numel() + dy.numel()) * 2.0):
                    check(lib.hc_dw3x3_dgrad(ptr(dy), ptr(wf), ptr(wb), ptr(dx), N, H, W, Cout_p, stride, stream()), "hc_dw3x3_dgrad")
            ws = torch.empty((lib.hc_dw3x3_wgrad_ws_bytes(Cout_p) // 4,), dtype=torch.float32, device=dev)
            dw = torch.empty_like(w, dtype=torch.float32)
            with cv.profiled("dwconv", 18.0 * dy.numel(), (x.numel() + dy.numel()) * 2.0):
                check(lib.hc_dw3x3_wgrad(ptr(x), ptr(dy), ptr(ws), ptr(dw), N, H, W, Cout_p, Cout, stride, 0, stream()), "hc_dw3x3_wgrad")
        else:
            if ctx.needs_input_grad[0]:
                key = ("d", N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
                if key not in st.desc:
                    st.desc[key] = cv.dgrad_desc(N, Cin_p, H, W, Cout_p, [(KH, KW, pad, 0, 0)], stride)
                dx = cv.empty_cl(N, Cin_p, H, W, dev)
                cv.launch_conv(st.desc[key], dy, wb, dx)
            with cv.side_stream_for_wgrad((w,), (x, dy)) as side:
                # the reduce kernel of the weight gradient drops the padding rows / columns itself: no slicing copy afterwards
                dw = cv.conv_wgrad(x, dy, Cin_p, Cout_p, KH, KW, stride, pad, flops=2.0 * npix * Cout * Cin * KH * KW,
                                   valid=None if (Cin_p == Cin and Cout_p == Cout) else (Cout, Cin))
                side.produced(dw)
        gres = None
        if res_C:
            gres = g if res_C == Cout_p and g_ld == Cout_p else g[:, :res_C]
        dcb = torch.zeros((Cout,), dtype=torch.float32, device=dev) if ctx.has_cbias else None
        return dx, dw, dgam, dbet, gres, dcb, None, None


def padded_unit_supported(conv, bn, act):
    if not (type(conv) is nn.Conv2d and isinstance(bn, nn.BatchNorm2d) and conv.dilation == (1, 1)
            and conv.padding_mode == "zeros" and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1] and conv.stride[0] in (1, 2) and act_code(act) is not None):
        return False
    if conv.groups == 1:
        return conv.kernel_size[0] * conv.kernel_size[1] <= _lib.HC_MAX_TAPS
    return (conv.groups == conv.in_channels == conv.out_channels and conv.kernel_size == (3, 3) and conv.padding == (1, 1))


def padded_conv_bn_act(x, conv, bn, act=None, residual=None):
    """One [Conv2d, BatchNorm2d, act?] unit on channel-padded activations (dense or depthwise 3x3)."""
    if not padded_unit_supported(conv, bn, act):
        raise NotImplementedError(f"conv/bn/act unit outside the HIP path: {conv}, {bn}, {act}")
    st = getattr(conv, "_hcp", None)
    if st is None:
        st = conv._hcp = ConvState()
    code, slope = act_code(act)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    depthwise = conv.groups != 1
    meta = (conv.stride[0], conv.padding[0], code, slope, (bn.running_mean, bn.running_var, bn.num_batches_tracked), bn.eps,
            momentum, bn.training, depthwise)
    cin_p = ceil16(conv.in_channels)
    if x.shape[1] != cin_p or cl_ld(x) != cin_p:
        x = _PadChannelsFn.apply(x, cin_p)
    return PadConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, residual, conv.bias, st, meta)


def _pooled_to_cl_bf16(pooled, N, Cp):
    """fp32 [N, Cp] pooled values -> the bf16 NHWC [N, Cp, 1, 1] leaf the gate convs read: ONE cast-copy launch."""
    p_in = torch.empty((N, Cp, 1, 1), dtype=torch.bfloat16, device=pooled.device).contiguous(memory_format=torch.channels_last)
    p_in.view(N, Cp).copy_(pooled)
    return p_in.requires_grad_(True)


class SeGateFn(torch.autograd.Function):
    """ReXNet's squeeze-excite block followed by the block's ReLU6 (rexnet.py:63-66,126-129):
    ``out = relu6(z * sigmoid(mlp(mean_hw(z))))``.  The two tiny convs of ``mlp`` run through the regular units on the
    [N, C, 1, 1] pooled tensor inside a private autograd graph that this node differentiates in its own backward, so
    that the gradient of z is produced by ONE pass (gate path + pooled path) instead of two tensors that autograd adds."""

    @staticmethod
    def forward(ctx, z, mlp, act):
        lib = _lib.load()
        N, Cp, H, W = z.shape
        pooled = torch.empty((N, Cp), dtype=torch.float32, device=z.device)
        check(lib.hc_gap_fwd(ptr(z), ptr(pooled), N, H * W, Cp, stream()), "hc_gap_fwd")
        with torch.enable_grad():
            p_in = _pooled_to_cl_bf16(pooled, N, Cp)
            logits = mlp(p_in)                                   # [N, Cp, 1, 1] bf16 gate logits (conv + bias)
        lg = logits.detach()
        if lg.shape[1] != Cp or lg.dtype != torch.bfloat16:
            raise _lib.HipError("squeeze-excite gate must return bf16 logits with the padded channel count")
        lg = lg.reshape(N, Cp).contiguous()
        out = torch.empty_like(z)
        check(lib.hc_se_scale_fwd(ptr(z), ptr(lg), ptr(out), N, H * W, Cp, act, stream()), "hc_se_scale_fwd")
        ctx.graph = (p_in, logits)
        ctx.act = act
        ctx.save_for_backward(z, lg)
        return out

    @staticmethod
    def backward(ctx, g):
        z, lg = ctx.saved_tensors
        p_in, logits = ctx.graph
        ctx.graph = None
        lib = _lib.load()
        N, Cp, H, W = z.shape
        g = cv.to_cl_bf16(g)
        if cl_ld(g) != Cp:
            g = g.contiguous(memory_format=torch.channels_last)
        dgate = torch.empty((N, Cp), dtype=torch.float32, device=z.device)
        dl = torch.empty((N, Cp), dtype=torch.bfloat16, device=z.device)
        check(lib.hc_se_scale_bwd_gate(ptr(g), ptr(z), ptr(lg), ptr(dgate), ptr(dl), N, H * W, Cp, ctx.act, stream()),
              "hc_se_scale_bwd_gate")
        # parameter gradients of the two tiny convs accumulate into .grad here; dpool comes back on p_in
        torch.autograd.backward(logits, dl.view(N, Cp, 1, 1).to(logits.dtype))
        dpool = p_in.grad.reshape(N, Cp).float().contiguous()
        dz = torch.empty_like(z)
        check(lib.hc_se_scale_bwd_apply(ptr(g), ptr(z), ptr(lg), ptr(dpool), ptr(dz), N, H * W, Cp, ctx.act, stream()),
              "hc_se_scale_bwd_apply")
        return dz, None, None


# (round 3 ran the squeeze-excite MLP through the generic conv units inside SeGateFn; its A/B switch is retired)
_SE_FUSED = True


def se_mlp_fusable(conv1, bn, act, conv2):
    """The layouts hc_se_mlp_fwd / bwd cover (rexnet.py:38-66 as the reference builds it): 1x1 conv without bias, a training-mode
    BatchNorm2d with running statistics and a float momentum, ReLU / ReLU6 / no activation, 1x1 conv, fp32 parameters."""
    ok = (_SE_FUSED and type(conv1) is nn.Conv2d and type(conv2) is nn.Conv2d and type(bn) is nn.BatchNorm2d
          and conv1.kernel_size == (1, 1) and conv2.kernel_size == (1, 1) and conv1.bias is None and conv1.groups == 1
          and conv2.groups == 1 and conv1.stride == (1, 1) and conv2.stride == (1, 1) and conv1.padding == (0, 0)
          and conv2.padding == (0, 0) and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
          and (act is None or type(act) in (nn.ReLU, nn.ReLU6)) and conv1.out_channels <= 128
          and conv1.out_channels == conv2.in_channels and conv2.out_channels == conv1.in_channels)
    return ok and all(p.dtype == torch.float32 for p in (conv1.weight, conv2.weight, bn.weight, bn.bias))


class SeGateFusedFn(torch.autograd.Function):
    """SeGateFn with the MLP as hc_se_mlp_fwd / hc_se_mlp_bwd (csrc/se_mlp.hip): 2 + 4 launches on the pooled vectors instead of the
    ~28 of the generic units, no private autograd graph.  ``out = act(z * sigmoid(mlp(mean_hw(z))))``."""

    @staticmethod
    def forward(ctx, z, w1, gamma, beta, w2, b2, bn_bufs, cfg):
        lib = _lib.load()
        N, Cp, H, W = z.shape
        act, mlp_act, eps, momentum = cfg
        R, Cc = w1.shape[0], w1.shape[1]
        dev = z.device
        pooled = torch.empty((N, Cp), dtype=torch.float32, device=dev)
        check(lib.hc_gap_fwd(ptr(z), ptr(pooled), N, H * W, Cp, stream()), "hc_gap_fwd")
        nf = int(lib.hc_se_mlp_part_floats(N, R))
        # one allocation for what the forward hands to the backward: h1 [N][R], part, stat [2][R]
        save = torch.empty((N * R + nf + 2 * R,), dtype=torch.float32, device=dev)
        lg = torch.empty((N, Cp), dtype=torch.bfloat16, device=dev)
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        d = _lib.SeMlpDesc()
        d.pooled, d.w1, d.gamma, d.beta, d.w2 = ptr(pooled), ptr(w1c), ptr(gamma.detach()), ptr(beta.detach()), ptr(w2c)
        d.b2 = ptr(b2.detach()) if b2 is not None else None
        rm, rv, nbt = bn_bufs
        d.running_mean, d.running_var, d.num_batches_tracked = ptr(rm), ptr(rv), ptr(nbt)
        d.h1, d.part, d.stat = save.data_ptr(), save.data_ptr() + 4 * N * R, save.data_ptr() + 4 * (N * R + nf)
        d.logits = ptr(lg)
        d.N, d.C, d.Cp, d.R, d.act, d.eps, d.momentum = N, Cc, Cp, R, mlp_act, eps, momentum
        check(lib.hc_se_mlp_fwd(C.byref(d), stream()), "hc_se_mlp_fwd")
        out = torch.empty_like(z)
        check(lib.hc_se_scale_fwd(ptr(z), ptr(lg), ptr(out), N, H * W, Cp, act, stream()), "hc_se_scale_fwd")
        ctx.cfg = (act, mlp_act, eps, momentum, nf, b2 is not None)
        ctx.save_for_backward(z, lg, pooled, save, w1c, gamma, beta, w2c)
        return out

    @staticmethod
    def backward(ctx, g):
        z, lg, pooled, save, w1c, gamma, beta, w2c = ctx.saved_tensors
        act, mlp_act, eps, momentum, nf, has_b2 = ctx.cfg
        lib = _lib.load()
        N, Cp, H, W = z.shape
        R, Cc = w1c.shape[0], w1c.shape[1]
        dev = z.device
        g = cv.to_cl_bf16(g)
        if cl_ld(g) != Cp:
            g = g.contiguous(memory_format=torch.channels_last)
        dgate = torch.empty((N, Cp), dtype=torch.float32, device=dev)
        dl = torch.empty((N, Cp), dtype=torch.bfloat16, device=dev)
        check(lib.hc_se_scale_bwd_gate(ptr(g), ptr(z), ptr(lg), ptr(dgate), ptr(dl), N, H * W, Cp, act, stream()),
              "hc_se_scale_bwd_gate")
        scratch = torch.empty((N * R + nf,), dtype=torch.float32, device=dev)
        dpool = torch.empty((N, Cp), dtype=torch.float32, device=dev)
        dw1, dw2 = torch.empty_like(w1c), torch.empty_like(w2c)
        dgb = torch.empty((2, R), dtype=torch.float32, device=dev)
        db2 = torch.empty((Cc,), dtype=torch.float32, device=dev) if has_b2 else None
        d = _lib.SeMlpDesc()
        d.pooled, d.w1, d.gamma, d.beta, d.w2 = ptr(pooled), ptr(w1c), ptr(gamma.detach()), ptr(beta.detach()), ptr(w2c)
        d.h1, d.part, d.stat = save.data_ptr(), save.data_ptr() + 4 * N * R, save.data_ptr() + 4 * (N * R + nf)
        d.dl, d.g, d.part2 = ptr(dl), scratch.data_ptr(), scratch.data_ptr() + 4 * N * R
        d.dpool, d.dw1, d.dw2 = ptr(dpool), ptr(dw1), ptr(dw2)
        d.dgamma, d.dbeta = dgb.data_ptr(), dgb.data_ptr() + 4 * R
        d.db2 = ptr(db2) if has_b2 else None
        d.N, d.C, d.Cp, d.R, d.act, d.eps, d.momentum = N, Cc, Cp, R, mlp_act, eps, momentum
        check(lib.hc_se_mlp_bwd(C.byref(d), stream()), "hc_se_mlp_bwd")
        dz = torch.empty_like(z)
        check(lib.hc_se_scale_bwd_apply(ptr(g), ptr(z), ptr(lg), ptr(dpool), ptr(dz), N, H * W, Cp, act, stream()),
              "hc_se_scale_bwd_apply")
        return dz, dw1, dgb[0], dgb[1], dw2, db2, None, None


def se_gate_fused(z, conv1, bn, mlp_act_module, conv2, act):
    """``act(z * sigmoid(conv2(mlp_act(bn(conv1(mean_hw(z)))))))`` on a channel-padded NHWC bf16 tensor (the caller checked
    se_mlp_fusable)."""
    R, Cc = conv1.out_channels, conv1.in_channels
    mlp_act = 0 if mlp_act_module is None else (1 if type(mlp_act_module) is nn.ReLU else 6)
    cfg = (act, mlp_act, float(bn.eps), float(bn.momentum))
    return SeGateFusedFn.apply(z, conv1.weight.view(R, Cc), bn.weight, bn.bias, conv2.weight.view(Cc, R), conv2.bias,
                               (bn.running_mean, bn.running_var, bn.num_batches_tracked), cfg)


class _MaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = cv.to_cl_bf16(a), cv.to_cl_bf16(b)
        out = torch.empty_like(a)
        check(_lib.load().hc_max_fwd(ptr(a), ptr(b), ptr(out), a.numel(), stream()), "hc_max_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = cv.to_cl_bf16(g)
        if cl_ld(g) != g.shape[1]:
            g = g.contiguous(memory_format=torch.channels_last)
        da, db = torch.empty_like(a), torch.empty_like(b)
        check(_lib.load().hc_max_bwd(ptr(a), ptr(b), ptr(g), ptr(da), ptr(db), a.numel(), stream()), "hc_max_bwd")
        return da, db


def elementwise_max(a, b):
    """max(a, b) of two NHWC bf16 tensors of the same padded shape, with torch.max's tie-splitting gradient."""
    if a.shape != b.shape or a.numel() % 8:
        raise _lib.HipError("elementwise_max needs equal shapes with a multiple of 8 elements")
    return _MaxFn.apply(a, b)


class PadConvBiasFn(torch.autograd.Function):
    """conv(x) + bias without normalisation on channel-padded activations (the squeeze-excite gate conv,
    rexnet.py:59: ``conv_sequence(channels // se_ratio, channels, nn.Sigmoid(), None, ...)``; the sigmoid is applied by
    the gate kernel)."""

    @staticmethod
    def forward(ctx, x, w, bias, st, meta):
        stride, pad = meta
        Cout, Cin, KH, KW = w.shape
        Cin_p, Cout_p = ceil16(Cin), ceil16(Cout)
        N, Cx, H, W = x.shape
        dev = x.device
        if Cx != Cin_p or cl_ld(x) != Cin_p:
            raise _lib.HipError(f"padded conv expects a dense NHWC bf16 input with {Cin_p} channels, got {tuple(x.shape)}")
        wf, wb = _pack_padded(st, w, Cin_p, Cout_p, False)
        bp = getattr(st, "bias_pad", None)          # zero-padded bias: the padding stays zero, only the live part is refreshed
        if bp is None or bp.device != dev or bp.numel() != Cout_p:
            bp = st.bias_pad = torch.zeros((Cout_p,), dtype=torch.float32, device=dev)
        if bias is not None:
            bp[:Cout].copy_(bias.detach())
        key = ("fb", N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
        if key not in st.desc:
            st.desc[key] = cv.fwd_desc(N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
        fd = st.desc[key]
        y = cv.empty_cl(N, Cout_p, fd.OH, fd.OW, dev)
        cv.launch_conv(fd, x, wf, y, bias=bp, act=0, flops=2.0 * N * fd.OH * fd.OW * Cout * Cin * KH * KW)
        ctx.st, ctx.meta2 = st, (stride, pad, bias is not None)
        ctx.geom = (N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, fd.OH, fd.OW)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        st = ctx.st
        stride, pad, has_bias = ctx.meta2
        N, Cin, Cin_p, H, W, Cout, Cout_p, KH, KW, OH, OW = ctx.geom
        dev = g.device
        dy = cv.to_cl_bf16(g)
        if cl_ld(dy) != Cout_p:
            dy = dy.contiguous(memory_format=torch.channels_last)
        lib = _lib.load()
        db = None
        if has_bias:
            if OH * OW <= 64:       # the squeeze-excite gate convs ([N, C, 1, 1]): one fp32-accumulating reduction, no scratch
                db = dy.permute(0, 2, 3, 1).reshape(-1, Cout_p).sum(0, dtype=torch.float32)[:Cout]
            else:
                stats = torch.zeros((_lib.stat_replicas(), 2, Cout_p), dtype=torch.float32, device=dev)
                check(lib.hc_channel_stats(ptr(dy), ptr(stats), N * OH * OW, Cout_p, stream()), "hc_channel_stats")
                db = stats[:, 0].sum(0)[:Cout]
        dx = None
        if ctx.needs_input_grad[0]:
            key = ("db", N, Cin_p, H, W, Cout_p, KH, KW, stride, pad)
            if key not in st.desc:
                st.desc[key] = cv.dgrad_desc(N, Cin_p, H, W, Cout_p, [(KH, KW, pad, 0, 0)], stride)
            dx = cv.empty_cl(N, Cin_p, H, W, dev)
            cv.launch_conv(st.desc[key], dy, st.pw[1], dx)
        with cv.side_stream_for_wgrad((w,), (x, dy)) as side:
            dw = cv.conv_wgrad(x, dy, Cin_p, Cout_p, KH, KW, stride, pad, flops=2.0 * N * OH * OW * Cout * Cin * KH * KW,
                               valid=None if (Cin_p == Cin and Cout_p == Cout) else (Cout, Cin))
            side.produced(dw)
        return dx, dw, db, None, None


def padded_conv_bias(x, conv):
    """``conv(x)`` for a bias-carrying dense nn.Conv2d on channel-padded activations; returns the padded output."""
    if not (type(conv) is nn.Conv2d and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == "zeros"
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.padding[0] == conv.padding[1]):
        raise NotImplementedError("padded_conv_bias: unsupported convolution geometry for the HIP path")
    st = getattr(conv, "_hcp", None)
    if st is None:
        st = conv._hcp = ConvState()
    cin_p = ceil16(conv.in_channels)
    if x.shape[1] != cin_p or cl_ld(x) != cin_p:
        x = _PadChannelsFn.apply(x, cin_p)
    return PadConvBiasFn.apply(x, conv.weight, conv.bias, st, (conv.stride[0], conv.padding[0]))


class SlimGateFn(torch.autograd.Function):
    """SlimConv2d's channel gate and fold (nn/modules/conv.py:352-364): returns the two half-width tensors
    ``fold(x * w)`` and ``fold(x * flip(w))`` with ``w = sigmoid(mlp(mean_hw(x)))`` as channel-padded NHWC bf16.  Like
    SeGateFn the tiny gate network runs in a private autograd graph differentiated inside this node's backward."""

    @staticmethod
    def forward(ctx, x, mlp, channels):
        lib = _lib.load()
        N, Cp, H, W = x.shape
        h = channels // 2
        hp = ceil16(h)
        pooled = torch.empty((N, Cp), dtype=torch.float32, device=x.device)
        check(lib.hc_gap_fwd(ptr(x), ptr(pooled), N, H * W, Cp, stream()), "hc_gap_fwd")
        with torch.enable_grad():
            p_in = _pooled_to_cl_bf16(pooled, N, Cp)
            logits = mlp(p_in)
        lg = logits.detach().reshape(N, -1).contiguous()
        if lg.dtype != torch.bfloat16 or lg.shape[1] < channels:
            raise _lib.HipError("SlimConv2d gate must return bf16 logits for every channel")
        top, bot = cv.empty_cl(N, hp, H, W, x.device), cv.empty_cl(N, hp, H, W, x.device)
        check(lib.hc_slim_fold_fwd(ptr(x), Cp, ptr(lg), lg.shape[1], ptr(top), ptr(bot), hp, N, H * W, channels, stream()),
              "hc_slim_fold_fwd")
        ctx.graph = (p_in, logits)
        ctx.channels = channels
        ctx.save_for_backward(x, lg)
        return top, bot

    @staticmethod
    def backward(ctx, gtop, gbot):
        x, lg = ctx.saved_tensors
        p_in, logits = ctx.graph
        ctx.graph = None
        lib = _lib.load()
        N, Cp, H, W = x.shape
        Cc = ctx.channels
        hp = ceil16(Cc // 2)

        def dense(g):
            g = cv.to_cl_bf16(g)
            return g if cl_ld(g) == hp else g.contiguous(memory_format=torch.channels_last)
        gtop, gbot = dense(gtop), dense(gbot)
        dl = torch.empty_like(lg)
        check(lib.hc_slim_fold_bwd_gate(ptr(x), Cp, ptr(lg), lg.shape[1], ptr(gtop), ptr(gbot), hp, ptr(dl), N, H * W, Cc, stream()),
              "hc_slim_fold_bwd_gate")
        torch.autograd.backward(logits, dl.view(logits.shape).to(logits.dtype))
        dpool = p_in.grad.reshape(N, Cp).float().contiguous()
        dx = torch.empty_like(x)
        check(lib.hc_slim_fold_bwd_apply(ptr(lg), lg.shape[1], ptr(gtop), ptr(gbot), hp, ptr(dpool), Cp, ptr(dx), Cp, N, H * W, Cc,
                                         stream()), "hc_slim_fold_bwd_apply")
        return dx, None, None
