"""MobileOne over-parameterised blocks on the MI355X kernels (reference: holocron/models/classification/mobileone.py).

``DepthConvBlock.forward`` (:66-67) and ``PointConvBlock.forward`` (:120-121) are ``sum(mod(x) for mod in self)`` over
parallel [conv, BatchNorm2d] branches plus (stride 1 / equal widths) a bare BatchNorm2d of the input, followed in
``MobileOneBlock`` (:169-174) by the activation.  Here one block is one autograd node:

  depth block   K + 1 ``hc_dw3x3_fwd`` planes (the depthwise 1x1 rides as a centre-only 3x3) with the BatchNorm statistics
                in their epilogues -> ``hc_msbn_finalize`` -> ``hc_msbn_apply`` (all BatchNorm affines, the sum and the ReLU
                in one pass, statistics of the output for the next block's identity BatchNorm);
                backward ``hc_msbn_bwd_reduce`` -> ``hc_msbn_bwd_finalize`` -> ``hc_msbn_bwd_apply`` -> ``hc_dwrep_dgrad``
                (all planes + the identity gradient in one pass) + ``hc_dw3x3_wgrad`` per plane
  point block   the K dense 1x1 convolutions as ONE ``hc_conv_gather`` with K * Cout stacked output channels -> the same
                msbn passes reading channel slices of the stacked tensor; backward: one stacked data-gradient conv (the
                identity gradient rides as its residual) and one stacked ``hc_conv_wgrad``

Activations are NHWC bf16 with ``ceil16(C)`` channels per pixel (zeros in the padding), like the ReXNet units.
"""
import ctypes as C

import torch
import torch.nn.functional as TF

from .. import _lib
from .._lib import MsbnDesc, MsbnIo, check, ptr, stream
from ..ops import conv as cv
from .convbn_op import as_cl_view, cl_ld
from .mbconv_op import ceil16
from .repblock_op import POOL, _stats_of


def _R():
    return _lib.stat_replicas()
MAXB = _lib.HC_MSBN_MAX_BRANCHES


class BlockState:
    """Host state of one block: packed weights keyed by the parameters' versions, conv descriptors."""

    def __init__(self):
        self.pkey = None
        self.pw = None
        self.desc = {}
        self.ptable = None


def _bn_info(bn):
    return (bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.eps), 0.1 if bn.momentum is None else float(bn.momentum))


def _fill_desc(d, infos, gammas, betas, stats_ptrs, stats_lds, Cp, c_valid, count, training):
    d.B, d.C, d.c_valid, d.count, d.training = len(infos), Cp, c_valid, count, 1 if training else 0
    for b, (info, g, bt) in enumerate(zip(infos, gammas, betas)):
        rm, rv, nbt, eps, mom = info
        br = d.br[b]
        br.stats = stats_ptrs[b] if training else None
        br.stats_ld = stats_lds[b]
        br.gamma, br.beta = ptr(g), ptr(bt)
        br.running_mean, br.running_var, br.num_batches_tracked = ptr(rm), ptr(rv), ptr(nbt)
        br.eps, br.momentum = eps, mom


def _io(srcs, lds, npix, Cp):
    io = MsbnIo()
    io.B, io.C, io.npix = len(srcs), Cp, npix
    for b, (p, ld) in enumerate(zip(srcs, lds)):
        io.y[b], io.ld[b] = p, ld
    return io


def _msbn_backward(ctx, lib, g, out, save, io, B, gammas, Cp, c_valid, npix, dy_ptrs, dy_lds, dev):
    """reduce -> finalize -> apply; returns the per-branch [B, 2, c_valid] parameter gradients."""
    g, g_ld = as_cl_view(g)
    red = POOL.claim(ctx.red, getattr(ctx, "red_gen", -1), (_R(), B + 1, Cp), dev)   # stale after another forward's POOL.begin()
    ctx.red = None
    check(lib.hc_msbn_bwd_reduce(C.byref(io), ptr(g), g_ld, ptr(out), ptr(red), ctx.act, stream()), "hc_msbn_bwd_reduce")
    pgrad = torch.empty((B, 2, max(c_valid, 1)), dtype=torch.float32, device=dev)
    bcoef = torch.empty((B, 3, Cp), dtype=torch.float32, device=dev)
    d = MsbnDesc()
    d.B, d.C, d.c_valid, d.count, d.training, d.accumulate = B, Cp, c_valid, npix, 1 if ctx.training else 0, 0
    d.red, d.save, d.bcoef = ptr(red), ptr(save), ptr(bcoef)
    for b in range(B):
        d.br[b].gamma = ptr(gammas[b])
        d.br[b].dgamma = pgrad.data_ptr() + (b * 2 + 0) * pgrad.shape[2] * 4
        d.br[b].dbeta = pgrad.data_ptr() + (b * 2 + 1) * pgrad.shape[2] * 4
    check(lib.hc_msbn_bwd_finalize(C.byref(d), stream()), "hc_msbn_bwd_finalize")
    for b in range(B):
        io.dy[b], io.dld[b] = dy_ptrs[b], dy_lds[b]
    check(lib.hc_msbn_bwd_apply(C.byref(io), ptr(g), g_ld, ptr(out), ptr(bcoef), ctx.act, stream()), "hc_msbn_bwd_apply")
    return pgrad


class DepthRepFn(torch.autograd.Function):
    """act(sum_b BN_b(dwconv_b(x)) [+ BN_id(x)]);  params = [g_id, b_id]? + (w, gamma, beta) per conv branch, the 1x1 first."""

    @staticmethod
    def forward(ctx, x, st, meta, *params):
        lib = _lib.load()
        stride, has_id, infos, training, act, Cc = meta
        N, Cp, H, W = x.shape
        dev = x.device
        if Cp != ceil16(Cc) or cl_ld(x) != Cp:
            raise _lib.HipError(f"depth block expects a dense NHWC bf16 input with {ceil16(Cc)} channels, got {tuple(x.shape)}")
        off = 2 if has_id else 0
        ws = params[off::3]
        gammas = list(params[off + 1::3]) + ([params[0]] if has_id else [])
        betas = list(params[off + 2::3]) + ([params[1]] if has_id else [])
        conv_infos = infos[1:] if has_id else infos
        all_infos = list(conv_infos) + ([infos[0]] if has_id else [])
        P = len(ws)
        B = P + (1 if has_id else 0)
        if B > MAXB:
            raise NotImplementedError(f"at most {MAXB} parallel branches per block")
        # ---- packed taps, refreshed when a parameter changed
        key = tuple((w.data_ptr(), w._version) for w in ws) + (cv.weights_epoch(),)
        if st.pkey != key or st.pw is None or st.pw.device != dev:
            if st.pw is None or st.pw.device != dev or st.pw.shape != (P, 9, Cp):
                st.pw = torch.empty((P, 9, Cp), dtype=torch.float32, device=dev)
            for b, w in enumerate(ws):
                w3 = w.detach()
                if w3.shape[-1] == 1:                      # depthwise 1x1 -> centre tap of a 3x3
                    w3 = TF.pad(w3, (1, 1, 1, 1))
                w3 = w3.float().contiguous()
                check(lib.hc_dw3x3_pack(ptr(w3), st.pw[b].data_ptr(), Cc, Cp, 0, stream()), "hc_dw3x3_pack")
            st.pkey = key
        OH, OW = cv.conv_out_size(H, 3, stride, 1), cv.conv_out_size(W, 3, stride, 1)
        npix = N * OH * OW
        planes = torch.empty((P, N, OH, OW, Cp), dtype=torch.bfloat16, device=dev)
        psz = npix * Cp * 2
        stats = POOL.take((P, _R(), 2, Cp), dev) if training else None
        for b in range(P):
            check(lib.hc_dw3x3_fwd(ptr(x), st.pw[b].data_ptr(), planes.data_ptr() + b * psz,
                                   None if stats is None else stats.data_ptr() + b * _R() * 2 * Cp * 4, N, H, W, Cp, stride, stream()),
                  "hc_dw3x3_fwd")
        srcs = [planes.data_ptr() + b * psz for b in range(P)]
        lds = [Cp] * P
        sptr = [None if stats is None else stats.data_ptr() + b * _R() * 2 * Cp * 4 for b in range(P)]
        slds = [Cp] * P
        if has_id:
            srcs.append(x.data_ptr())
            lds.append(Cp)
            sptr.append(ptr(_stats_of(x)) if training else None)
            slds.append(Cp)
        coef = torch.empty((B, 2, Cp), dtype=torch.float32, device=dev)
        save = torch.empty((B, 2, Cp), dtype=torch.float32, device=dev)
        d = MsbnDesc()
        _fill_desc(d, all_infos, gammas, betas, sptr, slds, Cp, Cc, npix, training)
        d.coef, d.save = ptr(coef), ptr(save)
        check(lib.hc_msbn_finalize(C.byref(d), stream()), "hc_msbn_finalize")
        out = cv.empty_cl(N, Cp, OH, OW, dev)
        out_stats = POOL.take((_R(), 2, Cp), dev) if training else None
        io = _io(srcs, lds, npix, Cp)
        check(lib.hc_msbn_apply(C.byref(io), ptr(coef), ptr(out), ptr(out_stats), act, stream()), "hc_msbn_apply")
        st.last_out_stats = out_stats
        ctx.st, ctx.meta = st, meta
        ctx.act, ctx.training = act, training
        ctx.geom = (N, Cp, H, W, OH, OW, P, B)
        ctx.red, ctx.red_gen = (POOL.take_for_backward((_R(), B + 1, Cp), dev)
                                if any(t.requires_grad for t in params) or x.requires_grad else (None, -1))
        ctx.nb = (P, B)
        ctx.save_for_backward(x, out, save, planes, *gammas, *ws)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        st = ctx.st
        stride, has_id, infos, training, act, Cc = ctx.meta
        N, Cp, H, W, OH, OW, P, B = ctx.geom
        x, out, save, planes = ctx.saved_tensors[:4]
        gammas = ctx.saved_tensors[4:4 + B]
        ws = ctx.saved_tensors[4 + B:]
        dev = g.device
        npix = N * OH * OW
        psz = npix * Cp * 2
        srcs = [planes.data_ptr() + b * psz for b in range(P)] + ([x.data_ptr()] if has_id else [])
        io = _io(srcs, [Cp] * B, npix, Cp)
        dplanes = torch.empty_like(planes)
        did = cv.empty_cl(N, Cp, H, W, dev) if has_id else None
        dy_ptrs = [dplanes.data_ptr() + b * psz for b in range(P)] + ([did.data_ptr()] if has_id else [])
        pgrad = _msbn_backward(ctx, lib, g, out, save, io, B, gammas, Cp, Cc, npix, dy_ptrs, [Cp] * B, dev)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = cv.empty_cl(N, Cp, H, W, dev)
            dyarr = (C.c_void_p * P)(*dy_ptrs[:P])
            warr = (C.c_void_p * P)(*[st.pw[b].data_ptr() for b in range(P)])
            check(lib.hc_dwrep_dgrad(dyarr, warr, P, ptr(did), ptr(dx), N, H, W, Cp, stride, stream()), "hc_dwrep_dgrad")
        # one workspace slice per plane: successive memset + atomics rounds on ONE buffer came back corrupted from hipGraph
        # replays (the memset nodes of a captured stream are not kept ordered against the kernels between them)
        wsn = lib.hc_dw3x3_wgrad_ws_bytes(Cp) // 4
        wsb = torch.empty((P, wsn), dtype=torch.float32, device=dev)
        grads = []
        if has_id:
            grads += [pgrad[B - 1, 0, :Cc], pgrad[B - 1, 1, :Cc]]
        for b, w in enumerate(ws):
            dw3 = torch.empty((Cc, 1, 3, 3), dtype=torch.float32, device=dev)
            check(lib.hc_dw3x3_wgrad(ptr(x), dy_ptrs[b], wsb.data_ptr() + b * wsn * 4, ptr(dw3), N, H, W, Cp, Cc, stride, 0, stream()),
                  "hc_dw3x3_wgrad")
            dw = dw3 if w.shape[-1] == 3 else dw3[:, :, 1:2, 1:2].contiguous()
            grads += [dw, pgrad[b, 0, :Cc], pgrad[b, 1, :Cc]]
        return (dx, None, None, *grads)


class PointRepFn(torch.autograd.Function):
    """act(sum_b BN_b(conv1x1_b(x)) [+ BN_id(x)]);  params = [g_id, b_id]? + (w, gamma, beta) per branch."""

    @staticmethod
    def forward(ctx, x, st, meta, *params):
        lib = _lib.load()
        has_id, infos, training, act, Cin, Cout = meta
        N, Cx, H, W = x.shape
        dev = x.device
        Cin_p, Cout_p = ceil16(Cin), ceil16(Cout)
        if Cx != Cin_p or cl_ld(x) != Cin_p:
            raise _lib.HipError(f"point block expects a dense NHWC bf16 input with {Cin_p} channels, got {tuple(x.shape)}")
        off = 2 if has_id else 0
        ws = params[off::3]
        K = len(ws)
        B = K + (1 if has_id else 0)
        if B > MAXB:
            raise NotImplementedError(f"at most {MAXB} parallel branches per block")
        gammas = list(params[off + 1::3]) + ([params[0]] if has_id else [])
        betas = list(params[off + 2::3]) + ([params[1]] if has_id else [])
        all_infos = list(infos[1:] if has_id else infos) + ([infos[0]] if has_id else [])
        KC = K * Cout_p
        # ---- stacked packed weights: forward [K * Cout_p][1][Cin_p], data gradient [Cin_p][1][K * Cout_p]
        key = tuple((w.data_ptr(), w._version) for w in ws) + (cv.weights_epoch(),)
        if st.pkey != key or st.pw is None or st.pw[0].device != dev:
            if st.pw is None or st.pw[0].device != dev:
                st.pw = (torch.zeros((KC, 1, Cin_p), dtype=torch.bfloat16, device=dev),
                         torch.zeros((Cin_p, 1, KC), dtype=torch.bfloat16, device=dev))
                st.ptable = None
            ptrs = tuple(w.data_ptr() for w in ws)
            if st.ptable is None or st.ptable[1] != ptrs:
                import numpy as np
                arr = (_lib.PackItem * (2 * K))()
                for b, w in enumerate(ws):
                    if w.dtype != torch.float32 or not w.is_contiguous():
                        raise RuntimeError("point block (HIP) expects contiguous fp32 conv weights")
                    a = arr[2 * b]
                    a.w, a.dst = w.data_ptr(), st.pw[0].data_ptr() + b * Cout_p * Cin_p * 2
                    a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T, a.ld = Cout, Cin, 1, 1, 0, 0, 1, Cin_p
                    a = arr[2 * b + 1]
                    a.w, a.dst = w.data_ptr(), st.pw[1].data_ptr() + b * Cout_p * 2
                    a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T, a.ld = Cout, Cin, 1, 1, 1, 0, 1, KC
                st.ptable = (torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev), ptrs)
            check(lib.hc_pack_conv_weights_multi(st.ptable[0].data_ptr(), 2 * K, Cout * Cin, stream()), "hc_pack_conv_weights_multi")
            st.pkey = key
        npix = N * H * W
        fkey = ("f", N, H, W)
        if fkey not in st.desc:
            st.desc[fkey] = cv.fwd_desc(N, Cin_p, H, W, KC, 1, 1, 1, 0)
        Y = cv.empty_cl(N, KC, H, W, dev)
        stats = POOL.take((_R(), 2, KC), dev) if training else None
        cv.launch_conv(st.desc[fkey], x, st.pw[0], Y, stats=stats, flops=2.0 * npix * K * Cout * Cin)
        srcs = [Y.data_ptr() + b * Cout_p * 2 for b in range(K)]
        lds = [KC] * K
        sptr = [None if stats is None else stats.data_ptr() + b * Cout_p * 4 for b in range(K)]
        slds = [KC] * K
        if has_id:
            srcs.append(x.data_ptr())
            lds.append(Cin_p)
            sptr.append(ptr(_stats_of(x)) if training else None)
            slds.append(Cin_p)
        coef = torch.empty((B, 2, Cout_p), dtype=torch.float32, device=dev)
        save = torch.empty((B, 2, Cout_p), dtype=torch.float32, device=dev)
        d = MsbnDesc()
        _fill_desc(d, all_infos, gammas, betas, sptr, slds, Cout_p, Cout, npix, training)
        d.coef, d.save = ptr(coef), ptr(save)
        check(lib.hc_msbn_finalize(C.byref(d), stream()), "hc_msbn_finalize")
        out = cv.empty_cl(N, Cout_p, H, W, dev)
        out_stats = POOL.take((_R(), 2, Cout_p), dev) if training else None
        io = _io(srcs, lds, npix, Cout_p)
        check(lib.hc_msbn_apply(C.byref(io), ptr(coef), ptr(out), ptr(out_stats), act, stream()), "hc_msbn_apply")
        st.last_out_stats = out_stats
        ctx.st, ctx.meta = st, meta
        ctx.act, ctx.training = act, training
        ctx.geom = (N, H, W, Cin_p, Cout_p, K, B)
        ctx.red, ctx.red_gen = (POOL.take_for_backward((_R(), B + 1, Cout_p), dev)
                                if any(t.requires_grad for t in params) or x.requires_grad else (None, -1))
        ctx.save_for_backward(x, out, save, Y, *gammas)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        st = ctx.st
        has_id, infos, training, act, Cin, Cout = ctx.meta
        N, H, W, Cin_p, Cout_p, K, B = ctx.geom
        x, out, save, Y = ctx.saved_tensors[:4]
        gammas = ctx.saved_tensors[4:]
        dev = g.device
        npix = N * H * W
        KC = K * Cout_p
        srcs = [Y.data_ptr() + b * Cout_p * 2 for b in range(K)] + ([x.data_ptr()] if has_id else [])
        lds = [KC] * K + ([Cin_p] if has_id else [])
        io = _io(srcs, lds, npix, Cout_p)
        dY = cv.empty_cl(N, KC, H, W, dev)
        did = cv.empty_cl(N, Cin_p, H, W, dev) if has_id else None
        dy_ptrs = [dY.data_ptr() + b * Cout_p * 2 for b in range(K)] + ([did.data_ptr()] if has_id else [])
        dy_lds = [KC] * K + ([Cin_p] if has_id else [])
        pgrad = _msbn_backward(ctx, lib, g, out, save, io, B, gammas, Cout_p, Cout, npix, dy_ptrs, dy_lds, dev)
        dx = None
        if ctx.needs_input_grad[0]:
            dkey = ("d", N, H, W)
            if dkey not in st.desc:
                st.desc[dkey] = cv.dgrad_desc(N, Cin_p, H, W, KC, [(1, 1, 0, 0, 0)], 1)
            dx = cv.empty_cl(N, Cin_p, H, W, dev)
            cv.launch_conv(st.desc[dkey], dY, st.pw[1], dx, resid=did, flops=2.0 * npix * K * Cout * Cin)
        dwp = cv.conv_wgrad(x, dY, Cin_p, KC, 1, 1, 1, 0, flops=2.0 * npix * K * Cout * Cin)
        grads = []
        if has_id:
            grads += [pgrad[B - 1, 0, :Cout], pgrad[B - 1, 1, :Cout]]
        for b in range(K):
            grads += [dwp[b * Cout_p:b * Cout_p + Cout, :Cin].contiguous(), pgrad[b, 0, :Cout], pgrad[b, 1, :Cout]]
        return (dx, None, None, *grads)


class DwBiasActFn(torch.autograd.Function):
    """Re-parametrised depth block (mobileone.py:69-98: one depthwise 3x3 with bias) + ReLU, inference only."""

    @staticmethod
    def forward(ctx, x, w, bias, st, stride, act):
        lib = _lib.load()
        N, Cp, H, W = x.shape
        Cc = w.shape[0]
        dev = x.device
        if Cp != ceil16(Cc) or cl_ld(x) != Cp:
            raise _lib.HipError(f"depthwise conv expects a dense NHWC bf16 input with {ceil16(Cc)} channels, got {tuple(x.shape)}")
        key = ((w.data_ptr(), w._version), (bias.data_ptr(), bias._version))
        if st.pkey != key or st.pw is None or st.pw[0].device != dev:
            wpk = torch.empty((9, Cp), dtype=torch.float32, device=dev)
            check(lib.hc_dw3x3_pack(ptr(w.detach().float().contiguous()), ptr(wpk), Cc, Cp, 0, stream()), "hc_dw3x3_pack")
            coef = torch.zeros((1, 2, Cp), dtype=torch.float32, device=dev)
            coef[0, 0, :Cc] = 1.0
            coef[0, 1, :Cc] = bias.detach().float()
            st.pw, st.pkey = (wpk, coef), key
        wpk, coef = st.pw
        OH, OW = cv.conv_out_size(H, 3, stride, 1), cv.conv_out_size(W, 3, stride, 1)
        y = cv.empty_cl(N, Cp, OH, OW, dev)
        check(lib.hc_dw3x3_fwd(ptr(x), ptr(wpk), ptr(y), None, N, H, W, Cp, stride, stream()), "hc_dw3x3_fwd")
        out = cv.empty_cl(N, Cp, OH, OW, dev)
        io = _io([y.data_ptr()], [Cp], N * OH * OW, Cp)
        check(lib.hc_msbn_apply(C.byref(io), ptr(coef), ptr(out), None, act, stream()), "hc_msbn_apply")
        return out

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("the re-parametrised MobileOne block is an inference form; train the multi-branch form")


class PointBiasActFn(torch.autograd.Function):
    """Re-parametrised point block (mobileone.py:123-151: one dense 1x1 with bias) + ReLU, inference only."""

    @staticmethod
    def forward(ctx, x, w, bias, st, act):
        from .mbconv_op import _pack_padded
        Cout, Cin = w.shape[0], w.shape[1]
        Cin_p, Cout_p = ceil16(Cin), ceil16(Cout)
        N, Cx, H, W = x.shape
        dev = x.device
        if Cx != Cin_p or cl_ld(x) != Cin_p:
            raise _lib.HipError(f"pointwise conv expects a dense NHWC bf16 input with {Cin_p} channels, got {tuple(x.shape)}")
        wf, _ = _pack_padded(st, w, Cin_p, Cout_p, False)
        bkey = (bias.data_ptr(), bias._version)
        if getattr(st, "bkey", None) != bkey or st.bias_p.device != dev:
            st.bias_p = torch.zeros((Cout_p,), dtype=torch.float32, device=dev)
            st.bias_p[:Cout] = bias.detach().float()
            st.bkey = bkey
        key = ("fb", N, H, W)
        if key not in st.desc:
            st.desc[key] = cv.fwd_desc(N, Cin_p, H, W, Cout_p, 1, 1, 1, 0)
        out = cv.empty_cl(N, Cout_p, H, W, dev)
        cv.launch_conv(st.desc[key], x, wf, out, bias=st.bias_p, act=act, flops=2.0 * N * H * W * Cout * Cin)
        return out

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("the re-parametrised MobileOne block is an inference form; train the multi-branch form")
