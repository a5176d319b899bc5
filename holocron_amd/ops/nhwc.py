"""Data movement of the CSP / PAN / SPP stacks on NHWC bf16 activations (kernels: csrc/nhwc_ops.hip).

Reference call sites: ``x.chunk(2, dim=1)`` / ``torch.cat(..., dim=1)`` (holocron/models/classification/
darknetv4.py:112-115), ``nn.Upsample(scale_factor=2)`` + cat (holocron/models/detection/yolov4.py:64,134-139),
``SPP`` (holocron/nn/modules/downsample.py:154-167).

A concat is a buffer allocated up front whose channel slices are handed to the producers (``cat_buffer`` /
``slice_of``): ``cat_cl`` then only checks that every part already sits in place (copying the ones that do not),
and its backward hands out slices of the incoming gradient, which the fused BN/activation backward kernels read
in place (``g_ld``).
"""
import os
from typing import List, Optional, Sequence

import torch

from .. import _lib
from .._lib import check, ptr, stream
from .conv import empty_cl, to_cl_bf16


def slice_of(buf: torch.Tensor, c0: int, Cc: int) -> torch.Tensor:
    """Channels [c0, c0 + Cc) of a dense NHWC bf16 buffer as a tensor that shares its memory but is NOT an autograd
    view of it (custom Functions return it as a fresh output)."""
    N, Ct, H, W = buf.shape
    if c0 % 8 or Cc % 8 or c0 + Cc > Ct:
        raise _lib.HipError("channel slices of NHWC buffers must be multiples of 8 channels")
    out = torch.empty((0,), dtype=buf.dtype, device=buf.device)
    out.set_(buf.untyped_storage(), buf.storage_offset() + c0, (N, Cc, H, W), (H * W * Ct, 1, W * Ct, Ct))
    return out


def cat_buffer(N: int, channels: Sequence[int], H: int, W: int, device) -> (torch.Tensor, List[torch.Tensor]):
    """Dense NHWC buffer for ``torch.cat(parts, dim=1)`` and the slices its producers should write into."""
    buf = empty_cl(N, sum(channels), H, W, device)
    parts, c0 = [], 0
    for c in channels:
        parts.append(slice_of(buf, c0, c))
        c0 += c
    return buf, parts


def _ld(t):
    from ..nn.convbn_op import cl_ld
    return cl_ld(t)


def _copy(src, dst, Cc):
    """dst[:, :Cc] = src[:, :Cc] for two NHWC bf16 tensors / slices."""
    N, _, H, W = src.shape
    check(_lib.load().hc_nhwc_copy(ptr(src), _ld(src), 0, ptr(dst), _ld(dst), 0, N * H * W, Cc, stream()), "hc_nhwc_copy")


SPLIT_STATS = {"reused": 0, "copied": 0}      # _SplitKeepFn.backward: concat gradient buffer reused / two-copy fallback (tests)
_CAT_GRADS = {}        # data_ptr of a concat gradient's first slice -> the whole gradient buffer (see _CatFn.backward)


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, *parts):
        _CAT_GRADS.clear()            # gradient buffers of an earlier backward pass that nobody claimed
        buf = holder[0]
        N, Ct, H, W = buf.shape
        c0 = 0
        base = buf.data_ptr()
        chans = []
        for p in parts:
            Cc = p.shape[1]
            if not (p.data_ptr() == base + 2 * c0 and _ld(p) == Ct):
                src = p if _ld(p) is not None else to_cl_bf16(p)
                _copy(src, slice_of(buf, c0, Cc), Cc)
            chans.append(Cc)
            c0 += Cc
        ctx.chans = chans
        return slice_of(buf, 0, Ct)

    @staticmethod
    def backward(ctx, g):
        if _ld(g) is None:
            g = to_cl_bf16(g)
        if _ld(g) != g.shape[1]:
            g = g.contiguous(memory_format=torch.channels_last)
        outs, c0 = [], 0
        for Cc in ctx.chans:
            outs.append(slice_of(g, c0, Cc))
            c0 += Cc
        # the gradient buffer itself, for a consumer that can finish its own result in it (_SplitKeepFn.backward); a handful of
        # entries at most, dropped when used
        if len(_CAT_GRADS) >= 8:
            _CAT_GRADS.clear()
        _CAT_GRADS[outs[0].data_ptr()] = g
        return (None, *outs)


def cat_cl(parts: Sequence[torch.Tensor], buf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch.cat(parts, dim=1)`` on NHWC bf16 tensors.  With ``buf`` (from ``cat_buffer``) the parts that were
    produced in place are not copied."""
    _lib.require_gpu(*parts)
    N, _, H, W = parts[0].shape
    if any(p.shape[1] % 8 for p in parts):
        raise _lib.HipError("cat_cl: channel counts must be multiples of 8")
    if buf is None:
        buf = empty_cl(N, sum(p.shape[1] for p in parts), H, W, parts[0].device)
    return _CatFn.apply([buf], *parts)


class _Chunk2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_cl_bf16(x)
        N, Ct, H, W = x.shape
        h = Ct // 2
        a, b = empty_cl(N, h, H, W, x.device), empty_cl(N, h, H, W, x.device)
        lib = _lib.load()
        check(lib.hc_nhwc_copy(ptr(x), Ct, 0, ptr(a), h, 0, N * H * W, h, stream()), "hc_nhwc_copy")
        check(lib.hc_nhwc_copy(ptr(x), Ct, h, ptr(b), h, 0, N * H * W, h, stream()), "hc_nhwc_copy")
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        N, h, H, W = ga.shape
        dx = empty_cl(N, 2 * h, H, W, ga.device)
        for k, g in enumerate((ga, gb)):
            if _ld(g) is None:
                g = to_cl_bf16(g)
            check(_lib.load().hc_nhwc_copy(ptr(g), _ld(g), 0, ptr(dx), 2 * h, k * h, N * H * W, h, stream()), "hc_nhwc_copy")
        return dx


class _SplitKeepFn(torch.autograd.Function):
    """``x.chunk(2, dim=1)`` of an ``x`` that already lives in the concat buffer its FIRST half is headed for (CSPStage: the first
    half of the base layer's output is concatenated with what the main path makes of the second): the first half is handed out in
    place, only the second - which the main path overwrites in the buffer and needs for its own backward - is copied out.  One
    activation-sized copy per stage instead of three (x -> a, x -> b, a -> buffer)."""

    @staticmethod
    def forward(ctx, x):
        N, Ct, H, W = x.shape
        h = Ct // 2
        if _ld(x) != Ct:
            raise _lib.HipError("split_keep_cl: dense NHWC bf16 input expected")
        a = slice_of(x, 0, h)
        b = empty_cl(N, h, H, W, x.device)
        check(_lib.load().hc_nhwc_copy(ptr(x), Ct, h, ptr(b), h, 0, N * H * W, h, stream()), "hc_nhwc_copy")
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        # ga is normally the first half of the gradient of the concat [x1 | main(x2)], whose second half every reader (the main
        # path's last unit) is done with by now: the gradient of x is that buffer with gb copied over its second half - one copy,
        # not two.  Anything else (a dense ga, another layout): the two-copy path of _Chunk2Fn.
        N, h, H, W = ga.shape
        g = _CAT_GRADS.pop(ga.data_ptr(), None)
        if (g is not None and tuple(g.shape) == (N, 2 * h, H, W) and _ld(g) == 2 * h and g.data_ptr() == ga.data_ptr()
                and _ld(ga) == 2 * h and os.environ.get("HC_CSP_SPLIT", "1") != "2"):
            if _ld(gb) is None:
                gb = to_cl_bf16(gb)
            check(_lib.load().hc_nhwc_copy(ptr(gb), _ld(gb), 0, ptr(g), 2 * h, h, N * H * W, h, stream()), "hc_nhwc_copy")
            SPLIT_STATS["reused"] += 1
            return g
        SPLIT_STATS["copied"] += 1
        return _Chunk2Fn.backward(ctx, ga, gb)


def split_keep_cl(x: torch.Tensor):
    """``x.chunk(2, dim=1)``: (first half IN PLACE as a slice of ``x``, second half as a dense copy)."""
    _lib.require_gpu(x)
    if x.shape[1] % 16:
        raise _lib.HipError("split_keep_cl: channel count must be a multiple of 16")
    return _SplitKeepFn.apply(x)


def chunk2_cl(x: torch.Tensor):
    """``x.chunk(2, dim=1)`` as two dense NHWC tensors (darknetv4.py:114)."""
    _lib.require_gpu(x)
    if x.shape[1] % 16:
        raise _lib.HipError("chunk2_cl: channel count must be a multiple of 16")
    return _Chunk2Fn.apply(x)


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, holder):
        x = x if _ld(x) is not None else to_cl_bf16(x)
        N, Cc, H, W = x.shape
        out = holder[0] if holder is not None else empty_cl(N, Cc, 2 * H, 2 * W, x.device)
        if tuple(out.shape) != (N, Cc, 2 * H, 2 * W) or _ld(out) is None:
            raise _lib.HipError("upsample2x: bad `out` view")
        check(_lib.load().hc_upsample2x_fwd(ptr(x), _ld(x), 0, ptr(out), _ld(out), 0, N, H, W, Cc, stream()), "hc_upsample2x_fwd")
        ctx.geom = (N, Cc, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Cc, H, W = ctx.geom
        if _ld(g) is None:
            g = to_cl_bf16(g)
        dx = empty_cl(N, Cc, H, W, g.device)
        check(_lib.load().hc_upsample2x_bwd(ptr(g), _ld(g), 0, ptr(dx), Cc, 0, N, H, W, Cc, stream()), "hc_upsample2x_bwd")
        return dx, None


def upsample2x_cl(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Nearest-neighbour x2 upsampling (``nn.Upsample(scale_factor=2, mode="nearest")``), optionally written
    straight into a concat slice."""
    _lib.require_gpu(x)
    if x.shape[1] % 8:
        raise _lib.HipError("upsample2x_cl: channel count must be a multiple of 8")
    return _Upsample2xFn.apply(x, None if out is None else [out])


class _SppFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_cl_bf16(x)
        N, Cc, H, W = x.shape
        out = empty_cl(N, 4 * Cc, H, W, x.device)
        idx = torch.empty((3, N, H, W, Cc), dtype=torch.uint8, device=x.device)
        check(_lib.load().hc_spp_fwd(ptr(x), ptr(out), ptr(idx), N, H, W, Cc, stream()), "hc_spp_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, Cc, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        N, Cc, H, W = ctx.geom
        g = to_cl_bf16(g)
        if _ld(g) != 4 * Cc:
            g = g.contiguous(memory_format=torch.channels_last)
        dx = empty_cl(N, Cc, H, W, g.device)
        check(_lib.load().hc_spp_bwd(ptr(g), ptr(idx), ptr(dx), N, H, W, Cc, stream()), "hc_spp_bwd")
        return dx


def spp_cl(x: torch.Tensor) -> torch.Tensor:
    """``SPP([5, 9, 13])``: cat([x, maxpool5(x), maxpool9(x), maxpool13(x)], dim=1), stride 1, same padding."""
    _lib.require_gpu(x)
    if x.shape[1] % 8:
        raise _lib.HipError("spp_cl: channel count must be a multiple of 8")
    return _SppFn.apply(x)


class _MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x if (_ld(x) == x.shape[1]) else to_cl_bf16(x).contiguous(memory_format=torch.channels_last)
        N, Cc, H, W = x.shape
        out = empty_cl(N, Cc, H // 2, W // 2, x.device)
        idx = torch.empty((N, H // 2, W // 2, Cc // 8), dtype=torch.int16, device=x.device)
        check(_lib.load().hc_maxpool2_fwd(ptr(x), ptr(out), ptr(idx), N, H, W, Cc, stream()), "hc_maxpool2_fwd")
        ctx.geom = (N, Cc, H, W)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Cc, H, W = ctx.geom
        (idx,) = ctx.saved_tensors
        g = g if (_ld(g) == Cc) else to_cl_bf16(g).contiguous(memory_format=torch.channels_last)
        dx = empty_cl(N, Cc, H, W, g.device)
        check(_lib.load().hc_maxpool2_bwd(ptr(g), ptr(idx), ptr(dx), N, H, W, Cc, stream()), "hc_maxpool2_bwd")
        return dx


def maxpool2_cl(x: torch.Tensor) -> torch.Tensor:
    """``nn.MaxPool2d(2)`` on an NHWC bf16 activation (darknet.py:83, darknetv2.py:94)."""
    _lib.require_gpu(x)
    if x.shape[1] % 8:
        raise _lib.HipError("maxpool2_cl: channel count must be a multiple of 8")
    return _MaxPool2Fn.apply(x)


class _SpaceToDepthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, holder):
        x = x if (_ld(x) == x.shape[1]) else to_cl_bf16(x).contiguous(memory_format=torch.channels_last)
        N, Cc, H, W = x.shape
        OH, OW = H // s, W // s
        out = holder[0] if holder is not None else empty_cl(N, Cc * s * s, OH, OW, x.device)
        if tuple(out.shape) != (N, Cc * s * s, OH, OW) or _ld(out) is None:
            raise _lib.HipError("concat_downsample2d: bad `out` view")
        check(_lib.load().hc_space_to_depth(ptr(x), ptr(out), _ld(out), 0, N, OH, OW, Cc, s, 0, stream()), "hc_space_to_depth")
        ctx.geom = (N, Cc, H, W, s)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Cc, H, W, s = ctx.geom
        if _ld(g) is None:
            g = to_cl_bf16(g)
        dx = empty_cl(N, Cc, H, W, g.device)
        check(_lib.load().hc_space_to_depth(ptr(g), ptr(dx), _ld(g), 0, N, H // s, W // s, Cc, s, 1, stream()), "hc_space_to_depth")
        return dx, None, None


def concat_downsample2d_cl(x: torch.Tensor, scale_factor: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``concat_downsample2d`` (nn/functional.py:116-136), optionally written straight into a concat slice."""
    _lib.require_gpu(x)
    if (x.shape[2] % scale_factor != 0) or (x.shape[3] % scale_factor != 0):
        raise AssertionError("Spatial size of input tensor must be multiples of `scale_factor`")
    if x.shape[1] % 8:
        raise _lib.HipError("concat_downsample2d: channel count must be a multiple of 8")
    return _SpaceToDepthFn.apply(x, int(scale_factor), None if out is None else [out])
