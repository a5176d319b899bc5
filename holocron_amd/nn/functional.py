"""Functional ops of the hot path (reference: holocron/nn/functional.py).

GPU tensors run on the HIP kernels of libholocron_hip.so; CPU tensors are rejected loudly
(the CPU restatement of these ops is test infrastructure under oracle/).
"""
from typing import Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["hard_mish", "focal_loss", "dice_loss", "poly_loss", "dropblock2d", "global_avg_pool2d"]


class _HardMishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inplace):
        _lib.require_gpu(x)
        xc = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        if inplace and xc is x:
            ctx.save_for_backward(x.clone())
            check(_lib.load().hc_hard_mish_fwd(ptr(x), ptr(x), x.numel(), stream()), "hc_hard_mish_fwd")
            ctx.mark_dirty(x)
            return x
        y = torch.empty_like(xc)
        check(_lib.load().hc_hard_mish_fwd(ptr(xc), ptr(y), xc.numel(), stream()), "hc_hard_mish_fwd")
        ctx.save_for_backward(xc)
        y = y.to(x.dtype)
        if inplace:
            x.copy_(y)
            ctx.mark_dirty(x)
            return x
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dyc = dy.float().contiguous()
        dx = torch.empty_like(x)
        check(_lib.load().hc_hard_mish_bwd(ptr(x), ptr(dyc), ptr(dx), x.numel(), stream()), "hc_hard_mish_bwd")
        return dx.to(dy.dtype), None


def hard_mish(x: Tensor, inplace: bool = False) -> Tensor:
    """HardMish: ``0.5 * x * clamp(x + 2, 0, 2)`` (holocron/nn/functional.py:30-41)."""
    return _HardMishFn.apply(x, inplace)


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, weight, ignore_index, gamma):
        _lib.require_gpu(x, target)
        N, K = x.shape[0], x.shape[1]
        S = 1
        for s in x.shape[2:]:
            S *= s
        xc = x.float().contiguous()
        tc = target.contiguous()
        wc = None if weight is None else weight.to(device=x.device, dtype=torch.float32).contiguous()
        loss_el = torch.empty((N * S,), dtype=torch.float32, device=x.device)
        valid = torch.empty((N * S,), dtype=torch.uint8, device=x.device)
        check(_lib.load().hc_focal_loss_fwd(ptr(xc), ptr(tc), ptr(wc), ptr(loss_el), ptr(valid), N, K, S, ignore_index,
                                            gamma, stream()), "hc_focal_loss_fwd")
        ctx.save_for_backward(xc, tc, wc)
        ctx.meta = (N, K, S, gamma, x.dtype)
        ctx.mark_non_differentiable(valid)
        return loss_el, valid

    @staticmethod
    def backward(ctx, dloss, _dvalid):
        xc, tc, wc = ctx.saved_tensors
        N, K, S, gamma, dt = ctx.meta
        dx = torch.empty_like(xc)
        dl = dloss.float().contiguous()
        check(_lib.load().hc_focal_loss_bwd(ptr(xc), ptr(tc), ptr(wc), ptr(dl), ptr(dx), N, K, S, gamma, stream()),
              "hc_focal_loss_bwd")
        return dx.to(dt), None, None, None, None


def focal_loss(x: Tensor, target: Tensor, weight: Optional[Tensor] = None, ignore_index: int = -100,
               reduction: str = "mean", gamma: float = 2.0) -> Tensor:
    """Focal loss (holocron/nn/functional.py:59-113): ``-(1-p_t)^gamma * w_t * log p_t``; ``ignore_index``
    is honoured only when ``0 <= ignore_index < K`` and masks by target value; ``mean`` divides by the
    number of kept elements; ``none`` returns every element (ignored ones included) shaped like target."""
    loss_el, valid = _FocalFn.apply(x, target, weight, ignore_index, float(gamma))
    if reduction == "sum":
        return (loss_el * valid.to(loss_el.dtype)).sum().to(x.dtype)
    if reduction == "mean":
        v = valid.to(loss_el.dtype)
        return ((loss_el * v).sum() / v.sum()).to(x.dtype)
    return loss_el.view(*target.shape).to(x.dtype)


def _nks(x):
    N, K = x.shape[0], x.shape[1]
    S = 1
    for s in x.shape[2:]:
        S *= s
    return N, K, S


class _PolyHardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, weight, ignore_index, eps):
        _lib.require_gpu(x, target)
        N, K, S = _nks(x)
        xc, tc = x.float().contiguous(), target.contiguous()
        wc = None if weight is None else weight.to(device=x.device, dtype=torch.float32).contiguous()
        loss_el = torch.empty((N * S,), dtype=torch.float32, device=x.device)
        valid = torch.empty((N * S,), dtype=torch.uint8, device=x.device)
        check(_lib.load().hc_poly_loss_hard_fwd(ptr(xc), ptr(tc), ptr(wc), ptr(loss_el), ptr(valid), N, K, S, ignore_index,
                                                eps, stream()), "hc_poly_loss_hard_fwd")
        ctx.save_for_backward(xc, tc, wc)
        ctx.meta = (N, K, S, eps, x.dtype)
        ctx.mark_non_differentiable(valid)
        return loss_el, valid

    @staticmethod
    def backward(ctx, dloss, _dvalid):
        xc, tc, wc = ctx.saved_tensors
        N, K, S, eps, dt = ctx.meta
        dx = torch.empty_like(xc)
        dl = dloss.float().contiguous()
        check(_lib.load().hc_poly_loss_hard_bwd(ptr(xc), ptr(tc), ptr(wc), ptr(dl), ptr(dx), N, K, S, eps, stream()),
              "hc_poly_loss_hard_bwd")
        return dx.to(dt), None, None, None, None


class _PolySoftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, weight, ignore_index, eps):
        _lib.require_gpu(x, target)
        N, K, S = _nks(x)
        xc, tc = x.float().contiguous(), target.float().contiguous()
        wc = None if weight is None else weight.to(device=x.device, dtype=torch.float32).contiguous()
        loss_pos = torch.empty((N * S,), dtype=torch.float32, device=x.device)
        check(_lib.load().hc_poly_loss_soft_fwd(ptr(xc), ptr(tc), ptr(wc), ptr(loss_pos), N, K, S, ignore_index, eps,
                                                stream()), "hc_poly_loss_soft_fwd")
        ctx.save_for_backward(xc, tc, wc)
        ctx.meta = (N, K, S, ignore_index, eps, x.dtype)
        return loss_pos

    @staticmethod
    def backward(ctx, dloss):
        xc, tc, wc = ctx.saved_tensors
        N, K, S, ign, eps, dt = ctx.meta
        dx = torch.empty_like(xc)
        dl = dloss.float().contiguous()
        check(_lib.load().hc_poly_loss_soft_bwd(ptr(xc), ptr(tc), ptr(wc), ptr(dl), ptr(dx), N, K, S, ign, eps, stream()),
              "hc_poly_loss_soft_bwd")
        return dx.to(dt), None, None, None, None


def poly_loss(x: Tensor, target: Tensor, eps: float = 2.0, weight: Optional[Tensor] = None, ignore_index: int = -100,
              reduction: str = "mean") -> Tensor:
    """Poly-1 loss (holocron/nn/functional.py:540-613): ``-log p_t + eps (1 - p_t)`` for int64 hard labels
    (``target.ndim == x.ndim - 1``) or soft labels of x's shape.  Errors as in the reference: ``TypeError`` for a
    non-int64 hard target, ``ValueError`` for a soft target whose shape disagrees (functional.py:568-575)."""
    if target.ndim == x.ndim - 1:
        if target.dtype != torch.long:
            raise TypeError("target dtype is expected to be torch.int64")
        loss_el, valid = _PolyHardFn.apply(x, target, weight, ignore_index, float(eps))
        v = valid.to(loss_el.dtype)
        if reduction == "sum":
            return (loss_el * v).sum().to(x.dtype)
        if reduction == "mean":
            return ((loss_el * v).sum() / v.sum()).to(x.dtype)
        return loss_el.to(x.dtype)  # the reference leaves the unreduced hard-label loss flat (N*S,)
    if target.ndim != x.ndim or target.shape[0] != x.shape[0] or target.shape[1] != x.shape[1]:
        raise ValueError("invalid target shape")
    loss_pos = _PolySoftFn.apply(x, target, weight, ignore_index, float(eps))
    if reduction == "sum":
        return loss_pos.sum().to(x.dtype)
    if reduction == "mean":
        # `loss[:, valid].sum(1).mean()`: mean over the (N, ...) positions
        return loss_pos.mean().to(x.dtype)
    return loss_pos.view(x.shape[0], *x.shape[2:]).to(x.dtype)


class _DiceSumsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        _lib.require_gpu(x, target)
        N, K, S = _nks(x)
        xc, tc = x.float().contiguous(), target.float().contiguous()
        sums = torch.empty((3, K), dtype=torch.float32, device=x.device)
        check(_lib.load().hc_dice_sums(ptr(xc), ptr(tc), ptr(sums), N, K, S, stream()), "hc_dice_sums")
        ctx.save_for_backward(tc)
        ctx.meta = (N, K, S, x.shape, x.dtype)
        return sums

    @staticmethod
    def backward(ctx, dsums):
        (tc,) = ctx.saved_tensors
        N, K, S, shape, dt = ctx.meta
        dx = torch.empty(shape, dtype=torch.float32, device=tc.device)
        ds = dsums.float().contiguous()
        check(_lib.load().hc_dice_bwd(ptr(tc), ptr(ds), ptr(dx), N, K, S, stream()), "hc_dice_bwd")
        return dx.to(dt), None


def dice_loss(x: Tensor, target: Tensor, weight: Optional[Tensor] = None, gamma: float = 1.0, eps: float = 1e-8) -> Tensor:
    """Dice loss (holocron/nn/functional.py:503-537): ``1 - (1 + 1/gamma) * mean_k (gamma*sum(x t) + eps) /
    (sum(x + gamma t) + eps)``; the (N, S) reductions run on the GPU kernel, the K-sized expression here."""
    sums = _DiceSumsFn.apply(x, target)
    inter = gamma * sums[0]
    cardinality = sums[1] + gamma * sums[2]
    dice_coeff = (inter + eps) / (cardinality + eps)
    if weight is None:
        loss = 1 - (1 + 1 / gamma) * dice_coeff.mean()
    else:
        w = weight.to(device=x.device, dtype=dice_coeff.dtype)
        loss = 1 - (1 + 1 / gamma) * (w * dice_coeff).sum() / w.sum()
    return loss.to(x.dtype)


class _DropBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, gamma, block_size, inplace):
        _lib.require_gpu(x, noise)
        if block_size % 2 == 0:
            # F.max_pool2d(kernel=bs, stride=1, padding=bs//2) grows the map by one for even sizes and the
            # reference then fails to broadcast (functional.py:485-491)
            raise RuntimeError("dropblock2d: block_size must be odd (mask and input shapes do not broadcast otherwise)")
        N, Cc, H, W = x.shape
        lib = _lib.load()
        nz = noise.float().contiguous()
        keep = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
        count = torch.empty((1,), dtype=torch.float32, device=x.device)
        check(lib.hc_dropblock_mask(ptr(nz), ptr(keep), ptr(count), N, H, W, block_size, gamma, stream()), "hc_dropblock_mask")
        nhwc = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if x.dtype not in (torch.float32, torch.bfloat16) or not (nhwc or x.is_contiguous()):
            raise _lib.HipError("dropblock2d expects a dense fp32 or bf16 tensor (NCHW or channels_last)")
        y = x if inplace else torch.empty_like(x)
        check(lib.hc_dropblock_apply(ptr(x), ptr(keep), ptr(count), ptr(y), N, Cc, H * W,
                                     0 if x.dtype == torch.float32 else 1, int(nhwc), stream()), "hc_dropblock_apply")
        ctx.save_for_backward(keep, count)
        ctx.meta = (N, Cc, H * W)
        if inplace:
            ctx.mark_dirty(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        keep, count = ctx.saved_tensors
        N, Cc, HW = ctx.meta
        nhwc = dy.is_contiguous(memory_format=torch.channels_last) and not dy.is_contiguous()
        if not nhwc:
            dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty_like(dy)
        check(_lib.load().hc_dropblock_apply(ptr(dy), ptr(keep), ptr(count), ptr(dx), N, Cc, HW,
                                             0 if dy.dtype == torch.float32 else 1, int(nhwc), stream()), "hc_dropblock_apply")
        return dx, None, None, None, None


def _noise(shape, device) -> Tensor:
    """Uniform samples for DropBlock (functional.py:482).  A module-level hook so that parity tests can replay the
    draws recorded from the reference."""
    return torch.rand(shape, device=device)


def dropblock2d(x: Tensor, drop_prob: float, block_size: int, inplace: bool = False, training: bool = True,
                noise: Optional[Tensor] = None) -> Tensor:
    """DropBlock (holocron/nn/functional.py:465-500).  ``gamma = drop_prob / block_size**2`` (the module already
    divides once more — reference quirk Q3, kept); block centres where ``U[0,1) <= gamma``; dropped blocks are the
    max-pool dilation of the centres; the output is rescaled by ``numel / kept``.  ``noise`` (N, H, W) may be passed
    in for reproducible parity tests; otherwise it is drawn with ``torch.rand`` on x's device like the reference.
    The reference's host-side ``if one_count > 0`` is a device-side select here (no sync)."""
    if not training or drop_prob == 0:
        return x
    gamma = drop_prob / block_size**2
    if noise is None:
        noise = _noise((x.shape[0], *x.shape[2:]), x.device)
    return _DropBlockFn.apply(x, noise, float(gamma), int(block_size), inplace)


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        check(_lib.load().hc_gap_fwd(ptr(x), ptr(y), N, H * W, Cc, stream()), "hc_gap_fwd")
        ctx.shape = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W = ctx.shape
        dx = torch.empty((N, Cc, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        dyc = dy.float().contiguous()
        check(_lib.load().hc_gap_bwd(ptr(dyc), ptr(dx), N, H * W, Cc, stream()), "hc_gap_bwd")
        return dx


def global_avg_pool2d(x: Tensor) -> Tensor:
    """Mean over H*W of an NHWC-bf16 activation -> fp32 [N, C]."""
    from ..ops.conv import to_cl_bf16
    if x.shape[1] % 8 != 0:
        return x.float().flatten(2).mean(2)
    return _GapFn.apply(to_cl_bf16(x))
