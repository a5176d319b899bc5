"""Functional ops of the hot path (reference: holocron/nn/functional.py).

GPU tensors run on the HIP kernels of libholocron_hip.so; CPU tensors are rejected loudly
(the CPU restatement of these ops is test infrastructure under oracle/).
"""
from typing import Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["hard_mish", "focal_loss", "global_avg_pool2d"]


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
