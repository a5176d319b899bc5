"""Functional ops of the hot path (reference: holocron/nn/functional.py).

GPU tensors run on the HIP kernels of libholocron_hip.so; CPU tensors are rejected loudly
(the CPU restatement of these ops is test infrastructure under oracle/).
"""
from typing import Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["hard_mish", "focal_loss", "dice_loss", "poly_loss", "dropblock2d", "global_avg_pool2d", "concat_downsample2d",
           "norm_conv2d", "cross_entropy"]


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


class _CrossEntropyMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, label_smoothing, ignore_index):
        _lib.require_gpu(x, target)
        xc = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        tc = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.long().contiguous()
        N, K = xc.shape
        lib = _lib.load()
        out = torch.empty((1 + int(lib.hc_ce_mean_aux_floats(N)),), dtype=torch.float32, device=x.device)   # {loss, aux: valid rows, ...}
        check(lib.hc_ce_mean_fwd(ptr(xc), ptr(tc), out.data_ptr(), out.data_ptr() + 4, N, K, float(label_smoothing),
                                         int(ignore_index), stream()), "hc_ce_mean_fwd")
        ctx.save_for_backward(xc, tc, out)
        ctx.cfg = (float(label_smoothing), int(ignore_index), x.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        xc, tc, out = ctx.saved_tensors
        ls, ign, dt = ctx.cfg
        N, K = xc.shape
        g = dloss if (dloss.dtype == torch.float32 and dloss.is_contiguous()) else dloss.float().contiguous()
        dx = torch.empty_like(xc)
        check(_lib.load().hc_ce_mean_bwd(ptr(xc), ptr(tc), ptr(g), out.data_ptr() + 4, ptr(dx), N, K, ls, ign, stream()),
              "hc_ce_mean_bwd")
        return (dx if dt == torch.float32 else dx.to(dt)), None, None, None


def cross_entropy(x: Tensor, target: Tensor, label_smoothing: float = 0.0, ignore_index: int = -100) -> Tensor:
    """Mean softmax cross entropy of ``[N, K]`` logits with label smoothing: the criterion of the reference's training loop
    (``nn.CrossEntropyLoss(label_smoothing=...)``, references/classification/train.py:194) as two launches - forward, backward -
    instead of torch's ~25.  Class-index targets, mean reduction over the rows whose target is not ``ignore_index``."""
    if x.dim() != 2:
        raise ValueError("cross_entropy expects [N, K] logits")
    if target.shape != x.shape[:1]:
        raise ValueError("cross_entropy expects class-index targets of shape [N]")
    return _CrossEntropyMeanFn.apply(x, target, label_smoothing, ignore_index)


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
    def forward(ctx, x, keep, count, inplace):
        _lib.require_gpu(x, keep)
        N, Cc, H, W = x.shape
        lib = _lib.load()
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
        return dx, None, None, None


def _noise(shape, device) -> Tensor:
    """Uniform samples for DropBlock (functional.py:482).  A module-level hook so that parity tests can replay the
    draws recorded from the reference."""
    return torch.rand(shape, device=device)


_DEFAULT_NOISE = _noise


class DropPlan:
    """Step-level batching of DropBlock (YOLOv4 runs it 123 times per forward): after a recording pass that notes the
    (N, H, W, drop_prob, block_size) of every call, ONE ``torch.rand`` and ONE ``hc_dropblock_mask_batched`` launch at the
    start of each forward produce every keep-map and its count; the calls then just pick up their slices.  Any shape
    mismatch (or a test-installed ``_noise`` hook) drops back to the per-call kernels."""

    def __init__(self):
        self.entries = []        # recorded (N, H, W, drop_prob, block_size)
        self.ready = False
        self.cursor = 0
        self.table = None

    def begin(self, device, training):
        self.cursor = 0
        self.live = False
        if not training or _noise is not _DEFAULT_NOISE:
            self.recording = False
            return
        if not self.ready:
            self.entries, self.recording = [], True
            return
        if any(bs > 13 for (_, _, _, _, bs) in self.entries):     # the batched kernel tiles with a 6-pixel halo at most
            self.recording = False
            return
        self.recording = False
        import ctypes as C
        import numpy as np
        if self.table is None or self.table[0].device != device:
            arr = (_lib.DropItem * len(self.entries))()
            off = 0
            mx = 0
            for a, (N, H, W, p, bs) in zip(arr, self.entries):
                a.off, a.N, a.H, a.W, a.block_size, a.gamma = off, N, H, W, bs, p / bs**2
                off += N * H * W
                mx = max(mx, N * H * W)
            host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
            self.table = (host.to(device), off, mx, [a.off for a in arr])
        tab, total, mx, _ = self.table
        if total == 0:
            return
        noise = torch.rand((total,), device=device)
        self.keep = torch.empty((total,), dtype=torch.float32, device=device)
        self.counts = torch.empty((len(self.entries),), dtype=torch.float32, device=device)
        check(_lib.load().hc_dropblock_mask_batched(tab.data_ptr(), len(self.entries), mx, ptr(noise), ptr(self.keep), ptr(self.counts),
                                                    stream()), "hc_dropblock_mask_batched")
        self.live = True

    def end(self):
        if getattr(self, "recording", False):
            self.ready = True
            self.table = None
        elif self.live and self.cursor != len(self.entries):
            self.ready = False        # fewer calls than planned: re-record next time
        self.live = False

    def take(self, N, H, W, drop_prob, block_size):
        """(keep, count) of the next DropBlock call, or None when the call has to run its own kernels."""
        if getattr(self, "recording", False):
            self.entries.append((N, H, W, drop_prob, block_size))
            return None
        if not self.live:
            return None
        i = self.cursor
        if i >= len(self.entries) or self.entries[i] != (N, H, W, drop_prob, block_size):
            self.live, self.ready = False, False     # the model changed shape: abandon the plan for this forward
            return None
        self.cursor += 1
        off = self.table[3][i]
        return self.keep[off:off + N * H * W].view(N, H, W), self.counts[i:i + 1]


_ACTIVE_PLAN = [None]


class drop_plan_scope:
    """``with drop_plan_scope(model, device):`` around a model forward activates the model's DropPlan."""

    def __init__(self, model, device):
        plan = getattr(model, "_hc_drop_plan", None)
        if plan is None:
            plan = model._hc_drop_plan = DropPlan()
        self.plan, self.device, self.training = plan, device, model.training

    def __enter__(self):
        self.prev = _ACTIVE_PLAN[0]
        _ACTIVE_PLAN[0] = self.plan
        self.plan.begin(self.device, self.training)
        return self.plan

    def __exit__(self, *exc):
        _ACTIVE_PLAN[0] = self.prev
        if exc[0] is None:
            self.plan.end()
        else:
            self.plan.live, self.plan.recording = False, False
        return False


def dropblock_keep(N, H, W, drop_prob, block_size, device):
    """keep map [N][H][W] and its sum (device scalar) for DropBlock with ``gamma = drop_prob / block_size**2``."""
    if block_size % 2 == 0:
        # F.max_pool2d(kernel=bs, stride=1, padding=bs//2) grows the map by one for even sizes and the reference then
        # fails to broadcast (functional.py:485-491)
        raise RuntimeError("dropblock2d: block_size must be odd (mask and input shapes do not broadcast otherwise)")
    plan = _ACTIVE_PLAN[0]
    if plan is not None:
        got = plan.take(N, H, W, float(drop_prob), int(block_size))
        if got is not None:
            return got
    noise = _noise((N, H, W), device).float().contiguous()
    keep = torch.empty((N, H, W), dtype=torch.float32, device=device)
    count = torch.empty((1,), dtype=torch.float32, device=device)
    check(_lib.load().hc_dropblock_mask(ptr(noise), ptr(keep), ptr(count), N, H, W, block_size, drop_prob / block_size**2, stream()),
          "hc_dropblock_mask")
    return keep, count


def dropblock2d(x: Tensor, drop_prob: float, block_size: int, inplace: bool = False, training: bool = True,
                noise: Optional[Tensor] = None) -> Tensor:
    """DropBlock (holocron/nn/functional.py:465-500).  ``gamma = drop_prob / block_size**2`` (the module already
    divides once more — reference quirk Q3, kept); block centres where ``U[0,1) <= gamma``; dropped blocks are the
    max-pool dilation of the centres; the output is rescaled by ``numel / kept``.  ``noise`` (N, H, W) may be passed
    in for reproducible parity tests; otherwise it is drawn with ``torch.rand`` on x's device like the reference.
    The reference's host-side ``if one_count > 0`` is a device-side select here (no sync)."""
    if not training or drop_prob == 0:
        return x
    _lib.require_gpu(x)
    N, _, H, W = x.shape
    if noise is None:
        keep, count = dropblock_keep(N, H, W, float(drop_prob), int(block_size), x.device)
    else:
        if block_size % 2 == 0:
            raise RuntimeError("dropblock2d: block_size must be odd (mask and input shapes do not broadcast otherwise)")
        nz = noise.float().contiguous()
        keep = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
        count = torch.empty((1,), dtype=torch.float32, device=x.device)
        check(_lib.load().hc_dropblock_mask(ptr(nz), ptr(keep), ptr(count), N, H, W, int(block_size), drop_prob / block_size**2,
                                            stream()), "hc_dropblock_mask")
    return _DropBlockFn.apply(x, keep, count, inplace)


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
    from ..ops.conv import is_cl_bf16, to_cl_bf16
    _lib.require_gpu(x)
    if not is_cl_bf16(x):
        if x.shape[1] % 8 != 0:
            from .mbconv_op import _PadChannelsFn, ceil16
            return _GapFn.apply(_PadChannelsFn.apply(x, ceil16(x.shape[1])))[:, :x.shape[1]]
        # layout / dtype change through autograd-aware torch ops when a gradient has to flow back
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if x.requires_grad else to_cl_bf16(x)
        if x.stride(1) != 1:
            x = to_cl_bf16(x.detach()) if not x.requires_grad else x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return _GapFn.apply(x)


def concat_downsample2d(x: Tensor, scale_factor: int) -> Tensor:
    """``[N, C, H, W] -> [N, s*s*C, H/s, W/s]`` with output channel ``(a*s + b)*C + c`` = input pixel ``(h*s + a, w*s + b)``
    (holocron/nn/functional.py:116-136).  Channel counts that are not multiples of 8 are padded for the kernel and sliced back."""
    from ..ops.nhwc import concat_downsample2d_cl
    _lib.require_gpu(x)
    if (x.shape[2] % scale_factor != 0) or (x.shape[3] % scale_factor != 0):
        raise AssertionError("Spatial size of input tensor must be multiples of `scale_factor`")
    c = x.shape[1]
    if c % 8 == 0:
        return concat_downsample2d_cl(x, scale_factor)
    from .mbconv_op import _PadChannelsFn, ceil16
    cp = ceil16(c)
    out = concat_downsample2d_cl(_PadChannelsFn.apply(x, cp), scale_factor)
    s2 = scale_factor * scale_factor
    return torch.cat([out[:, k * cp:k * cp + c] for k in range(s2)], dim=1)


class _NormConvState:
    """Host state of the functional form, keyed on the weight tensor (packed-weight cache + descriptors)."""
    cache = {}


def norm_conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, stride=1, padding=0, dilation=1, groups: int = 1,
                eps: float = 1e-14) -> Tensor:
    """Normalised convolution, functional form (holocron/nn/functional.py:366-413): every input patch is normalised by its own
    mean / biased variance over its Cin*KH*KW entries before the filters are applied.  Same kernels as ``nn.NormConv2d``
    (nn/normconv_op.py: no unfold - patch statistics + a normalising conv epilogue); square stride / padding, dilation 1.
    Like the reference (in-place normalisation of the unfolded input) it has no gradient w.r.t. ``x``."""
    from .convbn_op import ConvState
    from .normconv_op import NormConv2dFn
    _lib.require_gpu(x, weight)

    def pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)
    stride, padding, dilation = pair(stride), pair(padding), pair(dilation)
    if dilation != (1, 1) or stride[0] != stride[1] or padding[0] != padding[1] \
            or weight.shape[2] * weight.shape[3] > _lib.HC_MAX_TAPS:
        raise NotImplementedError("norm_conv2d on the HIP path: square stride / padding, dilation 1, at most 12 taps")
    if groups != 1:   # the reference ignores `groups` (functional.py:322-363) and then fails on the matmul shapes
        raise RuntimeError("norm_conv2d: the reference ignores `groups` and fails on the matmul shapes for groups != 1")
    key = id(weight)
    ent = _NormConvState.cache.get(key)
    if ent is None or ent[0]() is not weight:
        import weakref
        if len(_NormConvState.cache) > 64:
            _NormConvState.cache.clear()
        ent = _NormConvState.cache[key] = (weakref.ref(weight), ConvState())
    return NormConv2dFn.apply(x, weight, bias, ent[1], (stride[0], padding[0], float(eps)))
