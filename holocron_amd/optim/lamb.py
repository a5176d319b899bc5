"""LAMB and RaLars on multi-tensor HIP kernels (reference: holocron/optim/lamb.py:17-137, holocron/optim/ralars.py:17-140).

Same constructors, ``param_groups`` / ``state`` keys (``step``, ``exp_avg``, ``exp_avg_sq``, ``local_lr``) and update rules as
the reference, including LAMB's missing bias correction (lamb.py:121-123 divides the raw moments).  The per-tensor trust ratio
(two norms and a host-side ``if`` per tensor in the reference) is two launches for the whole model: moments + norms, then the
update with the ratio taken on the device.
"""
import math
from typing import Callable, Iterable, Optional, Tuple

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from .. import _lib
from .._lib import LambGroup, check, ptr, stream
from ..ops.conv import bump_weights_epoch, flush_deferred_wgrads
from ._multi_tensor import DeviceTables, chunk_rows, VGroups
from .adamp import _check_param, _upload

__all__ = ["LAMB", "RaLars"]


class _TrustRatioAdam(Optimizer):
    _name = "LAMB"

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas: Tuple[float, float], eps: float,
                 weight_decay: float, scale_clip: Optional[Tuple[float, float]], default_clip) -> None:
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        super().__init__(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay})
        self.scale_clip = default_clip if scale_clip is None else scale_clip

    def _mode(self, group, step):
        """(mode, rectification term) of hc_lamb_group for this group at this step."""
        return 0, 1.0

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries, owners, vg = [], [], VGroups()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check_param(p, self._name)
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p.data)
                    state["exp_avg_sq"] = torch.zeros_like(p.data)
                state["step"] += 1
                entries.append({"p": p.data, "g": p.grad, "m": state["exp_avg"], "s": state["exp_avg_sq"], "smax": None,
                                "group": vg.index(gi, state["step"]), "tensor": len(entries)})
                owners.append(state)
        gbuf = (LambGroup * max(len(vg), 1))()
        for g, (gi, st) in zip(gbuf, vg.keys):      # one launch group per (param group, step count)
            group = self.param_groups[gi]
            g.lr, g.beta1, g.beta2, g.eps = float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])
            g.weight_decay, g.clip_lo, g.clip_hi = float(group["weight_decay"]), float(self.scale_clip[0]), float(self.scale_clip[1])
            g.step = st
            g.mode, g.rect = self._mode(group, g.step)
        if not entries:
            return loss
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)       # device-resident tables: re-uploaded only when an address / hyper-parameter / step changed
        chunks = tabs.get("chunks", raw, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        norms = torch.empty((len(entries), 2), dtype=torch.float32, device=dev)
        local = torch.empty((len(entries),), dtype=torch.float32, device=dev)
        check(_lib.load().hc_lamb_step(ptr(chunks), n, ptr(gdev), ptr(norms), ptr(local), len(entries), stream()), "hc_lamb_step")
        for i, state in enumerate(owners):
            state["local_lr"] = local[i]
        self._hc_keep = (chunks, gdev, norms, local)      # alive until the stream has consumed them
        bump_weights_epoch()
        return loss


class LAMB(_TrustRatioAdam):
    """LAMB (lamb.py:17-137)."""

    _name = "LAMB"

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, scale_clip: Optional[Tuple[float, float]] = None) -> None:
        super().__init__(params, lr, betas, eps, weight_decay, scale_clip, (0.0, 10.0))


class RaLars(_TrustRatioAdam):
    """RAdam + LARS (ralars.py:17-140)."""

    _name = "RaLars"

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, force_adaptive_momentum: bool = False,
                 scale_clip: Optional[Tuple[float, float]] = None) -> None:
        super().__init__(params, lr, betas, eps, weight_decay, scale_clip, (0, 10))
        self.force_adaptive_momentum = force_adaptive_momentum

    def _mode(self, group, step):
        beta2 = group["betas"][1]
        if not isinstance(group.get("sma_inf"), float):
            group["sma_inf"] = 2 / (1 - beta2) - 1            # ralars.py:81-82
        sma_inf = group["sma_inf"]
        if step == 0:
            return 3, 1.0
        bc2 = 1 - beta2 ** step
        sma_t = sma_inf - 2 * step * (1 - bc2) / bc2          # ralars.py:106
        if sma_t > 4:
            return 1, math.sqrt((sma_t - 4) * (sma_t - 2) * sma_inf / ((sma_inf - 4) * (sma_inf - 2) * sma_t))
        return (2 if self.force_adaptive_momentum else 3), 1.0
