"""LARS on multi-tensor HIP kernels (reference: holocron/optim/lars.py:17-135).

Reproduces the reference step, quirks included: ``scale_clip`` is stored but never
used, the local LR is unclipped, and weight decay is added *in place* into ``p.grad``.  The two
norms per tensor are reduced on the device (no host sync per parameter).
"""
import ctypes as C
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from .. import _lib
from .._lib import LarsGroup, check, ptr, stream
from ..ops.conv import bump_weights_epoch, flush_deferred_wgrads
from ._multi_tensor import DeviceTables, chunk_rows

__all__ = ["LARS"]


class LARS(Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, momentum: float = 0.0,
                 dampening: float = 0.0, weight_decay: float = 0.0, nesterov: bool = False,
                 scale_clip: Optional[Tuple[float, float]] = None) -> None:
        if not isinstance(lr, float) or lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if momentum < 0.0:
            raise ValueError(f"Invalid momentum value: {momentum}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        defaults = {"lr": lr, "momentum": momentum, "dampening": dampening, "weight_decay": weight_decay,
                    "nesterov": nesterov}
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, defaults)
        self.scale_clip = scale_clip
        if self.scale_clip is None:
            self.scale_clip = (0.0, 10.0)

    def __setstate__(self, state: Dict[str, torch.Tensor]) -> None:
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("nesterov", False)

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries = []
        ngroups = len(self.param_groups)
        gbuf = (LarsGroup * max(ngroups, 1))()
        for gi, group in enumerate(self.param_groups):
            g = gbuf[gi]
            g.lr, g.momentum, g.dampening = float(group["lr"]), float(group["momentum"]), float(group["dampening"])
            g.weight_decay, g.nesterov = float(group["weight_decay"]), int(bool(group["nesterov"]))
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("LARS (HIP) expects contiguous fp32 parameters and gradients")
                flags, buf = 0, None
                if group["momentum"] != 0:
                    st = self.state[p]
                    if "momentum_buffer" not in st:
                        st["momentum_buffer"] = torch.empty_like(p)
                        flags = 1
                    buf = st["momentum_buffer"]
                entries.append({"p": p.data, "g": p.grad, "m": buf, "group": gi, "tensor": len(entries), "flags": flags})
        if not entries:
            return loss
        dev = entries[0]["p"].device
        # device-resident tables, re-uploaded only when an address or a hyper-parameter changed: no per-step H2D copy, and the step
        # can be captured in a hipGraph (LARS has no step-dependent scalar)
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)
        chunks = tabs.get("chunks", raw, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        norms = getattr(self, "_hc_norms", None)
        if norms is None or norms.shape[0] != len(entries) or norms.device != dev:
            norms = self._hc_norms = torch.empty((len(entries), 2), dtype=torch.float32, device=dev)
        check(_lib.load().hc_lars_step(ptr(chunks), n, ptr(gdev), ptr(norms), len(entries), stream()), "hc_lars_step")
        bump_weights_epoch()
        return loss
