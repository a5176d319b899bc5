"""AdaBelief on one multi-tensor HIP kernel (reference: holocron/optim/adabelief.py:16-167).

Follows the reference code, not the paper: epsilon is only added to the denominator (SURVEY.md
Q5) and weight decay is L2-into-gradient.  ``param_groups`` and ``state`` are re-discovered at
every ``step()`` so that ``Trainer._reset_opt`` (holocron/trainer/core.py:238-252) and per-batch LR
schedulers keep working; ``state[p]`` exposes ``step``, ``exp_avg``, ``exp_avg_sq``
[, ``max_exp_avg_sq``] like the reference.
"""
import ctypes as C
from typing import Callable, Optional

import numpy as np
import torch
from torch.optim import Adam

from .. import _lib
from .._lib import AdaBeliefGroup, check, ptr, stream
from ..ops.conv import bump_weights_epoch
from ._multi_tensor import build_chunks

__all__ = ["AdaBelief"]


class AdaBelief(Adam):
    """Same constructor as ``torch.optim.Adam`` (lr, betas, eps, weight_decay, amsgrad)."""

    def _table(self, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), gi) for p, gi in plist)
        cache = getattr(self, "_hc_table", None)
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        entries = []
        for ti, (p, gi) in enumerate(plist):
            st = self.state[p]
            entries.append({"p": p.data, "g": p.grad, "m": st["exp_avg"], "s": st["exp_avg_sq"],
                            "smax": st.get("max_exp_avg_sq"), "group": gi, "tensor": ti})
        host, n = build_chunks(entries)
        dev = host.to(plist[0][0].device)
        self._hc_table = (key, dev, n)
        return dev, n

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plist = []
        ngroups = len(self.param_groups)
        gbuf = (AdaBeliefGroup * max(ngroups, 1))()
        for gi, group in enumerate(self.param_groups):
            gstep = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError(f"{self.__class__.__name__} does not support sparse gradients")
                _lib.require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("AdaBelief (HIP) expects contiguous fp32 parameters and gradients")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        state["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                if gstep is None:
                    gstep = state["step"]
                elif gstep != state["step"]:
                    raise RuntimeError("AdaBelief (HIP): parameters of one group must share the step count")
                plist.append((p, gi))
            beta1, beta2 = group["betas"]
            g = gbuf[gi]
            g.lr, g.beta1, g.beta2, g.eps = float(group["lr"]), float(beta1), float(beta2), float(group["eps"])
            g.weight_decay, g.step, g.amsgrad = float(group["weight_decay"]), int(gstep or 1), int(bool(group["amsgrad"]))
        if not plist:
            return loss
        dev = plist[0][0].device
        chunks, n = self._table(plist)
        host = torch.from_numpy(np.frombuffer(bytes(gbuf), dtype=np.uint8).copy())
        gdev = getattr(self, "_hc_groups", None)
        if gdev is None or gdev.numel() != host.numel() or gdev.device != dev:
            gdev = self._hc_groups = torch.empty(host.numel(), dtype=torch.uint8, device=dev)
        gdev.copy_(host, non_blocking=False)
        check(_lib.load().hc_adabelief_step(ptr(chunks), n, ptr(gdev), stream()), "hc_adabelief_step")
        bump_weights_epoch()
        return loss
