"""AdamP and AdEMAMix on multi-tensor HIP kernels (reference: holocron/optim/adamp.py:17-200,
holocron/optim/ademamix.py:16-200) — the optimizers references/classification/train.py:33,206-213 imports next to
AdaBelief (``--opt adamp`` is that script's default).

Same constructors, ``param_groups`` / ``state`` keys (``step``, ``exp_avg``, ``exp_avg_sq`` [, ``max_exp_avg_sq`` |
``exp_avg_slow``]) and update rules as the reference.  AdamP's projection test (``cosine_similarity(param, grad) <
delta / sqrt(numel)``, a host-side ``if`` per tensor in the reference) is decided on the device from per-tensor sums.
"""
from typing import Callable, Iterable, Optional, Tuple

import numpy as np
import torch
from torch.optim import Adam
from torch.optim.optimizer import Optimizer

from .. import _lib
from .._lib import AdamxGroup, check, ptr, stream
from ..ops.conv import bump_weights_epoch, flush_deferred_wgrads
from ._multi_tensor import DeviceTables, chunk_rows, VGroups

__all__ = ["AdamP", "AdEMAMix"]


def _check_param(p, name):
    if p.grad.is_sparse:
        raise RuntimeError(f"{name} does not support sparse gradients")
    _lib.require_gpu(p)
    if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
        raise RuntimeError(f"{name} (HIP) expects contiguous fp32 parameters and gradients")


def _upload(gbuf, dev):
    return torch.from_numpy(np.frombuffer(bytes(gbuf), dtype=np.uint8).copy()).to(dev)


class AdamP(Adam):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False, delta: float = 0.1) -> None:
        super().__init__(params, lr, betas, eps, weight_decay, amsgrad)
        self.delta = delta

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries, numel, vg = [], [], VGroups()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check_param(p, "AdamP")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        state["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                entries.append({"p": p.data, "g": p.grad, "m": state["exp_avg"], "s": state["exp_avg_sq"],
                                "smax": state.get("max_exp_avg_sq"), "group": vg.index(gi, state["step"]), "tensor": len(entries)})
                numel.append(p.numel())
        gbuf = (AdamxGroup * max(len(vg), 1))()
        for g, (gi, st) in zip(gbuf, vg.keys):      # one launch group per (param group, step count)
            group = self.param_groups[gi]
            g.lr, g.beta1, g.beta2, g.eps = float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])
            g.weight_decay, g.delta, g.step, g.amsgrad = float(group["weight_decay"]), float(self.delta), st, int(bool(group["amsgrad"]))
        if not entries:
            return loss
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)       # device-resident tables: re-uploaded only when an address / hyper-parameter / step changed
        chunks = tabs.get("chunks", raw, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        sums = torch.empty((len(entries), 4), dtype=torch.float32, device=dev)
        nel = tabs.get("numel", np.asarray(numel, dtype=np.int32), dev)
        check(_lib.load().hc_adamp_step(ptr(chunks), n, ptr(gdev), ptr(sums), ptr(nel), len(entries), stream()), "hc_adamp_step")
        self._hc_keep = (chunks, gdev, sums, nel)      # alive until the stream has consumed them
        bump_weights_epoch()
        return loss


class AdEMAMix(Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float, float] = (0.9, 0.999, 0.9999),
                 alpha: float = 5.0, eps: float = 1e-8, weight_decay: float = 0.0) -> None:
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        for idx, beta in enumerate(betas):
            if not 0.0 <= beta < 1.0:
                raise ValueError(f"Invalid beta parameter at index {idx}: {beta}")
        super().__init__(params, {"lr": lr, "betas": betas, "alpha": alpha, "eps": eps, "weight_decay": weight_decay})

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries, vg = [], VGroups()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check_param(p, "AdEMAMix")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_slow"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                entries.append({"p": p.data, "g": p.grad, "m": state["exp_avg"], "s": state["exp_avg_sq"],
                                "smax": state["exp_avg_slow"], "group": vg.index(gi, state["step"]), "tensor": len(entries)})
        gbuf = (AdamxGroup * max(len(vg), 1))()
        for g, (gi, st) in zip(gbuf, vg.keys):
            group = self.param_groups[gi]
            b1, b2, b3 = group["betas"]
            g.lr, g.beta1, g.beta2, g.beta3, g.alpha = float(group["lr"]), float(b1), float(b2), float(b3), float(group["alpha"])
            g.eps, g.weight_decay, g.step = float(group["eps"]), float(group["weight_decay"]), st
        if not entries:
            return loss
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)
        chunks = tabs.get("chunks", raw, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        check(_lib.load().hc_ademamix_step(ptr(chunks), n, ptr(gdev), stream()), "hc_ademamix_step")
        self._hc_keep = (chunks, gdev)
        bump_weights_epoch()
        return loss
