"""TAdam and Adan on multi-tensor HIP kernels (reference: holocron/optim/tadam.py:17-212, holocron/optim/adan.py:17-199).

Same constructors, ``param_groups`` / ``state`` keys and update rules as the reference.  TAdam's per-tensor Student-t weight
(a reduction, four scalar ops and a host round trip per tensor in the reference) is three launches for the whole model; Adan
is one.  Adan keeps the reference's ``prev_grad`` state, which the reference allocates and reads but never writes
(adan.py:107,176): the gradient difference is taken against it exactly like there.
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
from .adamp import _check_param, _upload

__all__ = ["TAdam", "Adan"]


class TAdam(Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False, dof: Optional[float] = None) -> None:
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not weight_decay >= 0.0:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        super().__init__(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "amsgrad": amsgrad, "dof": dof})

    def __setstate__(self, state) -> None:
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries, numel, dofs, tgroup, wptr, vg = [], [], [], [], [], VGroups()
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check_param(p, "TAdam")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        state["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["W_t"] = beta1 / (1 - beta1) * torch.ones(1, dtype=p.data.dtype, device=p.data.device)
                state["step"] += 1
                v = vg.index(gi, state["step"])         # one launch group per (param group, step count)
                entries.append({"p": p.data, "g": p.grad, "m": state["exp_avg"], "s": state["exp_avg_sq"],
                                "smax": state.get("max_exp_avg_sq"), "group": v, "tensor": len(entries)})
                numel.append(p.numel())
                dofs.append(float(p.numel()) if group["dof"] is None else float(group["dof"]))
                tgroup.append(v)
                wptr.append(state["W_t"].data_ptr())
        gbuf = (AdamxGroup * max(len(vg), 1))()
        for g, (gi, st) in zip(gbuf, vg.keys):
            group = self.param_groups[gi]
            g.lr, g.beta1, g.beta2, g.eps = float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])
            g.weight_decay, g.step, g.amsgrad = float(group["weight_decay"]), st, int(bool(group["amsgrad"]))
        if not entries:
            return loss
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)       # device-resident tables: re-uploaded only when an address / hyper-parameter / step changed
        chunks = tabs.get("chunks", raw, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        T = len(entries)
        scratch = torch.empty((3 * T,), dtype=torch.float32, device=dev)
        # one staging buffer for the three per-tensor tables: numel | group | dof | W_t pointers
        tab = np.zeros((T,), dtype=[("numel", "<i4"), ("group", "<i4"), ("dof", "<f4"), ("pad", "<i4"), ("w", "<u8")])
        tab["numel"], tab["group"], tab["dof"], tab["w"] = numel, tgroup, dofs, wptr
        nel = tabs.get("numel", np.ascontiguousarray(tab["numel"]), dev)
        grp = tabs.get("group", np.ascontiguousarray(tab["group"]), dev)
        dof = tabs.get("dof", np.ascontiguousarray(tab["dof"]), dev)
        wts = tabs.get("w", np.ascontiguousarray(tab["w"]), dev)
        check(_lib.load().hc_tadam_step(ptr(chunks), n, ptr(gdev), ptr(scratch), ptr(dof), ptr(nel), ptr(grp), ptr(wts), T, stream()),
              "hc_tadam_step")
        self._hc_keep = (chunks, gdev, scratch, nel, grp, dof, wts)
        bump_weights_epoch()
        return loss


class Adan(Adam):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float, float] = (0.98, 0.92, 0.99),
                 eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False) -> None:
        super().__init__(params, lr, betas, eps, weight_decay, amsgrad)  # type: ignore[arg-type]

    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        entries, extra = [], []
        vg = VGroups()
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check_param(p, "Adan")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_delta"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        state["max_exp_avg_delta"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["prev_grad"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t, v = len(entries), vg.index(gi, state["step"])
                entries.append({"p": p.data, "g": p.grad, "m": state["exp_avg"], "s": state["exp_avg_sq"],
                                "smax": state["exp_avg_delta"], "group": v, "tensor": t})
                extra.append({"p": p.data, "g": None, "m": state.get("max_exp_avg_delta"), "s": state["prev_grad"], "smax": None,
                              "group": v, "tensor": t})
        gbuf = (AdamxGroup * max(len(vg), 1))()
        for g, (gi, st) in zip(gbuf, vg.keys):
            group = self.param_groups[gi]
            b1, b2, b3 = group["betas"]
            g.lr, g.beta1, g.beta2, g.beta3, g.eps = float(group["lr"]), float(b1), float(b2), float(b3), float(group["eps"])
            g.weight_decay, g.step, g.amsgrad = float(group["weight_decay"]), st, int(bool(group["amsgrad"]))
        if not entries:
            return loss
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)
        raw2, n2 = chunk_rows(extra)
        assert n == n2
        chunks, chunks2 = tabs.get("chunks", raw, dev), tabs.get("chunks2", raw2, dev)
        gdev = tabs.get("groups", np.frombuffer(bytes(gbuf), dtype=np.uint8), dev)
        check(_lib.load().hc_adan_step(ptr(chunks), ptr(chunks2), n, ptr(gdev), stream()), "hc_adan_step")
        self._hc_keep = (chunks, chunks2, gdev)
        bump_weights_epoch()
        return loss
