"""Lookahead and Scout optimizer wrappers (reference: holocron/optim/wrapper.py:18-283).

Same constructors, ``state_dict`` layout (``base_state_dict``), ``fast_steps`` bookkeeping and synchronisation rule as the
reference; the synchronisation ``slow += rate * (fast - slow); fast = slow`` over all parameters is one multi-tensor launch
(``hc_lookahead_sync``) instead of three torch kernels per tensor.
"""
from collections import defaultdict
from typing import Any, Callable, Dict, Optional

import torch
from torch.optim.optimizer import Optimizer

from .. import _lib
from .._lib import check, ptr, stream
from ..ops.conv import bump_weights_epoch
from ._multi_tensor import DeviceTables, chunk_rows

__all__ = ["Lookahead", "Scout"]


def _check_sync_args(sync_rate: float, sync_period: int) -> None:
    if not 0 <= sync_rate <= 1:
        raise ValueError(f"expected positive float lower than 1 as sync_rate, received: {sync_rate}")
    if not (isinstance(sync_period, int) and sync_period >= 1):
        raise ValueError(f"expected positive integer as sync_period, received: {sync_period}")


class Lookahead(Optimizer):
    """k steps forward, 1 step back (wrapper.py:18-134).  ``param_groups`` hold the slow weights (detached copies of the base
    optimizer's parameters), ``fast_steps`` counts base steps, every ``sync_period``-th step synchronises."""

    def __init__(self, base_optimizer: torch.optim.Optimizer, sync_rate: float = 0.5, sync_period: int = 6) -> None:
        _check_sync_args(sync_rate, sync_period)
        # like the reference, torch's Optimizer.__init__ is not run: the wrapper owns no hyper-parameters of its own
        self.base_optimizer = base_optimizer
        self.defaults = dict(sync_rate=sync_rate, sync_period=sync_period)
        self.state = defaultdict(dict)
        self.fast_steps = 0
        self.param_groups = []
        for fast_group in base_optimizer.param_groups:
            self._add_param_group(fast_group)

    # ---- bookkeeping -------------------------------------------------------------------------------------------------
    def _add_param_group(self, param_group: Dict[str, Any]) -> None:
        """Slow twin of one base group: detached clones of its parameters, same learning rate entry."""
        slow = [fast.detach().clone() for fast in param_group["params"]]
        for t in slow:
            t.requires_grad = False
        self.param_groups.append({"params": slow, "lr": param_group["lr"]})

    def add_param_group(self, param_group: Dict[str, Any]) -> None:
        """New group for the base optimizer (fast weights) plus its slow twin."""
        self.base_optimizer.add_param_group(param_group)
        self._add_param_group(self.base_optimizer.param_groups[-1])

    def __getstate__(self) -> Dict[str, Any]:
        keys = ("defaults", "state", "fast_steps", "param_groups")
        out = {k: getattr(self, k) for k in keys}
        out["base_state"] = self.base_optimizer.__getstate__()
        return out

    def state_dict(self) -> Dict[str, Any]:
        own = super().state_dict()
        own["base_state_dict"] = self.base_optimizer.state_dict()
        return own

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        self.base_optimizer.load_state_dict(state_dict["base_state_dict"])
        super().load_state_dict(state_dict)
        self.__setstate__({"base_state_dict": self.base_optimizer.state_dict()})

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.base_optimizer.zero_grad(set_to_none)

    def __repr__(self) -> str:
        inner = repr(self.base_optimizer).replace("\n", "\n\t")
        lines = [f"{type(self).__name__} (", f"base_optimizer={inner},"] + [f"{k}={v}," for k, v in self.defaults.items()] + [")"]
        return "\n".join(lines)

    # ---- optimisation ------------------------------------------------------------------------------------------------
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = self.base_optimizer.step(closure)
        self.fast_steps += 1
        if self.fast_steps % self.defaults["sync_period"] == 0:
            self.sync_params(self.defaults["sync_rate"])
        return loss

    def sync_params(self, sync_rate: float = 0.0) -> None:
        """slow <- slow + sync_rate * (fast - slow) when sync_rate > 0, then fast <- slow (wrapper.py:121-134): one multi-tensor
        launch over every parameter."""
        entries = []
        for fast_group, slow_group in zip(self.base_optimizer.param_groups, self.param_groups):
            for fast_p, slow_p in zip(fast_group["params"], slow_group["params"]):
                _lib.require_gpu(fast_p)
                if fast_p.dtype != torch.float32 or slow_p.dtype != torch.float32 or not fast_p.is_contiguous() or not slow_p.is_contiguous():
                    raise RuntimeError("Lookahead (HIP) expects contiguous fp32 parameters")
                entries.append({"p": fast_p.data, "g": None, "m": slow_p.data, "s": None, "smax": None, "group": 0, "tensor": len(entries)})
        if not entries:
            return
        dev = entries[0]["p"].device
        tabs = getattr(self, "_hc_tabs", None)
        if tabs is None:
            tabs = self._hc_tabs = DeviceTables()
        raw, n = chunk_rows(entries)
        chunks = tabs.get("chunks", raw, dev)
        check(_lib.load().hc_lookahead_sync(ptr(chunks), n, float(sync_rate), stream()), "hc_lookahead_sync")
        self._hc_keep = chunks
        bump_weights_epoch()


class Scout(Lookahead):
    """Lookahead whose synchronisation rate follows the coherence of the last ``sync_period`` updates (wrapper.py:137-283)."""

    def __init__(self, base_optimizer: torch.optim.Optimizer, sync_rate: float = 0.5, sync_period: int = 6) -> None:
        super().__init__(base_optimizer, sync_rate, sync_period)
        self.buffer = [p.data.unsqueeze(0) for group in self.param_groups for p in group["params"]]

    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = self.base_optimizer.step(closure)
        self.fast_steps += 1
        idx = 0
        for group in self.base_optimizer.param_groups:
            for p in group["params"]:
                self.buffer[idx] = torch.cat((self.buffer[idx], p.data.clone().detach().unsqueeze(0)))
                idx += 1
        if self.fast_steps % self.defaults["sync_period"] == 0:
            # the trajectory statistics are the reference's own torch expressions (wrapper.py:218-225): a decision taken once
            # per sync_period on the host, not part of the per-step path
            update_similarity = []
            for _ in range(len(self.buffer)):
                p = self.buffer.pop()
                update = p[1:] - p[:-1]
                max_dev = (update - torch.mean(update, dim=0)).abs().max(dim=0).values
                update_similarity.append((torch.std(update, dim=0) / max_dev).mean().item())
            update_coherence = sum(update_similarity) / len(update_similarity)
            sync_rate = max(1 - update_coherence, self.defaults["sync_rate"])
            if sync_rate != sync_rate:      # NaN coherence (a parameter that did not move: 0 / 0): the reference's `if sync_rate > 0`
                sync_rate = 0.0             # (wrapper.py:270-283) is then false - no outer update, the fast weights go back to the slow ones
            self.sync_params(sync_rate)
            self.buffer = []
            for group in self.param_groups:
                for p in group["params"]:
                    self.buffer.append(p.data.unsqueeze(0))
        return loss
