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
from ._multi_tensor import build_chunks

__all__ = ["Lookahead", "Scout"]


class Lookahead(Optimizer):
    """k steps forward, 1 step back (wrapper.py:18-134)."""

    def __init__(self, base_optimizer: torch.optim.Optimizer, sync_rate: float = 0.5, sync_period: int = 6) -> None:
        if sync_rate < 0 or sync_rate > 1:
            raise ValueError(f"expected positive float lower than 1 as sync_rate, received: {sync_rate}")
        if not isinstance(sync_period, int) or sync_period < 1:
            raise ValueError(f"expected positive integer as sync_period, received: {sync_period}")
        self.defaults = {"sync_rate": sync_rate, "sync_period": sync_period}
        self.state = defaultdict(dict)
        self.base_optimizer = base_optimizer
        self.fast_steps = 0
        self.param_groups = []
        for group in self.base_optimizer.param_groups:
            self._add_param_group(group)

    def __getstate__(self) -> Dict[str, Any]:
        return {
            "defaults": self.defaults,
            "state": self.state,
            "base_state": self.base_optimizer.__getstate__(),
            "fast_steps": self.fast_steps,
            "param_groups": self.param_groups,
        }

    def state_dict(self) -> Dict[str, Any]:
        return dict(**super().state_dict(), base_state_dict=self.base_optimizer.state_dict())

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        self.base_optimizer.load_state_dict(state_dict["base_state_dict"])
        super().load_state_dict(state_dict)
        self.__setstate__({"base_state_dict": self.base_optimizer.state_dict()})

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.base_optimizer.zero_grad(set_to_none)

    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = self.base_optimizer.step(closure)
        self.fast_steps += 1
        if self.fast_steps % self.defaults["sync_period"] == 0:
            self.sync_params(self.defaults["sync_rate"])
        return loss

    def __repr__(self) -> str:
        format_string = self.__class__.__name__ + " ("
        optimizer_repr = self.base_optimizer.__repr__().replace("\n", "\n\t")
        format_string += f"\nbase_optimizer={optimizer_repr},"
        for arg, val in self.defaults.items():
            format_string += f"\n{arg}={val},"
        format_string += "\n)"
        return format_string

    def _add_param_group(self, param_group: Dict[str, Any]) -> None:
        """Adds a new slow parameter group (a detached copy of the fast weights)."""
        group = {"params": [p.clone().detach() for p in param_group["params"]], "lr": param_group["lr"]}
        for p in group["params"]:
            p.requires_grad = False
        self.param_groups.append(group)

    def add_param_group(self, param_group: Dict[str, Any]) -> None:
        """Adds a parameter group to the base optimizer (fast weights) and its slow counterpart."""
        self.base_optimizer.add_param_group(param_group)
        self._add_param_group(self.base_optimizer.param_groups[-1])

    def sync_params(self, sync_rate: float = 0.0) -> None:
        """slow_param <- slow_param + sync_rate * (fast_param - slow_param); fast_param <- slow_param (wrapper.py:121-134)."""
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
        host, n = build_chunks(entries)
        chunks = host.to(dev)
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
            self.sync_params(sync_rate)
            self.buffer = []
            for group in self.param_groups:
                for p in group["params"]:
                    self.buffer.append(p.data.unsqueeze(0))
        return loss
