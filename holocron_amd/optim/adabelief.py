"""AdaBelief on one multi-tensor HIP kernel (reference: holocron/optim/adabelief.py:16-167).

Follows the reference code, not the paper: epsilon is only added to the denominator (SURVEY.md
Q5) and weight decay is L2-into-gradient.  ``param_groups`` and ``state`` are re-discovered at
every ``step()`` so that ``Trainer._reset_opt`` (holocron/trainer/core.py:238-252) and per-batch LR
schedulers keep working; ``state[p]`` exposes ``step``, ``exp_avg``, ``exp_avg_sq``
[, ``max_exp_avg_sq``] like the reference.

Device-side protocol: the hyper-parameter block (one hc_adabelief_group per param group) lives in
HBM together with the step counter, which the kernel launch advances itself; the host re-uploads
the block (stream-ordered, from pinned memory) only when a hyper-parameter changed or the counters
diverged (state reset / load_state_dict).
"""
from typing import Callable, Optional

import numpy as np
import torch
from torch.optim import Adam

from .. import _lib
from .._lib import AdaBeliefGroup, check, ptr, stream
from ..ops.conv import bump_weights_epoch, flush_deferred_wgrads
from ._multi_tensor import Staging, VGroups, chunk_rows

__all__ = ["AdaBelief"]


class AdaBelief(Adam):
    """Same constructor as ``torch.optim.Adam`` (lr, betas, eps, weight_decay, amsgrad)."""

    # ---- host bookkeeping -----------------------------------------------------------------
    def _collect(self, advance_state: bool):
        plist, vg = [], VGroups()
        for gi, group in enumerate(self.param_groups):
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
                if advance_state:
                    state["step"] += 1
                plist.append((p, vg.index(gi, state["step"])))      # one launch group per (param group, step count)
        hyper, steps = [], []
        for gi, st in vg.keys:
            group = self.param_groups[gi]
            beta1, beta2 = group["betas"]
            hyper.append((float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                          float(group["weight_decay"]), int(bool(group["amsgrad"]))))
            steps.append(st)
        return plist, hyper, steps

    def _sync_groups(self, dev, hyper, steps_after):
        """Make the device block hold `hyper` and step = steps_after - 1 (the launch increments)."""
        ng = len(hyper)
        st = getattr(self, "_hc_gstage", None)
        nbytes = ng * _lib.C.sizeof(AdaBeliefGroup)
        if st is None or st.dev.numel() != nbytes or st.dev.device != dev:
            st = self._hc_gstage = Staging(nbytes, dev)
            self._hc_ghyper, self._hc_gsteps = None, None
        want = [s - 1 for s in steps_after]
        if self._hc_ghyper != hyper or self._hc_gsteps != want:
            gbuf = (AdaBeliefGroup * ng)()
            for g, (lr, b1, b2, eps, wd, ams), s in zip(gbuf, hyper, want):
                g.lr, g.beta1, g.beta2, g.eps, g.weight_decay, g.step, g.amsgrad = lr, b1, b2, eps, wd, s, ams
            st.upload(np.frombuffer(bytes(gbuf), dtype=np.uint8))
        self._hc_ghyper, self._hc_gsteps = hyper, list(steps_after)  # device value after this launch
        return st.dev

    def _table(self, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), gi) for p, gi in plist)
        cache = getattr(self, "_hc_table", None)
        if cache is not None and cache[0] == key:
            return cache[1].dev, cache[2]
        entries = []
        for ti, (p, gi) in enumerate(plist):
            st = self.state[p]
            entries.append({"p": p.data, "g": p.grad, "m": st["exp_avg"], "s": st["exp_avg_sq"],
                            "smax": st.get("max_exp_avg_sq"), "group": gi, "tensor": ti})
        raw, n = chunk_rows(entries)
        stage = cache[1] if (cache is not None and cache[1].dev.numel() == raw.size) else Staging(raw.size, plist[0][0].device)
        stage.upload(raw)
        self._hc_table = (key, stage, n)
        return stage.dev, n

    # ---- public API -------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flush_deferred_wgrads()     # weight gradients a backward pass only queued (normally flushed at its end)
        plist, hyper, steps = self._collect(advance_state=True)
        if not plist:
            return loss
        dev = plist[0][0].device
        chunks, n = self._table(plist)
        gdev = self._sync_groups(dev, hyper, steps)
        check(_lib.load().hc_adabelief_step(ptr(chunks), n, ptr(gdev), len(hyper), 1, stream()), "hc_adabelief_step")
        bump_weights_epoch()
        return loss

    def advance_for_replay(self) -> None:
        """Host-side half of ``step()`` for a training step that is replayed from a captured hipGraph:
        advances ``state[p]['step']`` and, if a scheduler changed a hyper-parameter, enqueues the new
        block on the current stream ahead of the replay.  The device counter advances by itself."""
        plist, hyper, steps = self._collect(advance_state=True)
        if plist:
            self._sync_groups(plist[0][0].device, hyper, steps)
        bump_weights_epoch()

    def rewind_after_capture(self) -> None:
        """``step()`` under stream capture advanced the host-side counters, but the launch was only
        recorded: take the host back by one so that it agrees with the device again."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and p.grad is not None:
                    st["step"] -= 1
        if getattr(self, "_hc_gsteps", None) is not None:
            self._hc_gsteps = [s - 1 for s in self._hc_gsteps]
