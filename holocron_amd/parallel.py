"""Data-parallel training: one process per GPU, gradients all-reduced over RCCL (xGMI).

The reference has no distributed code at all; this is the new functionality behind
the same single-process training loop: wrap the model's parameters in a ``GradReducer`` and call
``finalize()`` between ``loss.backward()`` and ``optimizer.step()``.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large buckets
(default 32 MiB) in reverse-registration order so that a bucket's all-reduce is issued from the
autograd thread as soon as its last gradient is produced and overlaps the rest of backward;
``torch.distributed`` runs the collective on RCCL's own stream and orders it with events.
BatchNorm statistics stay local to each rank, like running the reference on one device.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

__all__ = ["GradReducer", "init_process_group_from_env", "broadcast_parameters"]


def init_process_group_from_env(backend: Optional[str] = None) -> bool:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when launched by torchrun."""
    import os
    if dist.is_initialized():
        return True
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return True


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not dist.is_initialized():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], dtype: torch.dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, dtype=dtype, device=params[0].device)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.work = None


class GradReducer:
    """Bucketed, overlapped gradient averaging.

    ``comm_dtype=torch.bfloat16`` halves the bytes on the links (gradients are averaged in bf16,
    parameters and optimizer state stay fp32)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0,
                 comm_dtype: torch.dtype = torch.float32, group=None) -> None:
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self.buckets: List[_Bucket] = []
        self._of = {}
        self._hooks = []
        if self.world > 1:
            self._build(bucket_mb)

    def _build(self, bucket_mb: float) -> None:
        cap = int(bucket_mb * 1024 * 1024 / torch.empty((), dtype=self.comm_dtype).element_size())
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in reversed(self.params):      # backward produces the last layers' gradients first
            if cur and size + p.numel() > cap:
                self.buckets.append(_Bucket(cur, self.comm_dtype))
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, self.comm_dtype))
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._of[p] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b, i = self._of[p]
        b.views[i].copy_(p.grad)
        b.pending -= 1
        if b.pending == 0:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finalize(self) -> None:
        """Wait for the collectives and write the averaged gradients back into ``p.grad``."""
        if self.world <= 1:
            return
        inv = 1.0 / self.world
        for b in self.buckets:
            if b.pending != 0:   # parameters that received no gradient this step: reduce what we have
                for v, p in zip(b.views, b.params):
                    if p.grad is None:
                        v.zero_()
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            b.work.wait()
            for v, p in zip(b.views, b.params):
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(v).mul_(inv) if p.grad.dtype == v.dtype else p.grad.copy_(v.to(p.grad.dtype) * inv)
            b.pending = len(b.params)
            b.work = None

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
