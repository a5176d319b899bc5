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
import ctypes as C
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

__all__ = ["GradReducer", "GraphedStep", "BackwardCut", "init_process_group_from_env", "broadcast_parameters"]


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


def ensure_ranks(gpus: int, script: str, argv: List[str]) -> None:
    """`script --gpus N` must measure N ranks, wrapper or not (bench.py and the scripts/bench_*.py drivers call this first).

    * launched by torchrun (WORLD_SIZE set): WORLD_SIZE has to equal ``gpus``, anything else is a mis-launch and raises;
    * ``gpus == 1`` without a launcher: returns, the caller runs single-process;
    * ``gpus > 1`` without a launcher: this process becomes the launcher - it runs
      ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv...``
      (one rank per GPU, the driver's own command form), forwards its exit status and never returns.

    The reference trains on one device only (trainer/core.py:90-104), so none of this has a counterpart there."""
    import os
    import socket
    import subprocess
    import sys
    if gpus < 1:
        raise SystemExit(f"--gpus must be >= 1 (got {gpus})")
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit(f"--gpus {gpus} disagrees with WORLD_SIZE={ws}: launch with --nproc-per-node {gpus} (or drop the launcher, "
                             f"`{os.path.basename(script)} --gpus {gpus}` starts its own ranks)")
        return
    if gpus == 1:
        return
    if torch.cuda.is_available() and torch.cuda.device_count() < gpus:
        raise SystemExit(f"--gpus {gpus} but only {torch.cuda.device_count()} device(s) visible")
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL fails with the legacy mode)
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not dist.is_initialized():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], flat: torch.Tensor):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = flat                      # a contiguous slice of the reducer's communication buffer
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.ready = False
        self.work = None


def _conv_ops():
    from .ops import conv
    return conv


_HIP_COPY_DTYPES = (torch.float32, torch.bfloat16)


def _copy_pieces(items, piece: int, per_launch: int):
    """Launch plan of ``hc_multi_copy``: ``items`` = (src pointer, dst pointer, elements, src element size, dst element size) per
    tensor -> list of launches, each a list of at most ``per_launch`` (src pointer, dst pointer, elements) pieces of at most
    ``piece`` elements, tensors split in order (so that every piece is swept by the same 16 workgroups)."""
    launches, cur = [], []
    for sp, dp, n, se, de in items:
        for o in range(0, n, piece):
            cur.append((sp + o * se, dp + o * de, min(piece, n - o)))
            if len(cur) == per_launch:
                launches.append(cur)
                cur = []
    if cur:
        launches.append(cur)
    return launches


def _hip_copy_all(dst: List[torch.Tensor], src: List[torch.Tensor], scale: float) -> bool:
    """The same on one MI355X through ``hc_multi_copy`` (csrc/optim.hip): up to 64 pieces of at most 256 K elements per launch, the
    piece table in the kernel arguments - so a bucket of RepVGG-A0 (24.7 M elements, ~300 pieces) is five launches that stream at
    HBM rate instead of ``torch._foreach_copy_``'s chunked kernels plus a separate scaling pass.  False when the lists do not
    qualify (other devices / dtypes / layouts): the caller takes the torch path."""
    d0, s0 = dst[0], src[0]
    if not d0.is_cuda or d0.dtype not in _HIP_COPY_DTYPES or s0.dtype not in _HIP_COPY_DTYPES:
        return False
    for d, t in zip(dst, src):
        if (d.device != d0.device or t.device != d0.device or d.dtype != d0.dtype or t.dtype != s0.dtype
                or d.shape != t.shape or not d.is_contiguous() or not t.is_contiguous()):
            return False
    from . import _lib
    from .ops import conv as _cv
    lib, st = _lib.load(), _cv.stream()
    desc = _lib.MultiCopyDesc()
    desc.src_bf16, desc.dst_bf16, desc.scale = int(s0.dtype == torch.bfloat16), int(d0.dtype == torch.bfloat16), float(scale)
    se, de = s0.element_size(), d0.element_size()
    items = [(t.data_ptr(), d.data_ptr(), d.numel(), se, de) for d, t in zip(dst, src)]
    for pieces in _copy_pieces(items, _lib.HC_MULTI_COPY_PIECE, _lib.HC_MULTI_COPY_MAX):
        for k, (sp, dp, n) in enumerate(pieces):
            desc.src[k], desc.dst[k], desc.n[k] = sp, dp, n
        desc.nitems = len(pieces)
        _cv.check(lib.hc_multi_copy(C.byref(desc), st), "hc_multi_copy")
    return True


def _copy_all(dst: List[torch.Tensor], src: List[torch.Tensor], scale: float = 1.0) -> None:
    """dst[i] <- scale * src[i] (casting) in as few launches as the tensors allow."""
    if not dst:
        return
    if dst[0].is_cuda and _hip_copy_all(dst, src, scale):
        return
    try:
        torch._foreach_copy_(dst, src)
    except RuntimeError as e:
        # the one refusal that has a per-tensor answer: tensor lists the fused path does not take (mixed devices / dtypes /
        # layouts).  Anything else - a launch failure, an out-of-memory, a shape mismatch - is not ours to hide.
        msg = str(e)
        if "foreach" not in msg and "same device" not in msg and "same dtype" not in msg:
            raise
        for d, s in zip(dst, src):
            d.copy_(s)
    if scale != 1.0:
        torch._foreach_mul_(dst, scale)


class GradReducer:
    """Bucketed gradient averaging over one flat communication buffer.

    Two ways to drive it:

    * overlapped (default): post-accumulate hooks count a bucket's gradients; when the last one
      arrives the bucket is packed (one fused cast-copy) and its all-reduce is issued from the autograd
      thread, so it runs behind the rest of backward.  ``finalize()`` waits, unpacks and scales.
    * deferred (``overlap=False``): no hooks.  ``pack()`` / ``reduce()`` / ``unpack()`` are three
      separate calls so that a training step replayed from hipGraphs can keep ``pack`` at the end of
      the captured backward and ``unpack`` in front of the captured optimizer step, with the single
      whole-buffer collective issued eagerly between the two replays (RCCL stays outside the
      graphs).  ``finalize()`` runs the three in order.

    ``comm_dtype=torch.bfloat16`` halves the bytes on the links (gradients are summed in bf16,
    parameters and optimizer state stay fp32).  ``force=True`` builds the buckets even for a group of
    one rank (used to exercise the path on a single GPU)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0,
                 comm_dtype: torch.dtype = torch.float32, group=None, overlap: bool = True,
                 force: bool = False, new_bucket_at: Optional[Iterable[torch.nn.Parameter]] = None,
                 materialize_missing: bool = True) -> None:
        self.group = group
        self._sync = True                    # False inside no_sync(): hooks do not count, nothing is launched
        # A parameter without a gradient on this rank may have one on another: after the all-reduce every rank must hold the
        # same (averaged) gradient or the replicas drift apart, so by default missing gradients are created from the reduced
        # buffer.  The price: a parameter that is unused on EVERY rank gets a zero gradient where a single process would leave
        # it None, and an optimizer then applies weight decay / momentum to it.  Pass materialize_missing=False when unused
        # parameters are unused on all ranks (they then stay None, single-process semantics).
        self.materialize_missing = materialize_missing
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self.overlap = overlap
        self.active = (self.world > 1 or force) and dist.is_initialized() and len(self.params) > 0
        self.buckets: List[_Bucket] = []
        self.flat: Optional[torch.Tensor] = None
        self._of = {}
        self._hooks = []
        self._next = 0                       # first bucket whose collective has not been issued yet
        # parameters (walking in reverse registration order) at which a new bucket must begin: lets a bucket
        # end exactly where a backward segment ends (BackwardCut / GraphedStep)
        self._starts = {id(p) for p in (new_bucket_at or ())}
        if self.active:
            self._build(bucket_mb)

    def _build(self, bucket_mb: float) -> None:
        cap = int(bucket_mb * 1024 * 1024 / torch.empty((), dtype=self.comm_dtype).element_size())
        groups: List[List[torch.nn.Parameter]] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in reversed(self.params):      # backward produces the last layers' gradients first
            if cur and (size + p.numel() > cap or id(p) in self._starts):
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            groups.append(cur)
        # every bucket starts on a 256-byte boundary of the buffer (RCCL's vectorised paths, and a slice handed to a
        # collective on its own keeps the alignment of the whole); the padding stays zero on every rank
        align = 256 // torch.empty((), dtype=self.comm_dtype).element_size()
        sizes = [sum(p.numel() for p in g) for g in groups]
        starts, off = [], 0
        for n in sizes:
            starts.append(off)
            off += (n + align - 1) // align * align
        self.flat = torch.zeros(max(off, 1), dtype=self.comm_dtype, device=self.params[0].device)
        self._starts_of = starts + [off]
        for g, n, st in zip(groups, sizes, starts):
            self.buckets.append(_Bucket(g, self.flat[st:st + n]))
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._of[p] = (b, i)
        _conv_ops().mark_reducer_managed(self.params)      # every path that reads these gradients goes through _pack_bucket
        if self.overlap:
            self.set_overlap(True)

    def set_overlap(self, overlap: bool) -> None:
        """Switch between the hook-driven (overlapped) and the deferred mode."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.overlap = overlap
        if overlap and self.active:
            self._next = 0
            for b in self.buckets:
                b.pending, b.ready = len(b.params), False
                for p in b.params:
                    h = p.register_post_accumulate_grad_hook(self._on_grad)
                    _conv_ops().register_flush_aware_hook(p, h)     # _pack_bucket flushes the deferred weight gradients first
                    self._hooks.append(h)

    # ---- bucket-level pieces ----------------------------------------------------------------
    @staticmethod
    def _pack_bucket(b: _Bucket) -> None:
        from .ops import conv as _cv
        _cv.flush_deferred_wgrads()          # gradients that a backward node has only queued so far (fused RepBlock wgrad)
        dst, src = [], []
        for v, p in zip(b.views, b.params):
            if p.grad is None:               # no gradient this step: contributes zeros
                v.zero_()
            else:
                dst.append(v)
                src.append(p.grad)
        _copy_all(dst, src)

    def _unpack_bucket(self, b: _Bucket) -> None:
        grads, views = [], []
        for p, v in zip(b.params, b.views):
            if p.grad is None:
                if not self.materialize_missing:
                    continue
                p.grad = torch.empty_like(p)
            grads.append(p.grad)
            views.append(v)
        if not grads:
            return
        _copy_all(grads, views, 1.0 / self.world)

    def _launch(self, b: _Bucket) -> None:
        self._pack_bucket(b)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate into ``p.grad`` (no hook
        counts, no collective); the first backward outside it - or ``finalize()`` - reduces the accumulated gradients."""
        red = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev = red._sync
                red._sync = False

            def __exit__(self_inner, *exc):
                red._sync = self_inner.prev
                return False
        return _NoSync()

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._sync:
            return
        b, _ = self._of[p]
        if b.work is not None or b.pending <= 0:
            # a second backward before finalize(): the bucket's collective already ran on the first micro-batch's gradients and
            # finalize() would overwrite the accumulated ones with it
            raise RuntimeError("GradReducer (overlap mode) saw a second backward pass before finalize(): wrap the accumulation "
                               "micro-steps in `with reducer.no_sync():` (all but the last), or build it with overlap=False")
        b.pending -= 1
        if b.pending == 0:
            b.ready = True
            # collectives are issued strictly in bucket order: ranks whose graphs produce gradients
            # in different orders (or not at all for some parameters) still pair the same buffers
            while self._next < len(self.buckets) and self.buckets[self._next].ready:
                self._launch(self.buckets[self._next])
                self._next += 1

    # ---- segment-wise use (GraphedStep) --------------------------------------------------------
    def buckets_with_all_grads(self, exclude=()) -> List[int]:
        """Indices of the buckets (not in ``exclude``) all of whose parameters hold a gradient right now."""
        return [i for i, b in enumerate(self.buckets)
                if i not in exclude and all(p.grad is not None for p in b.params)]

    def spans(self, indices: Iterable[int]) -> List[torch.Tensor]:
        """The buckets ``indices`` as maximal contiguous slices of the communication buffer."""
        out: List[torch.Tensor] = []
        offs = self._starts_of            # bucket i occupies [offs[i], offs[i + 1]) including its tail padding
        run = None
        for i in sorted(indices):
            if run is not None and run[1] == i:
                run[1] = i + 1
            else:
                if run is not None:
                    out.append(self.flat[offs[run[0]]:offs[run[1]]])
                run = [i, i + 1]
        if run is not None:
            out.append(self.flat[offs[run[0]]:offs[run[1]]])
        return out

    # ---- deferred mode ------------------------------------------------------------------------
    @torch.no_grad()
    def pack(self) -> None:
        """Gradients -> communication buffer (capturable: plain device copies)."""
        if self.active:
            for b in self.buckets:
                self._pack_bucket(b)

    def reduce(self) -> None:
        """One all-reduce over the whole buffer, ordered after the current stream's work; the
        current stream waits for it (no host synchronisation)."""
        if self.active:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def unpack(self) -> None:
        """Communication buffer / world -> gradients (capturable)."""
        if self.active:
            for b in self.buckets:
                self._unpack_bucket(b)

    # ---- one-call form ----------------------------------------------------------------------------
    @torch.no_grad()
    def finalize(self) -> None:
        """Make every ``p.grad`` the average over ranks (call between backward and the optimizer)."""
        if not self.active:
            return
        if not self.overlap:
            self.pack()
            self.reduce()
            self.unpack()
            return
        for b in self.buckets[self._next:]:   # buckets some of whose parameters received no gradient
            self._launch(b)
        self._next = 0
        for b in self.buckets:
            b.work.wait()
            self._unpack_bucket(b)
            b.pending = len(b.params)
            b.ready = False
            b.work = None

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.active:
            _conv_ops().mark_reducer_managed(self.params, False)


class BackwardCut:
    """Cuts the autograd graph at the input of ``module`` so that backward can run in two pieces.

    A forward pre-hook replaces the module's input ``h`` by ``h.detach().requires_grad_()``.  ``loss.backward()``
    then stops there (gradients of ``module`` and everything after it, plus d loss / d h);
    ``continue_backward()`` pushes that gradient through the part of the network in front of ``module``.
    Between the two calls the gradients of the rear part are complete: a GraphedStep uses the gap to start
    their all-reduce while the front part's backward is still to run (RepVGG-A0: the last block and the
    head hold 66 % of the parameters and their gradients are the first ones backward produces)."""

    def __init__(self, module: torch.nn.Module) -> None:
        self.pair = None
        self._handle = module.register_forward_pre_hook(self._pre)

    def _pre(self, module, args):
        h = args[0]
        if not (torch.is_grad_enabled() and isinstance(h, torch.Tensor) and h.requires_grad):
            self.pair = None
            return None
        hd = h.detach().requires_grad_(True)
        for attr in ("_hc_stats",):           # side information the fused blocks hand to their consumer
            if hasattr(h, attr):
                setattr(hd, attr, getattr(h, attr))
        self.pair = (h, hd)
        return (hd,) + tuple(args[1:])

    def continue_backward(self) -> None:
        if self.pair is None:
            return
        h, hd = self.pair
        self.pair = None
        if hd.grad is not None:
            h.backward(hd.grad)

    def remove(self) -> None:
        self._handle.remove()


class GraphedStep:
    """A whole training step replayed from hipGraphs, with the gradient all-reduce kept outside them.

    A RepVGG-A0 step is ~600 kernel launches of 5-100 us each: issued one by one the host is the
    bottleneck, replayed from a graph it is not.  On one rank the step is one graph
    (``fwd_bwd`` + ``optimizer.step``).  With an active ``GradReducer`` the collectives run eagerly
    between graphs, so that RCCL never executes under stream capture::

        graph 0:  zero_grad, forward, loss, backward [up to a BackwardCut], pack the complete buckets
        eager  :  async all-reduce of those buckets (RCCL's stream, ordered against ours by events)
        graph 1:  [rest of backward], pack the remaining buckets        <- runs while RCCL reduces the first ones
        eager  :  all-reduce of the remaining buckets; our stream waits for all of them (no host wait)
        graph B:  reducer.unpack(), optimizer.step()

    ``fwd_bwd`` is one callable (no overlap: one collective after the whole backward) or a list of segment
    callables, e.g. ``[fwd_and_loss_backward, cut.continue_backward]``.  The first segment must start from
    ``zero_grad(set_to_none=True)``: which buckets a segment completes is read off ``p.grad is not None`` when
    the segment is captured.  Segments must work on fixed input buffers; ``optimizer`` is one of the
    multi-tensor HIP optimizers (``advance_for_replay`` does the host half of ``step``).  ``capture()`` runs
    one eager step on a side stream first (lazy allocations, autograd warm-up).
    """

    def __init__(self, fwd_bwd, optimizer, reducer: Optional["GradReducer"] = None,
                 capture_error_mode: Optional[str] = None) -> None:
        self.segments = list(fwd_bwd) if isinstance(fwd_bwd, (list, tuple)) else [fwd_bwd]
        self.optimizer = optimizer
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        # With a process group alive, its watchdog thread polls the events of outstanding collectives
        # (hipEventQuery); under a "global"-mode capture that call is illegal from ANY thread and takes the
        # process down (seen on the MI355X box).  A thread-local capture restricts only the capturing thread,
        # and _quiesce() additionally lets the watchdog retire finished collectives before a capture begins.
        if capture_error_mode is None:
            capture_error_mode = "thread_local" if dist.is_initialized() else "global"
        self.capture_error_mode = capture_error_mode
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.spans: List[List[torch.Tensor]] = []      # per segment graph: the slices reduced after it
        self.final: Optional[torch.cuda.CUDAGraph] = None
        self._works: list = []                         # collectives issued since the last _quiesce()
        self._tracking = False
        self._holds_pool = False                       # counted in POOL.graph_users (see release)
        if self.reducer is not None:
            self.reducer.set_overlap(False)

    def fwd_bwd(self) -> None:
        for seg in self.segments:
            seg()

    def _ready_after(self, i: int, packed: set) -> set:
        red = self.reducer
        if i == len(self.segments) - 1:
            return set(range(len(red.buckets))) - packed
        return set(red.buckets_with_all_grads(packed))

    def eager(self) -> None:
        """The same step without graphs (also the warm-up of ``capture``): with several segments the early
        buckets' collectives still run behind the later segments."""
        red = self.reducer
        if red is None:
            self.fwd_bwd()
            self.optimizer.step()
            return
        packed, works = set(), []
        for i, seg in enumerate(self.segments):
            seg()
            ready = self._ready_after(i, packed)
            with torch.no_grad():
                for bi in sorted(ready):
                    red._pack_bucket(red.buckets[bi])
            packed |= ready
            works += self._reduce_spans(red.spans(ready))
        for w in works:
            w.wait()
        red.unpack()
        self.optimizer.step()

    def _quiesce(self) -> None:
        """Nothing of ours is in flight when a capture begins: the device is idle and every collective this object issued says so
        through its own Work handle (``is_completed`` = an event query from THIS thread, legal outside capture).  The process group's
        watchdog thread also polls those events; that only matters to a "global"-mode capture (where hipEventQuery from any thread
        is an error): the default with a live process group is "thread_local", which needs no settling time at all.  In global
        mode the watchdog drops a finished collective within one of its 100 ms polling periods."""
        torch.cuda.synchronize()
        works, self._works = self._works, []
        for w in works:
            while not w.is_completed():          # cannot spin after the synchronize above unless the backend lags behind the device
                import time
                time.sleep(0.001)
        if dist.is_initialized() and self.capture_error_mode == "global":
            import time
            time.sleep(0.3)

    def _reduce_spans(self, spans):
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.reducer.group, async_op=True) for t in spans]
        if self._tracking:                       # inside capture(): _quiesce() checks these handles; plain steps keep none
            self._works += works
        return works

    def capture(self) -> None:
        self._tracking = True
        try:
            self._capture()
        finally:
            self._tracking = False
            self._works = []

    def _capture(self) -> None:
        if not self._holds_pool:
            from .nn.repblock_op import POOL
            POOL.graph_users += 1
            self._holds_pool = True
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.eager()
        cur.wait_stream(side)
        self._quiesce()
        mode = self.capture_error_mode
        if self.reducer is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=mode):
                self.fwd_bwd()
                self.optimizer.step()
            self.graphs, self.spans, self.final = [g], [[]], None
        else:
            red = self.reducer
            graphs, spans, packed, pool = [], [], set(), None
            for i, seg in enumerate(self.segments):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                    seg()
                    ready = self._ready_after(i, packed)
                    with torch.no_grad():
                        for bi in sorted(ready):
                            red._pack_bucket(red.buckets[bi])
                pool = g.pool() if pool is None else pool
                packed |= ready
                graphs.append(g)
                spans.append(red.spans(ready))
                for w in self._reduce_spans(spans[-1]):   # keeps the ranks' collective sequences identical
                    w.wait()
                self._quiesce()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=pool, capture_error_mode=mode):
                red.unpack()
                self.optimizer.step()
            self.graphs, self.spans, self.final = graphs, spans, gb
        rewind = getattr(self.optimizer, "rewind_after_capture", None)
        if rewind is not None:           # the captured optimizer launch did not execute
            rewind()
        torch.cuda.synchronize()

    def release(self) -> None:
        self.graphs, self.spans, self.final = [], [], None
        from .nn.repblock_op import POOL
        if self._holds_pool:             # statistics arenas that were only kept alive for captured graphs are freed when the LAST
            self._holds_pool = False     # live GraphedStep lets go: another one may still replay against them
            POOL.release_retired()

    def run(self) -> None:
        if not self.graphs:
            self.eager()
            return
        advance = getattr(self.optimizer, "advance_for_replay", None)
        if advance is not None:
            advance()
        works = []
        for g, spans in zip(self.graphs, self.spans):
            g.replay()
            if spans:
                works += self._reduce_spans(spans)
        for w in works:
            w.wait()                     # stream-side wait: the host does not block
        if self.final is not None:
            self.final.replay()
