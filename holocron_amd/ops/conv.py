"""Host side of the MFMA convolution kernels: descriptor construction and thin launch wrappers.

Activations are logically NCHW ``torch.Tensor``s in bf16 with ``torch.channels_last`` strides,
i.e. NHWC in HBM — the layout the kernels are written for.  Weights stay fp32 OIHW at the API
surface (reference ``state_dict`` layout, SURVEY.md §5 checkpoint row) and are re-packed to the
kernel layout lazily (cached on parameter version + optimizer epoch).
"""
import ctypes as C
import functools
import os

import torch

from .. import _lib
from .._lib import ConvDesc, ConvSmallDesc, RepWgradDesc, WgradDesc, WgradGroupDesc, check, ptr, stream, tap

_WEIGHTS_EPOCH = [0]  # bumped by holocron_amd.optim after every raw-pointer parameter update


def bump_weights_epoch():
    """Invalidate every cached packed-weight image (public as ``holocron_amd.bump_weights_epoch``).

    The conv units cache the bf16 images of their fp32 master weights keyed on ``Parameter._version`` and on this epoch.  torch's own
    in-place ops on a parameter bump ``_version``; the HIP optimizers write through raw pointers and call this after every step.
    Code that writes through ``p.data`` (``p.data.add_``, ``.data.copy_``, ``.data.clamp_``: older third-party optimizers, EMA weight
    swaps, weight clipping) does neither - call this after such an update, or the next forward multiplies with the old weights."""
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


def is_cl_bf16(x):
    return (x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_cuda
            and x.is_contiguous(memory_format=torch.channels_last)
            and (x.stride(1) == 1 or x.shape[1] == 1))


def to_cl_bf16(x):
    """Logical NCHW tensor -> bf16, NHWC in memory."""
    _lib.require_gpu(x)
    if is_cl_bf16(x):
        return x
    N, Cc, H, W = x.shape
    if x.dtype == torch.float32 and x.is_contiguous():
        out = torch.empty((N, Cc, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        check(_lib.load().hc_nchw_to_nhwc_bf16(ptr(x), ptr(out), N, Cc, H, W, Cc, stream()), "hc_nchw_to_nhwc_bf16")
        return out
    out = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if out.stride(1) != 1:  # C == 1 or degenerate: force dense NHWC
        out = out.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return out


def empty_cl(N, Cc, H, W, device):
    return torch.empty((N, Cc, H, W), dtype=torch.bfloat16, device=device, memory_format=torch.channels_last)


def conv_out_size(H, k, s, p):
    return (H + 2 * p - k) // s + 1


# ------------------------------------------------------------------ descriptors
def _fill_fwd_class(cl, OH, OW, KH, KW, stride, pad, tap0=0):
    cl.OHg, cl.OWg, cl.oy0, cl.ox0, cl.ostep, cl.istep = OH, OW, 0, 0, 1, stride
    n = 0
    for kh in range(KH):
        for kw in range(KW):
            cl.tap[n] = tap(kh - pad, kw - pad, 0, tap0 + kh * KW + kw)
            n += 1
    cl.ntaps = n


def fwd_desc(N, Cin, H, W, Cout, KH, KW, stride, pad, T=None, tap0=0):
    """Descriptor of a forward conv (pointers left NULL)."""
    if KH * KW > _lib.HC_MAX_TAPS:
        raise ValueError("kernel too large for the gather-conv tap table")
    d = ConvDesc()
    OH, OW = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
    d.N, d.IH, d.IW, d.srcC = N, H, W, Cin
    d.OH, d.OW, d.Cout = OH, OW, Cout
    d.T = KH * KW if T is None else T
    d.nclass = 1
    _fill_fwd_class(d.cls[0], OH, OW, KH, KW, stride, pad, tap0)
    return d


def dgrad_desc(N, Cin, H, W, Cout, branches, stride):
    """Descriptor of the data gradient dx[N,H,W,Cin] of one or two convs that share input/stride.

    ``branches``: list of (KH, KW, pad, src_index, tap0) — the weights of all branches are packed
    (mode 1) into one tensor [Cin][T][Cout], branch b at taps [tap0, tap0+KH*KW)."""
    if stride not in (1, 2):
        raise NotImplementedError("gather-conv data gradient supports stride 1 and 2")
    d = ConvDesc()
    KH0, KW0, pad0 = branches[0][0], branches[0][1], branches[0][2]
    OH, OW = conv_out_size(H, KH0, stride, pad0), conv_out_size(W, KW0, stride, pad0)
    d.N, d.IH, d.IW, d.srcC = N, OH, OW, Cout      # source = dy
    d.OH, d.OW, d.Cout = H, W, Cin                  # destination = dx
    d.T = sum(b[0] * b[1] for b in branches)
    d.nclass = stride * stride
    for py in range(stride):
        for px in range(stride):
            cl = d.cls[py * stride + px]
            cl.OHg = (H - py + stride - 1) // stride
            cl.OWg = (W - px + stride - 1) // stride
            cl.oy0, cl.ox0, cl.ostep, cl.istep = py, px, stride, 1
            n = 0
            for (KH, KW, pad, src, tap0) in branches:
                for kh in range(KH):
                    if (py + pad - kh) % stride:
                        continue
                    for kw in range(KW):
                        if (px + pad - kw) % stride:
                            continue
                        wt = tap0 + (KH - 1 - kh) * KW + (KW - 1 - kw)
                        cl.tap[n] = tap((py + pad - kh) // stride, (px + pad - kw) // stride, src, wt)
                        n += 1
            cl.ntaps = n
    return d


PROFILE = None  # bench.py sets this to a list: (family, algorithmic flops, start event, end event, algorithmic bytes)
PROFILE_TAGS = None  # scripts/layer_table.py: a list that receives one shape tag per PROFILE entry, in the same order


def _tag(text):
    if PROFILE_TAGS is not None:
        PROFILE_TAGS.append(text)


class profiled:
    """``with profiled(family, flops, nbytes): <launches>`` - when bench.py's instrumented step is running (``PROFILE`` is a list),
    brackets the launches with HIP events on the launch stream and books them under ``family`` with their ALGORITHMIC flops and
    bytes; otherwise free.  Used for the launches the conv wrappers below do not time themselves (BatchNorm / elementwise passes,
    packing, pooling, optimizer)."""

    __slots__ = ("family", "flops", "nbytes", "e0")

    def __init__(self, family, flops=0.0, nbytes=0.0):
        self.family, self.flops, self.nbytes, self.e0 = family, flops, nbytes, None

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.family, self.flops, self.e0, e1, self.nbytes))
            _tag("%s %.1fMB" % (self.family, self.nbytes / 1e6))
        return False


def _desc_flops(d):
    macs = 0
    for c in range(d.nclass):
        cl = d.cls[c]
        macs += d.N * cl.OHg * cl.OWg * cl.ntaps
    return 2.0 * macs * d.srcC * d.Cout


def launch_conv(d, src0, wpk, dst, src1=None, resid=None, stats=None, bias=None, act=0, flops=None, dst2=None, stats2=None,
                co_split=0, ch_scale=None, act_slope=0.0, resid_after_act=False):
    """``dst2`` / ``stats2`` / ``co_split``: two convolutions of ``src0`` in one launch (stacked weight rows, hc_conv_desc.co_split).
    ``ch_scale`` (+ ``bias``, ``act``, ``act_slope``, ``resid_after_act``): the inference epilogue of a conv -> BatchNorm -> activation
    unit (running statistics folded into a per-channel scale and shift)."""
    d.src0, d.src1, d.wpk, d.dst = ptr(src0), ptr(src1), ptr(wpk), ptr(dst)
    d.resid, d.stats, d.bias, d.act = ptr(resid), ptr(stats), ptr(bias), act
    d.dst2, d.stats2, d.co_split = ptr(dst2), ptr(stats2), co_split
    d.ch_scale, d.act_slope, d.resid_after_act = ptr(ch_scale), float(act_slope), 1 if resid_after_act else 0
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().hc_conv_gather(C.byref(d), stream()), "hc_conv_gather")
        e1.record()
        nbytes = sum(t.numel() * t.element_size() for t in (src0, src1, wpk, dst, resid, dst2) if t is not None)
        PROFILE.append(("conv_gather", _desc_flops(d) if flops is None else flops, e0, e1, nbytes))
        _tag("%s N%d %d@%dx%d -> %d@%dx%d taps%d%s%s" % ("fwd" if d.nclass == 1 and d.cls[0].istep >= 1 and stats is not None else "conv", d.N, d.srcC,
                                                      d.IH, d.IW, d.Cout, d.OH, d.OW, d.cls[0].ntaps, " cls%d" % d.nclass if d.nclass > 1 else "",
                                                      " +resid" if resid is not None else ""))
        return
    check(_lib.load().hc_conv_gather(C.byref(d), stream()), "hc_conv_gather")


# ------------------------------------------------------------------ weight packing
ROWS_IMAGE = 4   # HC_CONV_SMALL_ROWS_IMAGE: hc_conv_small reads the row-unit weight image (pack modes 3 / 4)


def rows_image(Cc, device):
    """Destination of pack modes 3 / 4: [10 * ceil(C / 32) k32-steps][C rows][32] bf16 (3x3 taps at tap0 = 0, the 1x1 at tap0 = 9),
    zero-filled: with C = 48 the second k32 step of every tap is half padding."""
    return torch.zeros((10 * ((Cc + 31) // 32), Cc, 32), dtype=torch.bfloat16, device=device)


def pack_weight(w, mode, out=None, tap0=0, T=None):
    """fp32 OIHW -> packed bf16 (mode 0 fwd [Cout][T][Cin], mode 1 dgrad [Cin][T][Cout], modes 3 / 4: forward / data-gradient
    row-unit image, ``out`` = rows_image(C))."""
    Cout, Cin, KH, KW = w.shape
    T = KH * KW if T is None else T
    if out is None:
        if mode >= 3:
            raise ValueError("pack modes 3 / 4 write into a rows_image() shared by the 3x3 and the 1x1 kernel: pass out=")
        shape = (Cout, T, Cin) if mode == 0 else (Cin, T, Cout)
        out = torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    wc = w.detach()
    if wc.dtype != torch.float32 or not wc.is_contiguous():
        wc = wc.float().contiguous()
    check(_lib.load().hc_pack_conv_weight(ptr(wc), ptr(out), Cout, Cin, KH, KW, mode, tap0, T, stream()),
          "hc_pack_conv_weight")
    return out


def pack_weight_im2col(w, Kpad, k0=0):
    """fp32 OIHW (tiny Cin) -> bf16 [Cout][1][Kpad] with k = k0 + (kh*KW+kw)*Cin + ci."""
    Cout, Cin, KH, KW = w.shape
    flat = w.detach().float().permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin)
    out = torch.zeros((Cout, 1, Kpad), dtype=torch.bfloat16, device=w.device)
    out[:, 0, k0:k0 + flat.shape[1]] = flat.to(torch.bfloat16)
    return out


class PackCache:
    """Caches packed weights of one module; invalidated by in-place updates of the parameters
    (torch version counter) or by our raw-pointer optimizers (weights epoch)."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, params, builder):
        key = tuple((p.data_ptr(), p._version) for p in params) + (weights_epoch(),)
        if key != self._key:
            self._val = builder()
            self._key = key
        return self._val


# ------------------------------------------------------------------ weight gradient
class _WgradSide:
    """Weight gradients on a second HIP stream (opt-in: ``set_wgrad_side_stream(True)``).

    In backward a block's weight gradients feed nothing but the optimizer, while its data gradient is on
    the critical path to the previous block.  With the side stream on, ``side_stream_for_wgrad`` forks a
    second stream after the block's dy tensors exist, the weight-gradient kernels (and their split-K
    reductions) are issued there and run next to the main chain's BN passes and data gradients; one autograd
    final callback per backward joins the side stream back into the caller's stream, so everything after
    ``backward()`` (reducer, optimizer, clipping) sees finished gradients.  Captured in a hipGraph the fork /
    join become a parallel branch of the graph.

    Safe by construction rather than by luck: it is used only when the parameters have no ``.grad`` yet
    (autograd then adopts the returned tensor without launching anything on the main stream), and the join
    callback checks that this adoption really happened and raises otherwise.  Inputs are
    ``record_stream``-ed so that the allocator does not hand their memory out while the side stream reads it."""

    def __init__(self):
        self.on = False
        self.streams = {}
        self.pending = False
        self.adopted = []      # (parameter, data_ptr of the gradient computed on the side stream)

    def stream_of(self, device):
        s = self.streams.get(device.index)
        if s is None:
            s = self.streams[device.index] = torch.cuda.Stream(device)
        return s

    def join(self):
        self.pending = False
        adopted, self.adopted = self.adopted, []
        for idx, s in self.streams.items():
            torch.cuda.current_stream(idx).wait_stream(s)
        for p, dptr in adopted:
            if p.grad is None or p.grad.data_ptr() != dptr:
                raise RuntimeError("weight gradient computed on the side stream was copied or accumulated by autograd "
                                   "on the main stream before the join (unsynchronised read); call "
                                   "holocron_amd.ops.conv.set_wgrad_side_stream(False) for this training loop")


_SIDE = _WgradSide()
_SIDE.on = False   # process-wide opt-in: set_wgrad_side_stream(True)


def set_wgrad_side_stream(on: bool) -> None:
    _SIDE.on = bool(on)


def wgrad_side_stream_enabled() -> bool:
    return _SIDE.on


class side_stream_for_wgrad:
    """Context manager around the weight-gradient launches of one backward node.  ``params`` are the
    parameters whose gradients are produced inside, ``inputs`` the tensors the kernels read."""

    def __init__(self, params, inputs):
        self.params = params
        self.inputs = [t for t in inputs if t is not None]
        self.ctx = None

    def __enter__(self):
        if not _SIDE.on or PROFILE is not None:
            return self
        if any(p.grad is not None for p in self.params):
            # autograd will accumulate into .grad on this stream.  If that .grad was itself produced on the side
            # stream earlier in this backward (a weight shared by two nodes), order the accumulation after it.
            mine = [i for i, (q, _) in enumerate(_SIDE.adopted) if any(q is p for p in self.params)]
            if mine:
                for idx, s in _SIDE.streams.items():
                    torch.cuda.current_stream(idx).wait_stream(s)
                for i in reversed(mine):
                    del _SIDE.adopted[i]
            return self
        cur = torch.cuda.current_stream()
        side = _SIDE.stream_of(self.inputs[0].device)
        side.wait_stream(cur)
        for t in self.inputs:
            t.record_stream(side)
        if not _SIDE.pending:
            _SIDE.pending = True
            torch.autograd.Variable._execution_engine.queue_callback(_SIDE.join)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def produced(self, *grads):
        """Tell the join which tensors autograd is expected to adopt as ``p.grad``."""
        if self.ctx is not None:
            for p, g in zip(self.params, grads):
                _SIDE.adopted.append((p, g.data_ptr()))

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def conv_wgrad(x, dy, Cin, Cout, KH, KW, stride, pad, out=None, accumulate=False, flops=None, valid=None):
    """dW (fp32 OIHW) of a conv from NHWC-bf16 ``x`` [N,Cin,H,W] and ``dy`` [N,Cout,OH,OW].  ``valid`` = (Cout_v, Cin_v): the
    operands are channel-padded and the result is the unpadded [Cout_v, Cin_v, KH, KW] gradient of the parameter itself."""
    N, _, H, W = x.shape
    _, _, OH, OW = dy.shape
    d = WgradDesc()
    d.N, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = N, H, W, Cin, OH, OW, Cout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.beta = 1 if (accumulate and out is not None) else 0
    lib = _lib.load()
    nbytes = lib.hc_conv_wgrad_ws_bytes(C.byref(d))
    ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=x.device)
    if valid is not None:
        d.co_valid, d.ci_valid = int(valid[0]), int(valid[1])
    if out is None:
        out = torch.empty((Cout, Cin, KH, KW) if valid is None else (int(valid[0]), int(valid[1]), KH, KW), dtype=torch.float32,
                          device=x.device)
    d.x, d.dy, d.dw, d.ws = ptr(x), ptr(dy), ptr(out), ptr(ws)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.hc_conv_wgrad(C.byref(d), stream()), "hc_conv_wgrad")
        e1.record()
        nbytes = sum(t.numel() * t.element_size() for t in (x, dy, out))
        PROFILE.append(("conv_wgrad", 2.0 * N * OH * OW * Cout * KH * KW * Cin if flops is None else flops, e0, e1, nbytes))
        _tag("wgrad N%d %d@%dx%d -> %d@%dx%d k%d s%d" % (N, Cin, H, W, Cout, OH, OW, KH, stride))
        return out
    if not WGRAD_KNOCKOUT:
        check(lib.hc_conv_wgrad(C.byref(d), stream()), "hc_conv_wgrad")
    return out


# HC_WGRAD_KNOCKOUT=1: timing experiment - the conv weight-gradient launches (hc_conv_wgrad, hc_rep_wgrad) are skipped, so a step's time
# is its main-stream critical path (gradients are wrong; VERDICT r4 item 3c: what the weight gradients cost that is NOT hidden)
WGRAD_KNOCKOUT = os.environ.get("HC_WGRAD_KNOCKOUT", "0") == "1"
if WGRAD_KNOCKOUT:
    import warnings
    warnings.warn("HC_WGRAD_KNOCKOUT=1: every conv weight-gradient launch is SKIPPED - weight gradients stay zero and training is wrong. "
                  "This is a timing experiment (scripts/prof_round*.sh), never a training mode.", RuntimeWarning, stacklevel=1)
    print("holocron_amd: HC_WGRAD_KNOCKOUT=1 - conv weight gradients are NOT computed (timing experiment)", file=__import__("sys").stderr)


# ------------------------------------------------------------------ RepBlock weight gradients: fused, grouped, deferred
class _RepWgradQueue:
    """Both weight gradients (3x3 and 1x1) of a RepBlock from one launch (``hc_rep_wgrad``, csrc/conv_wgrad_rep.hip), and the
    launches of same-shaped blocks folded into one.

    A block's weight gradients feed nothing but the optimizer, so a backward node only ENQUEUES them (the tensors it returns to
    autograd are allocated, not yet written); the queue is flushed - one launch per group of up to 16 same-shaped blocks - by an
    autograd final callback at the end of the backward call, or earlier by anything that is about to read a gradient
    (``flush_deferred_wgrads``: GradReducer before it packs a bucket).  Grouping is what keeps the split-K factor small: the 14
    identical 192-channel blocks of repvgg_a0 are 112 (block, channel tile) pairs, so each needs a 2-way pixel split instead of
    the 41-way split a single block needs to fill the chip (whose fp32 partial sums doubled the HBM traffic of the layer).

    Deferral is safe by construction: it is used only when the parameters have no ``.grad`` yet, the tensors handed to autograd
    are ZERO-filled views of one arena (one memset per backward pass), and the flush ADDS the gradients into them.  Autograd
    adopts such a tensor as ``p.grad`` without launching anything; whatever it accumulates into it before the flush (a weight
    shared by two nodes) commutes with the flush's own ``+=``; and if it cloned the tensor instead of adopting it
    (``create_graph``), the flush adds into the clone.  Reading ``p.grad`` from INSIDE the backward pass (a tensor hook) sees the
    gradient only after ``flush_deferred_wgrads()``.  The inputs stay referenced by the queue until the flush (a few GB at
    batch 256: nothing on a 288 GB part)."""

    def __init__(self):
        self.jobs = []
        self.armed = False
        self.task = -1              # autograd graph task the queued jobs (and the pending flush callback) belong to
        self.parked = {}            # task id -> jobs of a pass that another pass interrupted (re-entrant backward) or outlived (dead pass)
        self.cb_tasks = set()       # graph tasks whose final callback is queued
        self.support = {}
        self.enabled = os.environ.get("HC_WREP_DEFER", "1") != "0"
        # HC_WREP_SIDE=1: a group is launched on a second HIP stream as soon as the backward pass moves on to another block shape, so
        # that it runs next to the BatchNorm passes / data gradients of the following blocks; joined at the end of the pass
        self.side_on = os.environ.get("HC_WREP_SIDE", "0") == "1"
        self.side = None
        self.inflight = []          # inputs of launches on the side stream: referenced until the join
        self.side_params = set()
        self.arena = None           # zero-filled fp32 buffer of this backward pass (sized by the previous pass that began alike)
        self.arena_used = 0
        self.arena_want = 0
        self.arena_first = None     # shape key of the first block this pass submitted: passes of a step cut in several
        self.arena_sizes = {}       # (BackwardCut) segments begin with different blocks, and each gets an arena of ITS size

    def _zeros(self, shape, device, key=None):
        n = 1
        for v in shape:
            n *= v
        n64 = (n + 63) // 64 * 64
        if self.arena_want == 0:
            self.arena_first = key
        self.arena_want += n64
        if self.arena is None and self.arena_used == 0 and self.arena_sizes.get(self.arena_first, 0) > 0:
            self.arena = torch.zeros((self.arena_sizes[self.arena_first],), dtype=torch.float32, device=device)
        if self.arena is None or self.arena.device != device or self.arena_used + n64 > self.arena.numel():
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = self.arena[self.arena_used:self.arena_used + n].view(shape)
        self.arena_used += n64
        return out

    @staticmethod
    def _desc(key, njobs=1):
        N, Cin, H, W, Cout, stride = key
        d = RepWgradDesc()
        d.njobs, d.N, d.IH, d.IW, d.Cin, d.Cout, d.stride = njobs, N, H, W, Cin, Cout, stride
        d.OH, d.OW = conv_out_size(H, 3, stride, 1), conv_out_size(W, 3, stride, 1)
        d.accumulate = 0
        return d

    def supported(self, key):
        ok = self.support.get(key)
        if ok is None:
            ok = self.support[key] = bool(_lib.load().hc_rep_wgrad_supported(C.byref(self._desc(key))))
        return ok

    def launch(self, key, jobs, accumulate=False):
        """One launch per group of up to 16 ``jobs`` = (x, dy3, dy1, dw3 pointer, dw1 pointer) of shape ``key``."""
        lib = _lib.load()
        for i in range(0, len(jobs), _lib.HC_WREP_MAX_JOBS):
            grp = jobs[i:i + _lib.HC_WREP_MAX_JOBS]
            d = self._desc(key, len(grp))
            d.accumulate = 1 if accumulate else 0
            for j, (x, dy3, dy1, p3, p1) in enumerate(grp):
                d.x[j], d.dy3[j], d.dy1[j], d.dw3[j], d.dw1[j] = ptr(x), ptr(dy3), ptr(dy1), p3, p1
            nbytes = lib.hc_rep_wgrad_ws_bytes(C.byref(d))
            if nbytes < 0:
                raise _lib.HipError("hc_rep_wgrad: unsupported shape %s" % (key,))
            ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=grp[0][0].device)
            d.ws = ptr(ws)
            if PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.hc_rep_wgrad(C.byref(d), stream()), "hc_rep_wgrad")
                e1.record()
                N, Cin, H, W, Cout, stride = key
                flops = 2.0 * N * d.OH * d.OW * Cout * 10 * Cin * len(grp)
                nb = len(grp) * (2 * N * H * W * Cin + 4 * N * d.OH * d.OW * Cout + 40 * Cout * Cin)
                PROFILE.append(("conv_wgrad", flops, e0, e1, nb))
                _tag("wrep x%d N%d %d@%dx%d -> %d s%d" % (len(grp), N, Cin, H, W, Cout, stride))
            elif not WGRAD_KNOCKOUT:
                check(lib.hc_rep_wgrad(C.byref(d), stream()), "hc_rep_wgrad")

    def submit(self, key, x, dy3, dy1, w3, w1):
        """Queue one block; returns the (zero-filled, to be accumulated into) gradient tensors for autograd."""
        Cout, Cin = key[4], key[1]
        task = torch._C._current_graph_task_id()
        if self.armed and task != self.task:
            # Another graph task submits while the queue is armed.  Either a RE-ENTRANT backward nested inside a pass that is still
            # running (torch.utils.checkpoint(use_reentrant=True), a backward() inside a custom Function: the task ids run [0, 1, 0]),
            # or the armed pass DIED before its final callback (backward raised) and this is a retry on the retained graph with no
            # forward in between.  The two cannot be told apart here, so the armed pass's jobs are PARKED under its task id - neither
            # launched (a dead pass would be counted twice under gradient accumulation: ADVICE r4 / r5) nor dropped (a live outer
            # pass would leave zero placeholders in .grad: ADVICE r3).  A live pass takes them back on its next submit or launches them
            # from its final callback; a dead pass never comes back and its parked jobs go with the next forward (note_forward).
            self._park()
        dw3 = self._zeros((Cout, Cin, 3, 3), x.device, key)
        dw1 = self._zeros((Cout, Cin, 1, 1), x.device, key)
        # The queue must not hold the gradient TENSORS: AccumulateGrad only adopts a gradient it holds the sole reference to
        # (otherwise it clones it on the spot).  The storages keep the memory alive instead.
        if self.side_on and PROFILE is None:
            if id(w3) in self.side_params or id(w1) in self.side_params:
                self.join()             # a weight shared by two nodes: autograd will accumulate into a gradient the side stream writes
            elif self.jobs and self.jobs[-1][0] != key:
                self._launch_on_side()  # the pass moved on to another shape: the finished group goes to the side stream now
        self.jobs.append((key, x, dy3, dy1, dw3.untyped_storage(), dw3.data_ptr(), dw3.storage_offset(),
                          dw1.untyped_storage(), dw1.data_ptr(), dw1.storage_offset(), w3, w1))
        if not self.armed:
            self.armed, self.task = True, task
            back = self.parked.pop(task, None)
            if back:                                 # the outer pass of a re-entrant backward goes on: its parked jobs rejoin the queue
                self.jobs = back + self.jobs
            if task not in self.cb_tasks:            # one final callback per graph task
                self.cb_tasks.add(task)
                torch.autograd.Variable._execution_engine.queue_callback(functools.partial(self._final, task))
        return dw3, dw1

    def _park(self):
        self.join()
        if self.jobs:
            self.parked.setdefault(self.task, []).extend(self.jobs)
        self.jobs = []
        self.armed, self.task = False, -1
        self.arena, self.arena_used, self.arena_want, self.arena_first = None, 0, 0, None   # (a partial pass records no arena size)

    def _final(self, task):
        """Final callback of graph task ``task`` (runs only when that pass completed)."""
        self.cb_tasks.discard(task)
        if self.armed and self.task == task:
            self.flush()
            return
        jobs = self.parked.pop(task, None)           # the pass queued jobs, was interrupted by a nested pass and submitted nothing after it
        if jobs:
            self._launch_groups(jobs)
            self._fix_clones([jobs])

    def note_forward(self):
        """Called from RepBlockFn.forward.  A forward that runs while the queue is armed and NO backward pass is executing
        (`_current_graph_task_id() == -1`; the recomputation of a re-entrant checkpoint runs inside one) means the pass that armed the
        queue died before its final callback - backward raised and the caller went on.  Its jobs are DROPPED, not launched: the
        placeholders autograd adopted stay zero, so a retry of the micro-batch without zero_grad accumulates the right gradient into
        them (launching the stale jobs later, from inside the retry, would count the failed micro-batch twice: ADVICE r4)."""
        if (self.armed or self.parked) and torch._C._current_graph_task_id() == -1:
            self.jobs = []
            self.parked, self.cb_tasks = {}, set()
            if self.inflight:
                torch.cuda.current_stream().wait_stream(self.side)
                self.inflight, self.side_params = [], set()
            self.armed, self.task = False, -1
            self.arena, self.arena_used, self.arena_want, self.arena_first = None, 0, 0, None

    def _launch_on_side(self):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        cur = torch.cuda.current_stream()
        if self.side is None or self.side.device != cur.device:
            self.side = torch.cuda.Stream(cur.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self._launch_groups(jobs)
        self.inflight.append(jobs)
        for j in jobs:
            self.side_params.add(id(j[10]))
            self.side_params.add(id(j[11]))

    def join(self):
        """Everything launched on the side stream is ordered before what the caller's stream does next."""
        if self.inflight:
            torch.cuda.current_stream().wait_stream(self.side)
            self._fix_clones(self.inflight)
            self.inflight = []
            self.side_params = set()

    def _launch_groups(self, jobs):
        groups = {}
        for job in jobs:
            groups.setdefault(job[0], []).append(job)
        for key, grp in groups.items():
            self.launch(key, [(j[1], j[2], j[3], j[5], j[8]) for j in grp], accumulate=True)

    @staticmethod
    def _fix_clones(job_lists):
        # a gradient that autograd cloned instead of adopting (create_graph, a hook that kept a reference): the clone was
        # taken before the launch - add what the launch produced
        for jobs in job_lists:
            for (_, x, _, _, s3, p3, o3, s1, p1, o1, w3, w1) in jobs:
                for w, st, p, off in ((w3, s3, p3, o3), (w1, s1, p1, o1)):
                    if w is None:
                        continue
                    g = w.grad
                    if g is not None and g.data_ptr() != p:
                        g.add_(torch.empty(0, dtype=torch.float32, device=x.device).set_(st, off, g.shape))

    def flush(self):
        self.armed, self.task = False, -1
        jobs, self.jobs = self.jobs, []
        self.arena, self.arena_used = None, 0          # the views handed out keep the buffer alive
        if self.arena_want:
            self.arena_sizes[self.arena_first] = self.arena_want
        self.arena_want, self.arena_first = 0, None
        if jobs and self.inflight:          # side-stream mode: the last groups go there too, then everything is joined
            self.jobs = jobs
            self._launch_on_side()
            jobs = []
        if jobs:
            self._launch_groups(jobs)
            self._fix_clones([jobs])
        self.join()


_WREP = _RepWgradQueue()


class _ConvWgradQueue(_RepWgradQueue):
    """The same deferral for the PLAIN conv units (conv_sequence: conv -> BatchNorm -> activation, nn/convbn_op.py): the weight
    gradients of a backward pass are queued and launched at its end, same-shaped layers in ONE ``hc_conv_wgrad_group`` launch pair.
    The DarkNet / CSP / YOLO stacks repeat a handful of conv shapes 4-9 times per step (the ResBlocks of a CSP stage); one such layer
    has 3-6 weight-gradient tiles and on its own needs 40-85 split-K slabs to fill the chip - its fp32 partial sums are several times
    its operands.  Grouped, tiles x jobs fill the chip and the split factor drops by the group size.  Arming, parking, the zero
    placeholders and the end-of-pass flush are the RepBlock queue's (one weight per job instead of two); layers whose shape the
    grouped kernel does not take are launched one by one at the flush."""

    def __init__(self):
        super().__init__()
        self.enabled = os.environ.get("HC_WGRAD_DEFER", "1") != "0"
        # (a complete group launched on a second stream beside the rest of the pass - the RepBlock queue's HC_WREP_SIDE idea with a
        # "group is complete" trigger - measured SLOWER on YOLOv4: 26.67 / 26.69 -> 27.53 / 27.56 ms, same box; the weight-gradient
        # kernels and the main stream's passes compete for the same fill paths.  Removed; the groups run behind the pass.)
        self.side_on = False

    @staticmethod
    def _gdesc(key, njobs=1):
        N, Cin, H, W, Cout, KH, KW, stride, pad = key
        d = WgradGroupDesc()
        d.njobs, d.N, d.IH, d.IW, d.Cin, d.Cout, d.stride, d.pad = njobs, N, H, W, Cin, Cout, stride, pad
        d.KH, d.KW = KH, KW
        d.OH, d.OW = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
        return d

    def supported(self, key):
        ok = self.support.get(key)
        if ok is None:
            d = self._gdesc(key)
            d.x[0] = d.dy[0] = d.dw[0] = 1        # (the support query looks at the geometry only)
            ok = self.support[key] = bool(_lib.load().hc_conv_wgrad_group_supported(C.byref(d)))
        return ok

    def launch(self, key, jobs, accumulate=False):
        """``jobs`` = (x, dy, _, dw pointer, _) of shape ``key``: groups of up to 16 through the grouped kernel."""
        lib = _lib.load()
        N, Cin, H, W, Cout, KH, KW, stride, pad = key
        for i in range(0, len(jobs), _lib.HC_WGRAD_MAX_JOBS):
            grp = jobs[i:i + _lib.HC_WGRAD_MAX_JOBS]
            d = self._gdesc(key, len(grp))
            d.beta = 1 if accumulate else 0
            for j, (x, dy, _, p, _) in enumerate(grp):
                d.x[j], d.dy[j], d.dw[j] = ptr(x), ptr(dy), p
            nbytes = lib.hc_conv_wgrad_group_ws_bytes(C.byref(d))
            if nbytes < 0:
                raise _lib.HipError("hc_conv_wgrad_group: unsupported shape %s" % (key,))
            ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=grp[0][0].device)
            d.ws = ptr(ws)
            if PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.hc_conv_wgrad_group(C.byref(d), stream()), "hc_conv_wgrad_group")
                e1.record()
                flops = 2.0 * N * d.OH * d.OW * Cout * KH * KW * Cin * len(grp)
                nb = len(grp) * (2 * N * H * W * Cin + 2 * N * d.OH * d.OW * Cout + 4 * Cout * Cin * KH * KW)
                PROFILE.append(("conv_wgrad", flops, e0, e1, nb))
                _tag("wgrad x%d N%d %d@%dx%d -> %d@%dx%d k%d s%d" % (len(grp), N, Cin, H, W, Cout, d.OH, d.OW, KH, stride))
            elif not WGRAD_KNOCKOUT:
                check(lib.hc_conv_wgrad_group(C.byref(d), stream()), "hc_conv_wgrad_group")

    def submit1(self, key, x, dy, w):
        """Queue one layer; returns the zero-filled (to be accumulated into) gradient tensor for autograd."""
        Cout, Cin, KH, KW = key[4], key[1], key[5], key[6]
        task = torch._C._current_graph_task_id()
        if self.armed and task != self.task:
            self._park()
        dw = self._zeros((Cout, Cin, KH, KW), x.device, key)
        self.jobs.append((key, x, dy, None, dw.untyped_storage(), dw.data_ptr(), dw.storage_offset(), None, 0, 0, w, None))
        if not self.armed:
            self.armed, self.task = True, task
            back = self.parked.pop(task, None)
            if back:
                self.jobs = back + self.jobs
            if task not in self.cb_tasks:
                self.cb_tasks.add(task)
                torch.autograd.Variable._execution_engine.queue_callback(functools.partial(self._final, task))
        return dw


_WCONV = _ConvWgradQueue()


def conv_wgrad_unit(x, dy, w, Cin, Cout, KH, KW, stride, pad):
    """dW of a plain conv unit from inside its backward node: queued for the grouped end-of-pass launch when nothing can read the
    gradient before the pass ends (``_may_defer``) and the grouped kernel takes the shape, else computed now (``conv_wgrad``)."""
    N, _, H, W = x.shape
    if (_WCONV.enabled and not _SIDE.on and torch._C._current_graph_task_id() != -1 and tuple(w.shape) == (Cout, Cin, KH, KW)
            and w.dtype == torch.float32):
        key = (N, Cin, H, W, Cout, KH, KW, stride, pad)
        if _WCONV.supported(key) and _may_defer(w):
            return _WCONV.submit1(key, x, dy, w)
    return conv_wgrad(x, dy, Cin, Cout, KH, KW, stride, pad)


def flush_deferred_wgrads() -> None:
    """Launch every weight gradient that is still queued (call before reading ``.grad`` from inside a backward pass)."""
    for q in (_WREP, _WCONV):
        if q.jobs or q.inflight:
            arena = (q.arena, q.arena_used, q.arena_want, q.arena_first)
            sizes = dict(q.arena_sizes)
            task, cbs = q.task, set(q.cb_tasks)
            q.flush()
            # a mid-pass flush must not drop the arena of the pass that is still running (nor record its partial size); the pass's final
            # callback is still queued: the next submit of the pass re-arms without queueing a second one
            q.arena, q.arena_used, q.arena_want, q.arena_first = arena
            q.arena_sizes = sizes
            q.cb_tasks = cbs


_FLUSH_AWARE = {}    # id(parameter) -> ids of its post-accumulate hooks that call flush_deferred_wgrads() before reading .grad
_MANAGED = set()     # id(parameter) of everything a parallel.GradReducer averages (it flushes before it packs a bucket)


def mark_reducer_managed(params, on: bool = True) -> None:
    """parallel.GradReducer: these parameters' gradients are read by a reducer that calls ``flush_deferred_wgrads()`` first."""
    for p in params:
        (_MANAGED.add if on else _MANAGED.discard)(id(p))


def register_flush_aware_hook(p, handle) -> None:
    """Tell the deferred weight-gradient queue that the post-accumulate-grad hook behind ``handle`` (the object
    ``p.register_post_accumulate_grad_hook`` returned) calls ``flush_deferred_wgrads()`` before it reads any gradient
    (parallel.GradReducer does).  Every other hook makes the fused RepBlock launch immediately instead of deferring."""
    _FLUSH_AWARE.setdefault(id(p), set()).add(handle.id)


def set_deferred_wgrads(on: bool) -> None:
    """Process-wide switch of the deferred (grouped) RepBlock weight gradients (public as ``holocron_amd.set_deferred_wgrads``).
    Switch it off under a wrapper that reads gradients DURING backward through hooks this package cannot see (torch's
    DistributedDataParallel / FSDP register theirs in C++): with deferral on such a reader sees the zero-filled placeholder."""
    if not on:
        flush_deferred_wgrads()
    _WREP.enabled = bool(on)
    _WCONV.enabled = bool(on)


def _may_defer(*params) -> bool:
    """Deferral hands autograd a zero-filled gradient that the end-of-pass flush fills.  Anything that reads a gradient inside the
    pass would see zeros, so it is used only when nothing can: no tensor hooks, no post-accumulate hooks other than registered
    flush-aware ones, and - with more than one rank - only under this package's GradReducer (DistributedDataParallel / FSDP hang
    their bucket hooks on the gradient accumulators in C++, invisible from here, and would all-reduce the zeros: ADVICE r2)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    for p in params:
        if p.grad is not None:
            return False
        if getattr(p, "_backward_hooks", None):
            return False
        post = getattr(p, "_post_accumulate_grad_hooks", None)
        aware = _FLUSH_AWARE.get(id(p), ())
        if post and any(k not in aware for k in post):
            return False
        if multi and id(p) not in _MANAGED:
            return False
    return True


def rep_block_wgrad(x, dy3, dy1, w3, w1, stride, defer=False):
    """(dW3, dW1) of a RepBlock through the fused kernel, or None when the shape is outside its plan.  ``defer``: inside an
    autograd backward pass, enqueue and return tensors that the flush at the end of the pass fills."""
    N, Cin, H, W = x.shape
    Cout = w3.shape[0]
    key = (N, Cin, H, W, Cout, stride)
    if not _WREP.supported(key):
        return None
    if defer and _WREP.enabled and _may_defer(w3, w1):
        return _WREP.submit(key, x, dy3, dy1, w3, w1)
    dw3 = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    dw1 = torch.empty((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
    _WREP.launch(key, [(x, dy3, dy1, dw3.data_ptr(), dw1.data_ptr())])
    return dw3, dw1


def im2col_small(x, KH, KW, stride, pad, Kpad):
    """NCHW fp32 (tiny Cin) -> NHWC bf16 column tensor, logical shape [N, Kpad, OH, OW]."""
    N, Cin, H, W = x.shape
    OH, OW = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
    xc = x.detach()
    if xc.dtype != torch.float32 or not xc.is_contiguous():
        xc = xc.float().contiguous()
    col = empty_cl(N, Kpad, OH, OW, x.device)
    with profiled("stem_im2col", 0.0, xc.numel() * 4.0):     # algorithmically the stem reads its fp32 input once: the column tensor is overhead
        check(_lib.load().hc_im2col_small(ptr(xc), ptr(col), N, Cin, H, W, OH, OW, KH, KW, stride, pad, Kpad, stream()),
              "hc_im2col_small")
    return col


# ------------------------------------------------------------------ plain conv2d (inference / tests)
def conv2d(x, weight, bias=None, stride=1, padding=0, act=0, stats=None):
    """Forward convolution on the MFMA kernel (no autograd).  x: logical NCHW; returns bf16 NHWC."""
    Cout, Cin, KH, KW = weight.shape
    N, _, H, W = x.shape
    if Cin % 16 != 0:
        K = Cin * KH * KW
        Kpad = ((K + 15) // 16) * 16
        col = im2col_small(x, KH, KW, stride, padding, Kpad)
        wpk = pack_weight_im2col(weight, Kpad)
        d = fwd_desc(N, Kpad, col.shape[2], col.shape[3], Cout, 1, 1, 1, 0)
        src = col
    else:
        src = to_cl_bf16(x)
        wpk = pack_weight(weight, 0)
        d = fwd_desc(N, Cin, H, W, Cout, KH, KW, stride, padding)
    out = empty_cl(N, Cout, d.OH, d.OW, x.device)
    b = None if bias is None else bias.detach().float().contiguous()
    launch_conv(d, src, wpk, out, stats=stats, bias=b, act=act)
    return out


# ------------------------------------------------------------------ small-channel stride-1 3x3 (+1x1)
def conv_small_desc(N, H, W, Cc, Cout, mode):
    """Descriptor of hc_conv_small, or None when the shape is outside what that kernel supports."""
    d = ConvSmallDesc()
    d.N, d.H, d.W, d.C, d.Cout, d.mode = N, H, W, Cc, Cout, mode
    if not _lib.load().hc_conv_small_supported(C.byref(d)):
        return None
    return d


def launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, stats3=None, stats1=None):
    """y3 = conv3x3(x), y1 = conv1x1(x) (stride 1) with wp3 [Cout][9][C], wp1 [Cout][1][C]."""
    d.srcA, d.srcB, d.w3, d.w1 = ptr(x), None, ptr(wp3), ptr(wp1)
    d.w3_rstride, d.w1_rstride = 9 * d.C, d.C
    d.out3, d.out1, d.resid, d.stats3, d.stats1 = ptr(y3), ptr(y1), None, ptr(stats3), ptr(stats1)
    _launch_small(d, 2.0 * d.N * d.H * d.W * d.Cout * 10 * d.C, 2.0 * d.N * d.H * d.W * (d.C + 2 * d.Cout))


def launch_conv_small_dgrad(d, dy3, dy1, wpd, dx, resid=None):
    """dx = conv3x3^T(dy3) + conv1x1^T(dy1) + resid with wpd [Cin][10][Cout] (mode-1 packing);
    here d.C = Cout of the forward conv and d.Cout = its Cin."""
    d.srcA, d.srcB, d.w3 = ptr(dy3), ptr(dy1), ptr(wpd)
    d.w1 = ptr(wpd) + 9 * d.C * 2
    d.w3_rstride, d.w1_rstride = 10 * d.C, 10 * d.C
    d.out3, d.out1, d.resid, d.stats3, d.stats1 = ptr(dx), None, ptr(resid), None, None
    _launch_small(d, 2.0 * d.N * d.H * d.W * d.Cout * 10 * d.C,
                  2.0 * d.N * d.H * d.W * (2 * d.C + d.Cout * (2 if resid is not None else 1)))


def _launch_small(d, flops, nbytes=0):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().hc_conv_small(C.byref(d), stream()), "hc_conv_small")
        e1.record()
        PROFILE.append(("conv_rows" if (d.mode & ROWS_IMAGE) else "conv_small", flops, e0, e1, nbytes))
        _tag("%s N%d %d@%dx%d -> %d mode%d" % ("conv_rows" if (d.mode & ROWS_IMAGE) else "conv_small", d.N, d.C, d.H, d.W, d.Cout, d.mode))
        return
    check(_lib.load().hc_conv_small(C.byref(d), stream()), "hc_conv_small")
