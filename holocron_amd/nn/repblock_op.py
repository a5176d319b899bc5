"""Fused training/inference step of a RepVGG block on the HIP kernels.

Reference semantics (holocron/models/classification/repvgg.py:71-73):
    out = act( BN3(conv3x3(x)) + BN1(conv1x1(x)) [+ BN0(x)] )
with torch ``nn.BatchNorm2d`` in training mode (biased batch variance for normalisation,
unbiased for ``running_var``, momentum 0.1, ``num_batches_tracked += 1``).

Kernel sequence (DESIGN.md "RepBlock step"):
  forward : conv3x3 -> y3 (+ sum/sumsq epilogue) ; conv1x1 -> y1 (+ stats) ; bn_finalize ;
            rep_apply (3 x BN-apply + add + ReLU in one pass, optional stats of `out`)
  backward: rep_bwd_reduce (sum dz, dz*y3, dz*y1, dz*x) ; bn_bwd_finalize ; rep_bwd_apply
            (dy3, dy1, dx_identity) ; dual-source dgrad (3x3 + 1x1 in one accumulator, + dx_id) ;
            wgrad 3x3 ; wgrad 1x1
"""
import ctypes as C

import os

import torch

from .. import _lib
from .._lib import ConvS2Desc, ConvS2DgradDesc, PackItem, RepBnBwdDesc, RepBnDesc, StemBwdDesc, StemDesc, check, ptr, stream
from ..ops import conv as cv

STEM_KPAD = 32


class ZeroPool:
    """One zero-filled fp32 arena per training step instead of ~100 tiny ``torch.zeros`` launches:
    ``begin()`` zeroes the extent used so far with a single memset, ``take(n)`` hands out views.
    The buffer is allocated during warm-up, so a captured hipGraph keeps using the same addresses; a buffer that was handed out
    under stream capture and then had to be replaced is retired, not freed, for the same reason (``release_retired()`` drops those
    once the graphs are gone: parallel.GraphedStep.release does) - one that no capture ever saw is simply freed.

    Views are only valid until the next ``begin()``: ``gen`` counts the arena's generations, and anything that is taken
    in one autograd node's forward for use in its backward (``take_for_backward``) is re-validated there with ``claim`` -
    a second training-mode forward before the first backward (siamese / multi-crop / GAN steps, teacher + student) has
    re-zeroed and re-issued the same offsets, and the stale node then gets a private zero buffer instead."""

    def __init__(self):
        self.buf = None
        self.used = 0
        self.high = 0
        self.depth = 0
        self.gen = 0
        self.retired = []
        self.graph_users = 0        # live GraphedSteps whose graphs may reference a retired buffer
        self.buf_captured = False   # the current buffer was handed out while a stream capture was running

    def _retire(self):
        if self.buf is not None and self.buf_captured:
            self.retired.append(self.buf)
        self.buf, self.buf_captured = None, False

    def release_retired(self):
        """One holder of captured graphs (parallel.GraphedStep: ``graph_users`` counts them) has destroyed its graphs; the buffers
        kept alive for captured graphs are freed when the last holder has."""
        self.graph_users = max(0, self.graph_users - 1)
        if self.graph_users == 0:
            self.retired = []

    @property
    def active(self):
        return self.depth > 0

    def begin(self, device):
        if self.depth > 0:          # a model called inside another model's forward shares the outer arena
            self.depth += 1
            return
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.high:
            self._retire()
            self.buf = torch.zeros(max(self.high * 2, 1 << 20), dtype=torch.float32, device=device)
        else:
            self.buf[: max(self.high, 1)].zero_()
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self.buf_captured = True
        self.used = 0
        self.gen += 1
        self.depth = 1

    def end(self):
        self.depth = max(0, self.depth - 1)

    def reset(self):
        """Forget the high-water mark (the statistics replica count changed: deterministic mode sizes everything 256x larger and
        `begin()` zeroes up to the mark on every step).  The old buffer is freed unless a captured graph may still use it."""
        self._retire()
        self.used, self.high = 0, 0
        self.gen += 1

    def take(self, shape, device):
        n = 1
        for v in shape:
            n *= v
        n4 = (n + 3) // 4 * 4
        if not self.active or self.buf is None or self.used + n4 > self.buf.numel() or self.buf.device != device:
            self.high = max(self.high, self.used + n4)
            if self.active:
                self.used += n4
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = self.buf[self.used:self.used + n].view(shape)
        self.used += n4
        self.high = max(self.high, self.used)
        return out

    def take_for_backward(self, shape, device):
        """(zeroed view, generation) - pass both to ``claim`` in backward."""
        return self.take(shape, device), self.gen

    def claim(self, view, gen, shape, device):
        """The view handed out in forward if the arena has not been recycled since, else a fresh zero buffer."""
        if view is not None and gen == self.gen:
            return view
        return torch.zeros(shape, dtype=torch.float32, device=device)


POOL = ZeroPool()
_lib.on_replicas_changed(POOL.reset)


def fill_pack_items(arr, items):
    """items: (w, dst, Cout, Cin, KH, KW, mode, tap0, T[, ld]) tuples -> hc_pack_item array; returns the largest source size."""
    mx = 0
    for a, it in zip(arr, items):
        w, dst, Cout, Cin, KH, KW, mode, tap0, T = it[:9]
        a.w, a.dst, a.Cout, a.Cin, a.KH, a.KW, a.mode, a.tap0, a.T = w.data_ptr(), dst.data_ptr(), Cout, Cin, KH, KW, mode, tap0, T
        a.ld = it[9] if len(it) > 9 else 0
        mx = max(mx, w.numel())
    return mx


def launch_pack_items(items, cache=None):
    """One hc_pack_conv_weights_multi launch for a (small) item list: the path of a block whose images went stale on their own
    (first call, geometry change, a RepBlock used outside a model that packs all its blocks in one launch per step).

    ``cache``: a dict owned by the caller.  The device-side item table is uploaded once per (source, destination) pointer signature and
    kept there: a repack of the same buffers - every optimizer step for a free-standing block - is then a plain kernel launch, safe
    under stream capture / GraphedStep.  A pageable host-to-device copy per repack would either fail inside a capture or be recorded
    as a memcpy node whose host source is gone by the time the graph replays."""
    if not items:
        return
    sig = tuple((it[0].data_ptr(), it[1].data_ptr(), it[6], it[7]) for it in items)
    ent = cache.get("table") if cache is not None else None
    if ent is None or ent[0] != sig:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("RepBlock weight images must be packed once outside stream capture (run one eager step first): the item "
                               "table upload is a host-to-device copy")
        import numpy as np
        arr = (PackItem * len(items))()
        mx = fill_pack_items(arr, items)
        table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(items[0][0].device)
        ent = (sig, table, len(items), mx)
        if cache is not None:
            cache["table"] = ent
    check(_lib.load().hc_pack_conv_weights_multi(ent[1].data_ptr(), ent[2], ent[3], stream()), "hc_pack_conv_weights_multi")


class RepState:
    """Per-module host state: packed-weight caches, geometry descriptors, BN buffers."""

    def __init__(self, stride, identity):
        self.stride = stride
        self.identity = identity
        self.emit_stats = False     # set by the model builder when the consumer has an identity BN
        self.fwd_cache = cv.PackCache()
        self.bwd_cache = cv.PackCache()
        self.desc = {}
        self.bn = None              # list of (running_mean, running_var, num_batches_tracked)
        self.eps = 1e-5
        self.momentum = 0.1
        self.training = True
        self.last_out_stats = None
        self.packed = None          # persistent packed-weight buffers (wp3, wp1, wpd)
        self.packed_key = None
        self.pack_table = {}        # launch_pack_items' device-side item table (keyed by the buffer pointers)
        self.rows_image = (False, False)   # set by descs(): (forward, data gradient) run on a row-unit kernel, which reads its own weight image
        self.stack_fwd = False      # set by pack_items(): 3x3 + 1x1 forward as ONE gather-conv over stacked weight rows
        self.s2 = False             # set by descs(): the forward runs on the stride-2 row kernel (csrc/conv_s2.hip), which reads its own
        self.s2_images = None       # fragment images (hc_pack_conv_weights_multi modes 5 / 6): (3x3, 1x1)
        self.s2_dgrad = False       # set by descs(): the data gradient runs on hc_conv_s2_dgrad (image of pack mode 7)
        self.s2_dimg = None

    # ---- packed weights (persistent buffers; refreshed by one multi-tensor launch per model) ----
    @staticmethod
    def weights_key(w3, w1):
        return (w3.data_ptr(), w3._version, w1.data_ptr(), w1._version, cv.weights_epoch())

    def pack_items(self, w3, w1):
        """[(w, dst, Cout, Cin, KH, KW, mode, tap0, T)] — allocates the destination buffers once."""
        Cout, Cin = w3.shape[0], w3.shape[1]
        dev = w3.device
        stem = (Cin % 16) != 0
        if self.s2:
            return self._pack_items_s2(w3, w1, Cout, Cin, dev, stem)
        # small-channel stride-2 blocks and the stem read their input twice (3x3, then 1x1) in HBM-bound launches: stack the two
        # kernels as 2 * Cout weight rows (the 1x1 at the centre tap resp. at its im2col columns) and gather the input once
        self.stack_fwd = (stem or (self.stride == 2 and Cin <= 48)) and Cout % 4 == 0
        if self.packed is None or self.packed[0].device != dev:
            if self.stack_fwd:
                self.packed = (torch.zeros((2 * Cout, 1, STEM_KPAD) if stem else (2 * Cout, 9, Cin), dtype=torch.bfloat16, device=dev),
                               None, None if stem else torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device=dev))
            elif any(self.rows_image):     # a row-unit image is shared by the 3x3 and the 1x1 kernel
                rf, rb = self.rows_image
                self.packed = (cv.rows_image(Cout, dev) if rf else torch.empty((Cout, 9, Cin), dtype=torch.bfloat16, device=dev),
                               None if rf else torch.empty((Cout, 1, Cin), dtype=torch.bfloat16, device=dev),
                               cv.rows_image(Cout, dev) if rb else torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device=dev))
            elif stem:
                self.packed = (torch.zeros((Cout, 1, STEM_KPAD), dtype=torch.bfloat16, device=dev),
                               torch.zeros((Cout, 1, STEM_KPAD), dtype=torch.bfloat16, device=dev), None)
            else:
                self.packed = (torch.empty((Cout, 9, Cin), dtype=torch.bfloat16, device=dev),
                               torch.empty((Cout, 1, Cin), dtype=torch.bfloat16, device=dev),
                               torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device=dev))
            self.packed_key = None
        wp3, wp1, wpd = self.packed
        if self.stack_fwd:
            if stem:
                return [(w3, wp3, Cout, Cin, 3, 3, 2, 0, STEM_KPAD), (w1, wp3[Cout:], Cout, Cin, 1, 1, 2, 4 * Cin, STEM_KPAD)]
            return [(w3, wp3, Cout, Cin, 3, 3, 0, 0, 9), (w1, wp3[Cout:], Cout, Cin, 1, 1, 0, 4, 9),
                    (w3, wpd, Cout, Cin, 3, 3, 1, 0, 10), (w1, wpd, Cout, Cin, 1, 1, 1, 9, 10)]
        if any(self.rows_image):
            rf, rb = self.rows_image
            fwd = ([(w3, wp3, Cout, Cin, 3, 3, 3, 0, 10), (w1, wp3, Cout, Cin, 1, 1, 3, 9, 10)] if rf else
                   [(w3, wp3, Cout, Cin, 3, 3, 0, 0, 9), (w1, wp1, Cout, Cin, 1, 1, 0, 0, 1)])
            bwd = ([(w3, wpd, Cout, Cin, 3, 3, 4, 0, 10), (w1, wpd, Cout, Cin, 1, 1, 4, 9, 10)] if rb else
                   [(w3, wpd, Cout, Cin, 3, 3, 1, 0, 10), (w1, wpd, Cout, Cin, 1, 1, 1, 9, 10)])
            return fwd + bwd
        if stem:
            return [(w3, wp3, Cout, Cin, 3, 3, 2, 0, STEM_KPAD), (w1, wp1, Cout, Cin, 1, 1, 2, 4 * Cin, STEM_KPAD)]
        return [(w3, wp3, Cout, Cin, 3, 3, 0, 0, 9), (w1, wp1, Cout, Cin, 1, 1, 0, 0, 1),
                (w3, wpd, Cout, Cin, 3, 3, 1, 0, 10), (w1, wpd, Cout, Cin, 1, 1, 1, 9, 10)]

    def _pack_items_s2(self, w3, w1, Cout, Cin, dev, stem):
        """Stride-2 row kernel: fragment images [Cout / 16][steps][64][8] of both kernels (zero where the K stream carries the other
        kernel's pieces) + the usual data-gradient image (the data gradient still runs on the gather-conv)."""
        if stem:
            s3, s1, s1b, mode, tap1 = 2, 1, 0, 6, 4
        else:
            pt = Cin // 8
            s3, s1b, mode, tap1 = (9 * pt + 3) // 4, (9 * pt) // 4, 5, 9 * pt
            s1 = (10 * pt + 3) // 4 - s1b
        if self.s2_images is None or self.s2_images[0].device != dev:
            self.s2_images = (torch.zeros((Cout // 16, s3, 64, 8), dtype=torch.bfloat16, device=dev),
                              torch.zeros((Cout // 16, s1, 64, 8), dtype=torch.bfloat16, device=dev))
            self.packed_key = None
        gather_dgrad = not stem and not self.s2_dgrad
        if self.packed is None or (self.packed[2] is not None) != gather_dgrad or (gather_dgrad and self.packed[2].device != dev):
            self.packed = (None, None, torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device=dev) if gather_dgrad else None)
            self.packed_key = None
        i3, i1 = self.s2_images
        items = [(w3, i3, Cout, Cin, 3, 3, mode, 0, s3, 0), (w1, i1, Cout, Cin, 1, 1, mode, tap1, s1, s1b)]
        if gather_dgrad:
            wpd = self.packed[2]
            items += [(w3, wpd, Cout, Cin, 3, 3, 1, 0, 10), (w1, wpd, Cout, Cin, 1, 1, 1, 9, 10)]
        elif self.s2_dgrad:
            sd = 10 * (Cout // 8) // 4
            if self.s2_dimg is None or self.s2_dimg.device != dev:
                self.s2_dimg = torch.zeros((Cin // 16, sd, 64, 8), dtype=torch.bfloat16, device=dev)
                self.packed_key = None
            items += [(w3, self.s2_dimg, Cout, Cin, 3, 3, 7, 0, sd), (w1, self.s2_dimg, Cout, Cin, 1, 1, 7, 1, sd)]
        return items

    def ensure_packed(self, w3, w1):
        key = self.weights_key(w3, w1)
        if self.packed_key != key or self.packed is None:
            launch_pack_items(self.pack_items(w3, w1), self.pack_table)
            self.packed_key = key
        return self.packed

    def stacked_desc(self, N, Cin, H, W, Cout):
        """forward descriptor of the stacked 3x3 + 1x1 launch (2 * Cout output channels)"""
        key = ("stack", N, Cin, H, W, Cout)
        if key not in self.desc:
            s = self.stride
            if Cin % 16 == 0:
                self.desc[key] = cv.fwd_desc(N, Cin, H, W, 2 * Cout, 3, 3, s, 1)
            else:
                OH, OW = cv.conv_out_size(H, 3, s, 1), cv.conv_out_size(W, 3, s, 1)
                self.desc[key] = cv.fwd_desc(N, STEM_KPAD, OH, OW, 2 * Cout, 1, 1, 1, 0)
        return self.desc[key]

    def descs(self, N, Cin, H, W, Cout):
        key = (N, Cin, H, W, Cout)
        if key not in self.desc:
            s = self.stride
            if Cin % 16 == 0:
                f3 = cv.fwd_desc(N, Cin, H, W, Cout, 3, 3, s, 1)
                f1 = cv.fwd_desc(N, Cin, H, W, Cout, 1, 1, s, 0)
                dg = cv.dgrad_desc(N, Cin, H, W, Cout, [(3, 3, 1, 0, 0), (1, 1, 0, 1, 9)], s)
            else:  # stem: explicit im2col, both branches are 1x1 convs over the column tensor
                OH, OW = cv.conv_out_size(H, 3, s, 1), cv.conv_out_size(W, 3, s, 1)
                f3 = cv.fwd_desc(N, STEM_KPAD, OH, OW, Cout, 1, 1, 1, 0)
                f1 = cv.fwd_desc(N, STEM_KPAD, OH, OW, Cout, 1, 1, 1, 0)
                dg = None
            sf = sd = None
            rows = (False, False)
            if Cin % 16 == 0 and s == 1:   # fused 3x3 + 1x1 kernels: row-unit (own weight image), small-channel persistent, image-resident
                sf = cv.conv_small_desc(N, H, W, Cin, Cout, cv.ROWS_IMAGE)
                sd = cv.conv_small_desc(N, H, W, Cout, Cin, cv.ROWS_IMAGE | 1)
                rows = (sf is not None, sd is not None)
                if sf is None:
                    sf = cv.conv_small_desc(N, H, W, Cin, Cout, 0)
                if sd is None:
                    sd = cv.conv_small_desc(N, H, W, Cout, Cin, 1)
            s2d = None
            if s == 2:
                c = ConvS2Desc()
                c.N, c.H, c.W, c.Cin, c.Cout, c.x_nchw_f32 = N, H, W, Cin, Cout, 1 if Cin % 16 else 0
                if _lib.load().hc_conv_s2_supported(C.byref(c)):
                    s2d = c
            s2g = None
            if s == 2 and s2d is not None and Cin % 16 == 0:
                c = ConvS2DgradDesc()
                c.N, c.H, c.W, c.Cin, c.Cout = N, H, W, Cin, Cout
                if _lib.load().hc_conv_s2_dgrad_supported(C.byref(c)):
                    s2g = c
            self.desc[key] = (f3, f1, dg, sf, sd, rows, s2d, s2g)
        rows = self.desc[key][5]
        s2, s2g = self.desc[key][6] is not None, self.desc[key][7] is not None
        if rows != self.rows_image or s2 != self.s2 or s2g != self.s2_dgrad:    # this geometry reads other weight images: drop them, ensure_packed() rebuilds
            self.rows_image, self.s2, self.s2_dgrad, self.packed, self.packed_key = rows, s2, s2g, None, None
        return self.desc[key][:5]

    def s2_desc(self, N, Cin, H, W, Cout):
        """hc_conv_s2_desc of this geometry when the stride-2 row kernel takes its forward, else None."""
        self.descs(N, Cin, H, W, Cout)
        return self.desc[(N, Cin, H, W, Cout)][6]


def _stats_of(x):
    st = getattr(x, "_hc_stats", None)
    if st is not None:
        return st
    N, Cc, H, W = x.shape
    st = POOL.take((_lib.stat_replicas(), 2, Cc), x.device)
    with cv.profiled("bn_elementwise", 0.0, N * H * W * Cc * 2.0):
        check(_lib.load().hc_channel_stats(ptr(x), ptr(st), N * H * W, Cc, stream()), "hc_channel_stats")
    return st


# ------------------------------------------------------------------ the three MFMA passes of a block
# (kept as free functions so that the full-size parity tests drive exactly the launch paths the block uses)
def block_convs_forward(st, src, w3, w1, geom, stats=None, stem_cin=None):
    """y3 = conv3x3(src), y1 = conv1x1(src) (+ per-channel sum / sum of squares into stats[0] / stats[1]).
    ``geom`` = (N, Cin, H, W, Cout); ``src`` is NHWC bf16 (the im2col tensor for the stem, then ``stem_cin`` = 3)."""
    N, Cin, H, W, Cout = geom
    f3, f1, _, sf, _ = st.descs(N, Cin, H, W, Cout)
    OH, OW = f3.OH, f3.OW
    dev = src.device
    wp3, wp1, _ = st.ensure_packed(w3, w1)
    y3 = cv.empty_cl(N, Cout, OH, OW, dev)
    y1 = cv.empty_cl(N, Cout, OH, OW, dev)
    s2d = st.s2_desc(N, Cin, H, W, Cout)
    if s2d is not None:
        # stride-2 row kernel: `src` is the block input itself (the fp32 image batch for the stem - no column tensor)
        if s2d.x_nchw_f32 and (src.dtype != torch.float32 or not src.is_contiguous() or tuple(src.shape) != (N, Cin, H, W)):
            raise _lib.HipError("stride-2 stem kernel expects the contiguous NCHW fp32 image batch")
        i3, i1 = st.s2_images
        s2d.x, s2d.w3img, s2d.w1img, s2d.y3, s2d.y1 = ptr(src), ptr(i3), ptr(i1), ptr(y3), ptr(y1)
        s2d.stats3 = None if stats is None else ptr(stats[0])
        s2d.stats1 = None if stats is None else ptr(stats[1])
        flops = 2.0 * N * OH * OW * Cout * 10 * Cin
        nbytes = src.numel() * src.element_size() + 2 * y3.numel() * 2.0
        with cv.profiled("conv_s2", flops, nbytes):
            check(_lib.load().hc_conv_s2_fwd(C.byref(s2d), stream()), "hc_conv_s2_fwd")
        return y3, y1
    fl3 = fl1 = None
    if stem_cin is not None:  # algorithmic flops of the real 3x3 / 1x1 convs, not of the padded im2col GEMM
        fl3, fl1 = 2.0 * N * OH * OW * Cout * 9 * stem_cin, 2.0 * N * OH * OW * Cout * stem_cin
    s3 = None if stats is None else stats[0]
    s1 = None if stats is None else stats[1]
    if sf is not None:
        cv.launch_conv_small_fwd(sf, src, wp3, wp1, y3, y1, s3, s1)
    elif st.stack_fwd:
        f31 = st.stacked_desc(N, Cin, H, W, Cout)
        cv.launch_conv(f31, src, wp3, y3, stats=s3, dst2=y1, stats2=s1, co_split=Cout,
                       flops=None if fl3 is None else fl3 + fl1)
    else:
        cv.launch_conv(f3, src, wp3, y3, stats=s3, flops=fl3)
        cv.launch_conv(f1, src, wp1, y1, stats=s1, flops=fl1)
    return y3, y1


def block_dgrad(st, dy3, dy1, dxid, w3, w1, geom):
    """dx = conv3x3^T(dy3) + conv1x1^T(dy1) [+ dxid] in one accumulator."""
    N, Cin, H, W, Cout = geom
    _, _, dg, _, sdg = st.descs(N, Cin, H, W, Cout)
    wpd = st.ensure_packed(w3, w1)[2]
    dx = cv.empty_cl(N, Cin, H, W, dy3.device)
    s2g = st.desc[(N, Cin, H, W, Cout)][7]
    if s2g is not None:
        if dxid is not None:
            raise _lib.HipError("stride-2 data gradient: a stride-2 RepBlock has no identity branch")
        s2g.dy3, s2g.dy1, s2g.wimg, s2g.dx = ptr(dy3), ptr(dy1), ptr(st.s2_dimg), ptr(dx)
        with cv.profiled("conv_s2", 2.0 * N * dy3.shape[2] * dy3.shape[3] * Cout * 10 * Cin, dx.numel() * 2.0 + 2 * dy3.numel() * 2.0):
            check(_lib.load().hc_conv_s2_dgrad(C.byref(s2g), stream()), "hc_conv_s2_dgrad")
        return dx
    if sdg is not None:
        cv.launch_conv_small_dgrad(sdg, dy3, dy1, wpd, dx, resid=dxid)
    else:
        cv.launch_conv(dg, dy3, wpd, dx, src1=dy1, resid=dxid)
    return dx


def block_wgrad(st, src, dy3, dy1, w3, w1, geom, stem_cin=None, defer=False):
    """(dW3, dW1) fp32 OIHW from the block input ``src`` and the two branch gradients.  ``defer`` (only inside an autograd
    backward pass): the fused kernel's launch may be queued and grouped with same-shaped blocks (ops/conv.py _RepWgradQueue)."""
    N, Cin, H, W, Cout = geom
    lib = _lib.load()
    npix = dy3.shape[0] * dy3.shape[2] * dy3.shape[3]
    if stem_cin is not None and src.dtype == torch.float32:
        # the stride-2 stem kernels read the image batch itself (csrc/conv_s2.hip): no column tensor
        dw3 = torch.empty_like(w3, dtype=torch.float32)
        dw1 = torch.empty_like(w1, dtype=torch.float32)
        ws = torch.empty((int(lib.hc_conv_s2_stem_wgrad_ws_bytes()),), dtype=torch.uint8, device=src.device)
        with cv.profiled("conv_wgrad", 2.0 * npix * Cout * 10 * stem_cin, src.numel() * 4.0 + 2 * dy3.numel() * 2.0):
            rc = lib.hc_conv_s2_stem_wgrad(ptr(src), ptr(dy3), ptr(dy1), ptr(dw3), ptr(dw1), ptr(ws), N, H, W, 0, stream())
        if rc == 0:
            return dw3, dw1
        src = cv.im2col_small(src, 3, 3, st.stride, 1, STEM_KPAD)      # HC_CONV_S2_STEM_WGRAD=0 / other geometry: GEMM over the column tensor
    if stem_cin is not None:
        K = STEM_KPAD
        dwc3 = cv.conv_wgrad(src, dy3, K, Cout, 1, 1, 1, 0, flops=2.0 * npix * Cout * 9 * stem_cin)
        dwc1 = cv.conv_wgrad(src, dy1, K, Cout, 1, 1, 1, 0, flops=2.0 * npix * Cout * stem_cin)
        dw3 = torch.empty_like(w3, dtype=torch.float32)
        check(lib.hc_unpack_im2col_grad(ptr(dwc3), ptr(dw3), Cout, stem_cin, 3, 3, K, 0, stream()), "hc_unpack_im2col_grad")
        dw1 = dwc1.view(Cout, K)[:, 4 * stem_cin:5 * stem_cin].reshape(Cout, stem_cin, 1, 1).contiguous()
        return dw3, dw1
    fused = cv.rep_block_wgrad(src, dy3, dy1, w3, w1, st.stride, defer=defer)   # one launch for both (and for same-shaped blocks)
    if fused is not None:
        return fused
    dw3 = cv.conv_wgrad(src, dy3, Cin, Cout, 3, 3, st.stride, 1)
    dw1 = cv.conv_wgrad(src, dy1, Cin, Cout, 1, 1, st.stride, 0)
    return dw3, dw1


def stem_fused_desc(st, src, w3, w1, geom):
    """hc_stem_desc when the fused stem kernels take this block (the stride-2 stem kernel's geometry at 224 x 224, 48 channels), else
    None.  ``src`` is the contiguous fp32 NCHW image batch."""
    N, Cin, H, W, Cout = geom
    if Cin != 3 or Cout != 48 or st.s2_desc(N, Cin, H, W, Cout) is None or src.dtype != torch.float32 or not src.is_contiguous():
        return None
    d = StemDesc()
    d.N, d.H, d.W = N, H, W
    if not _lib.load().hc_stem_fused_supported(C.byref(d)):
        return None
    st.ensure_packed(w3, w1)
    i3, i1 = st.s2_images
    d.x, d.w3img, d.w1img = ptr(src), ptr(i3), ptr(i1)
    return d


class RepBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w3, w1, g3, b3, g1, b1, g0, b0, st, relu):
        lib = _lib.load()
        cv._WREP.note_forward()       # a queue still armed by a backward pass that died is emptied here
        Cout, Cin = w3.shape[0], w3.shape[1]
        N, _, H, W = x.shape
        dev = x.device
        stem = (Cin % 16) != 0
        f3, f1, _, sf, _ = st.descs(N, Cin, H, W, Cout)
        OH, OW = (f3.OH, f3.OW)
        x_stats = None
        stem_direct = False
        if stem:
            if st.identity:
                raise NotImplementedError("identity branch with Cin % 16 != 0")
            stem_direct = st.s2_desc(N, Cin, H, W, Cout) is not None
            if stem_direct:       # the stride-2 row kernel reads the image batch itself; the column tensor is built in backward
                src = x.detach()
                if src.dtype != torch.float32 or not src.is_contiguous():
                    src = src.float().contiguous()
            else:
                src = cv.im2col_small(x, 3, 3, st.stride, 1, STEM_KPAD)
        else:
            src = cv.to_cl_bf16(x)
            if st.identity and st.training:
                x_stats = _stats_of(src if src is not x else x)
        if w3.dtype != torch.float32 or not w3.is_contiguous() or not w1.is_contiguous():
            raise RuntimeError("RepBlock (HIP) expects contiguous fp32 conv weights")
        R = _lib.stat_replicas()
        stats = POOL.take((2, R, 2, Cout), dev) if st.training else None
        # the stem fused with its BatchNorm passes (csrc/conv_s2.hip stem_fused_kernel): both convs are recomputed from the image in
        # every pass, y3 / y1 never exist in HBM
        sfd = stem_fused_desc(st, src, w3, w1, (N, Cin, H, W, Cout)) if stem_direct else None
        if sfd is not None:
            y3 = y1 = None
            if st.training:
                fl = 2.0 * N * OH * OW * Cout * 10 * Cin
                with cv.profiled("conv_s2", fl, src.numel() * 4.0):
                    check(lib.hc_stem_stats(C.byref(sfd), ptr(stats[0]), ptr(stats[1]), stream()), "hc_stem_stats")
        else:
            y3, y1 = block_convs_forward(st, src, w3, w1, (N, Cin, H, W, Cout), stats, Cin if stem else None)

        coef = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        save = torch.empty((6, Cout), dtype=torch.float32, device=dev)
        d = RepBnDesc()
        gammas, betas = (g3, g1, g0), (b3, b1, b0)
        nb = 3 if st.identity else 2
        for b in range(3):
            live = b < nb
            d.gamma[b] = ptr(gammas[b]) if live else None
            d.beta[b] = ptr(betas[b]) if live else None
            rm, rv, nbt = st.bn[b] if live else (None, None, None)
            d.running_mean[b], d.running_var[b], d.num_batches_tracked[b] = ptr(rm), ptr(rv), ptr(nbt)
            d.stats[b] = None
        if st.training:
            d.stats[0], d.stats[1] = ptr(stats[0]), ptr(stats[1])
            if st.identity:
                d.stats[2] = ptr(x_stats)
        d.coef, d.save, d.C, d.count = ptr(coef), ptr(save), Cout, N * OH * OW
        d.eps, d.momentum, d.training = st.eps, st.momentum, 1 if st.training else 0
        with cv.profiled("bn_finalize", 0.0, 0.0):
            check(lib.hc_rep_bn_finalize(C.byref(d), stream()), "hc_rep_bn_finalize")

        out = cv.empty_cl(N, Cout, OH, OW, dev)
        out_stats = POOL.take((_lib.stat_replicas(), 2, Cout), dev) if (st.emit_stats and st.training) else None
        # backward's reduction target, zeroed with the rest of the arena; re-validated in backward (ZeroPool.claim)
        ctx.red, ctx.red_gen = POOL.take_for_backward((_lib.stat_replicas(), 4, Cout), dev) if st.training else (None, -1)
        tb = N * OH * OW * Cout * 2.0            # bytes of one activation tensor of this block
        if sfd is not None:
            with cv.profiled("bn_elementwise", 0.0, src.numel() * 4.0 + tb):          # reads the image, writes out
                check(lib.hc_stem_apply(C.byref(sfd), ptr(coef), 1 if relu else 0, ptr(out), ptr(out_stats), stream()), "hc_stem_apply")
        else:
            with cv.profiled("bn_elementwise", 0.0, tb * (4 if st.identity else 3)):      # reads y3, y1 [, x], writes out
                check(lib.hc_rep_apply(ptr(y3), ptr(y1), ptr(src) if st.identity else None, ptr(coef), ptr(out), ptr(out_stats),
                                       N * OH * OW, Cout, 1 if relu else 0, stream()), "hc_rep_apply")
        ctx.st, ctx.relu, ctx.stem, ctx.stem_direct = st, relu, stem, stem_direct
        ctx.stem_fused = sfd is not None
        ctx.geom = (N, Cin, H, W, Cout, OH, OW)
        ctx.was_training = st.training
        # `out` is not kept for the backward: its ReLU mask is recomputed from (y3, y1, src, coef) by the *_z kernels
        ctx.save_for_backward(src, y3, y1, coef, save, g3, g1, g0 if st.identity else None, w3, w1)
        st.last_out_stats = out_stats
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        st = ctx.st
        src, y3, y1, coef, save, g3, g1, g0, w3, w1 = ctx.saved_tensors
        N, Cin, H, W, Cout, OH, OW = ctx.geom
        dev = g.device
        g = cv.to_cl_bf16(g)
        npix = N * OH * OW
        act = 1 if ctx.relu else 0
        xid = src if st.identity else None

        tb = npix * Cout * 2.0
        if ctx.stem_fused:
            # the fused stem: ONE pass over (image, g) gives G = dz^T X and the Gram matrix of the conv windows, from which a single-
            # workgroup kernel forms the BatchNorm parameter gradients and both weight gradients (csrc/conv_s2.hip, mode 2)
            if ctx.needs_input_grad[0]:
                raise NotImplementedError("input gradient of the stem block")
            sfd = stem_fused_desc(st, src, w3, w1, (N, Cin, H, W, Cout))
            if sfd is None:
                raise _lib.HipError("the stem's fused backward needs the weight images its forward ran on (HC_STEM_FUSED changed mid-step?)")
            dgam = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            dbet = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            dw3 = torch.empty_like(w3, dtype=torch.float32)
            dw1 = torch.empty_like(w1, dtype=torch.float32)
            ws = torch.empty((int(lib.hc_stem_bwd_ws_bytes()),), dtype=torch.uint8, device=dev)
            b = StemBwdDesc()
            b.coef, b.g, b.save, b.gamma3, b.gamma1, b.w3, b.w1 = ptr(coef), ptr(g), ptr(save), ptr(g3), ptr(g1), ptr(w3), ptr(w1)
            b.dgamma3, b.dbeta3, b.dgamma1, b.dbeta1 = ptr(dgam[0]), ptr(dbet[0]), ptr(dgam[1]), ptr(dbet[1])
            b.dw3, b.dw1, b.ws = ptr(dw3), ptr(dw1), ptr(ws)
            b.act, b.frozen, b.accumulate = act, 0 if ctx.was_training else 1, 0
            # family: the weight gradient's (its FLOPs are the conv weight gradients'; the BatchNorm reduce rides in the same pass)
            with cv.profiled("conv_wgrad", 2.0 * npix * Cout * 10 * Cin, src.numel() * 4.0 + tb):
                check(lib.hc_stem_bwd(C.byref(sfd), C.byref(b), stream()), "hc_stem_bwd")
            return (None, dw3, dw1, dgam[0], dbet[0], dgam[1], dbet[1], None, None, None, None)
        red = POOL.claim(ctx.red, ctx.red_gen, (_lib.stat_replicas(), 4, Cout), dev)
        ctx.red = None      # a second backward through this node (retain_graph) gets a fresh buffer
        with cv.profiled("bn_elementwise", 0.0, tb * (4 if st.identity else 3)):      # reads g, y3, y1 [, x]
            check(lib.hc_rep_bwd_reduce_z(ptr(g), ptr(coef), act, ptr(y3), ptr(y1), ptr(xid), ptr(red), npix, Cout, stream()),
                  "hc_rep_bwd_reduce_z")
        nb = 3 if st.identity else 2
        dgam = torch.empty((3, Cout), dtype=torch.float32, device=dev)
        dbet = torch.empty((3, Cout), dtype=torch.float32, device=dev)
        bcoef = torch.empty((9, Cout), dtype=torch.float32, device=dev)
        d = RepBnBwdDesc()
        d.red, d.save, d.bcoef = ptr(red), ptr(save), ptr(bcoef)
        gam = (g3, g1, g0)
        for b in range(3):
            live = b < nb
            d.gamma[b] = ptr(gam[b]) if live else None
            d.dgamma[b] = ptr(dgam[b]) if live else None
            d.dbeta[b] = ptr(dbet[b]) if live else None
        d.C, d.count, d.has_identity, d.accumulate = Cout, npix, 1 if st.identity else 0, 0
        d.frozen = 0 if ctx.was_training else 1       # eval mode / freeze_bn: running statistics, dy = a * dz
        with cv.profiled("bn_finalize", 0.0, 0.0):
            check(lib.hc_rep_bn_bwd_finalize(C.byref(d), stream()), "hc_rep_bn_bwd_finalize")

        dy3 = torch.empty_like(y3)
        dy1 = torch.empty_like(y1)
        dxid = torch.empty_like(src) if st.identity else None
        with cv.profiled("bn_elementwise", 0.0, tb * (7 if st.identity else 5)):      # reads g, y3, y1 [, x], writes dy3, dy1 [, dx_id]
            check(lib.hc_rep_bwd_apply_z(ptr(g), ptr(coef), act, ptr(y3), ptr(y1), ptr(xid), ptr(bcoef), ptr(dy3), ptr(dy1),
                                         ptr(dxid), npix, Cout, stream()), "hc_rep_bwd_apply_z")

        dx = None
        geom = (N, Cin, H, W, Cout)
        if ctx.needs_input_grad[0]:
            if ctx.stem:
                raise NotImplementedError("input gradient of the im2col stem path")
            dx = block_dgrad(st, dy3, dy1, dxid, w3, w1, geom)

        with cv.side_stream_for_wgrad((w3, w1), (src, dy3, dy1)) as side:
            dw3, dw1 = block_wgrad(st, src, dy3, dy1, w3, w1, geom, Cin if ctx.stem else None, defer=True)
            side.produced(dw3, dw1)
        return (dx, dw3, dw1, dgam[0], dbet[0], dgam[1], dbet[1],
                dgam[2] if st.identity else None, dbet[2] if st.identity else None, None, None)


def rep_block_forward(x, w3, w1, bn3, bn1, bn0, st, relu=True):
    """x: logical NCHW tensor.  bnX: nn.BatchNorm2d modules (bn0 may be None)."""
    st.bn = [(bn3.running_mean, bn3.running_var, bn3.num_batches_tracked),
             (bn1.running_mean, bn1.running_var, bn1.num_batches_tracked),
             (bn0.running_mean, bn0.running_var, bn0.num_batches_tracked) if bn0 is not None else (None, None, None)]
    st.eps = bn3.eps
    st.momentum = 0.1 if bn3.momentum is None else bn3.momentum
    st.training = bn3.training
    if bn1.training != bn3.training or (bn0 is not None and bn0.training != bn3.training):
        raise NotImplementedError("RepBlock (HIP): the BatchNorm layers of one block must all be in the same mode (freeze the "
                                  "whole block, e.g. freeze_model(model, 'features.2.1'), not a single branch)")
    out = RepBlockFn.apply(x, w3, w1, bn3.weight, bn3.bias, bn1.weight, bn1.bias,
                           bn0.weight if bn0 is not None else None, bn0.bias if bn0 is not None else None, st, relu)
    if st.last_out_stats is not None:
        out._hc_stats = st.last_out_stats  # consumed by the next block's identity BatchNorm
        st.last_out_stats = None
    return out
