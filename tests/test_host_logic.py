"""CPU: host-side logic that does not need a GPU — gather-conv descriptors (interpreted by a tiny
torch emulator and compared with F.conv2d / autograd), chunk tables, C-ABI surface."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from holocron_amd import _lib
from holocron_amd.ops import conv as cv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _s8(v):
    return v - 256 if v >= 128 else v


def emulate(d, src0, src1, wpk):
    """Interpret an hc_conv_desc on CPU.  src*: [N, IH, IW, srcC] fp32; wpk: [Cout, T, srcC] fp32."""
    dst = torch.zeros((d.N, d.OH, d.OW, d.Cout))
    written = torch.zeros((d.OH, d.OW), dtype=torch.int32)
    for c in range(d.nclass):
        cl = d.cls[c]
        ii = torch.arange(cl.OHg)
        jj = torch.arange(cl.OWg)
        oy, ox = ii * cl.ostep + cl.oy0, jj * cl.ostep + cl.ox0
        assert (oy < d.OH).all() and (ox < d.OW).all()
        written[oy[:, None], ox[None, :]] += 1
        for t in range(cl.ntaps):
            tp = cl.tap[t] & 0xffffffff
            dy, dx, s, wt = _s8(tp & 0xff), _s8((tp >> 8) & 0xff), (tp >> 16) & 0xff, (tp >> 24) & 0xff
            src = src1 if s else src0
            iy, ix = ii * cl.istep + dy, jj * cl.istep + dx
            vy, vx = (iy >= 0) & (iy < d.IH), (ix >= 0) & (ix < d.IW)
            if not vy.any() or not vx.any():
                continue
            g = src[:, iy[vy]][:, :, ix[vx]]                       # [N, a, b, srcC]
            contrib = g @ wpk[:, wt, :].T                          # [N, a, b, Cout]
            yy, xx = oy[vy], ox[vx]
            dst[:, yy[:, None], xx[None, :], :] += contrib
    assert (written == 1).all(), "every output pixel must belong to exactly one class"
    return dst


def _pack_fwd(w):      # mode 0 of hc_pack_conv_weight
    Cout, Cin, KH, KW = w.shape
    return w.permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin)


def _pack_dgrad(w, T, tap0, out=None):   # mode 1
    Cout, Cin, KH, KW = w.shape
    if out is None:
        out = torch.zeros((Cin, T, Cout))
    flipped = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, KH * KW, Cout)
    out[:, tap0:tap0 + KH * KW] = flipped
    return out


@pytest.mark.parametrize("H,W,stride,k,pad", [(9, 9, 1, 3, 1), (10, 7, 2, 3, 1), (8, 8, 2, 1, 0), (7, 11, 1, 1, 0), (11, 11, 2, 3, 1)])
def test_forward_descriptor(H, W, stride, k, pad):
    torch.manual_seed(0)
    N, Cin, Cout = 2, 16, 8
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k)
    d = cv.fwd_desc(N, Cin, H, W, Cout, k, k, stride, pad)
    out = emulate(d, x.permute(0, 2, 3, 1), None, _pack_fwd(w)).permute(0, 3, 1, 2)
    assert torch.allclose(out, F.conv2d(x, w, None, stride, pad), atol=1e-4)


@pytest.mark.parametrize("H,W,stride", [(9, 9, 1), (10, 7, 2), (11, 11, 2), (8, 8, 2), (6, 6, 1)])
def test_dual_branch_dgrad_descriptor(H, W, stride):
    """dx of conv3x3(x) + conv1x1(x) (RepBlock) from one two-source gather-conv."""
    torch.manual_seed(1)
    N, Cin, Cout = 2, 16, 32
    x = torch.randn(N, Cin, H, W, requires_grad=True)
    w3, w1 = torch.randn(Cout, Cin, 3, 3), torch.randn(Cout, Cin, 1, 1)
    y3, y1 = F.conv2d(x, w3, None, stride, 1), F.conv2d(x, w1, None, stride, 0)
    g3, g1 = torch.randn_like(y3), torch.randn_like(y1)
    (dx,) = torch.autograd.grad((y3 * g3).sum() + (y1 * g1).sum(), x)
    d = cv.dgrad_desc(N, Cin, H, W, Cout, [(3, 3, 1, 0, 0), (1, 1, 0, 1, 9)], stride)
    wp = _pack_dgrad(w3, 10, 0)
    _pack_dgrad(w1, 10, 9, out=wp)
    out = emulate(d, g3.permute(0, 2, 3, 1), g1.permute(0, 2, 3, 1), wp).permute(0, 3, 1, 2)
    assert torch.allclose(out, dx, atol=1e-3)


def test_tap_packing_roundtrip():
    for dy, dx, s, wt in [(-1, -1, 0, 0), (1, 0, 1, 9), (0, 1, 0, 8), (-1, 1, 1, 11)]:
        tp = _lib.tap(dy, dx, s, wt) & 0xffffffff
        assert (_s8(tp & 0xff), _s8((tp >> 8) & 0xff), (tp >> 16) & 0xff, (tp >> 24) & 0xff) == (dy, dx, s, wt)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "holocron_hip.h")).read()
    declared = set(re.findall(r"\b(hc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/holocron_hip.h but missing from the library"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().hc_version().startswith(b"holocron_hip")


def test_struct_sizes_match_header_layout(tmp_path):
    """sizeof / offsetof of every struct of include/holocron_hip.h as the host C compiler lays it out (natural
    alignment, same rules as hipcc for these plain structs) against the ctypes mirrors of holocron_amd/_lib.py."""
    import subprocess
    pairs = {"hc_conv_class": _lib.ConvClass, "hc_conv_desc": _lib.ConvDesc, "hc_conv_small_desc": _lib.ConvSmallDesc,
             "hc_wgrad_desc": _lib.WgradDesc, "hc_pack_item": _lib.PackItem, "hc_rep_bn_desc": _lib.RepBnDesc,
             "hc_rep_bn_bwd_desc": _lib.RepBnBwdDesc, "hc_mt_chunk": _lib.MtChunk, "hc_adabelief_group": _lib.AdaBeliefGroup,
             "hc_lars_group": _lib.LarsGroup, "hc_drop_item": _lib.DropItem, "hc_adamx_group": _lib.AdamxGroup, "hc_lamb_group": _lib.LambGroup, "hc_msbn_branch": _lib.MsbnBranch,
             "hc_msbn_desc": _lib.MsbnDesc, "hc_msbn_io": _lib.MsbnIo, "hc_rep_wgrad_desc": _lib.RepWgradDesc,
             "hc_conv_s2_desc": _lib.ConvS2Desc, "hc_conv_s2_dgrad_desc": _lib.ConvS2DgradDesc, "hc_multi_copy_desc": _lib.MultiCopyDesc}
    last = {"hc_conv_desc": "co_split", "hc_rep_wgrad_desc": "accumulate", "hc_pack_item": "ld", "hc_rep_bn_desc": "c_valid", "hc_rep_bn_bwd_desc": "frozen",
            "hc_conv_small_desc": "mode", "hc_wgrad_desc": "beta", "hc_lamb_group": "mode", "hc_msbn_branch": "momentum", "hc_msbn_desc": "accumulate", "hc_msbn_io": "C",
            "hc_conv_s2_desc": "x_nchw_f32", "hc_conv_s2_dgrad_desc": "Cout", "hc_multi_copy_desc": "scale"}
    src = tmp_path / "sz.c"
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{os.path.join(ROOT, "include", "holocron_hip.h")}"', "int main(void) {"]
    for name in pairs:
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
    for name, field in last.items():
        lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in pairs.items():
        assert int(out[name]) == ctypes.sizeof(cls), name
    for name, field in last.items():
        assert int(out[f"{name}.{field}"]) == getattr(pairs[name], field).offset, (name, field)


def test_cpu_tensors_are_rejected_loudly():
    import holocron_amd as h
    with pytest.raises(_lib.HipError):
        h.nn.functional.hard_mish(torch.zeros(4))
    with pytest.raises(_lib.HipError):
        h.ops.boxes.box_iou(torch.zeros(1, 4), torch.zeros(1, 4))


def test_chunk_table():
    from holocron_amd.optim._multi_tensor import build_chunks
    p = torch.zeros(3 * _lib.HC_MT_CHUNK + 5)
    g = torch.zeros_like(p)
    tab, n = build_chunks([{"p": p, "g": g, "m": p, "s": p, "smax": None, "group": 2, "tensor": 7}])
    assert n == 4 and tab.numel() == 4 * ctypes.sizeof(_lib.MtChunk)
    arr = (_lib.MtChunk * 4).from_buffer_copy(tab.numpy().tobytes())
    assert [c.n for c in arr] == [_lib.HC_MT_CHUNK] * 3 + [5]
    assert arr[1].p == p.data_ptr() + 4 * _lib.HC_MT_CHUNK and arr[3].group == 2 and arr[3].tensor == 7
    assert arr[0].smax is None


def test_mixup_and_metrics_host_contract():
    """holocron/utils/data/collate.py:31-37: negative alpha is rejected; the device ops refuse CPU tensors loudly."""
    import holocron_amd as h
    with pytest.raises(ValueError):
        h.utils.data.Mixup(10, -0.5)
    m = h.utils.data.Mixup(10, 0.2)
    assert m.num_classes == 10 and m.alpha == 0.2
    with pytest.raises(_lib.HipError):
        m(torch.rand(2, 3, 4, 4), torch.tensor([1, 2]))
    with pytest.raises(_lib.HipError):
        h.utils.metrics.TopKAccuracy().update(torch.rand(2, 10), torch.tensor([1, 2]))
    assert h.utils.metrics.TopKAccuracy().compute() == (0.0, 0.0, 0)


def test_small_channel_conv_planner_accepts_and_rejects():
    """hc_conv_small_supported is pure host code (LDS budget, DMA instruction counts, tile geometry): the RepVGG-A0 shapes the
    persistent kernel is built for are accepted, everything else falls back to the row-unit image or the gather-conv (None)."""
    from holocron_amd.ops import conv as cv
    ok = [(256, 112, 112, 48, 48, 0), (256, 112, 112, 48, 48, 1), (256, 56, 56, 48, 48, 0), (256, 56, 56, 48, 48, 1),
          (4, 30, 30, 32, 48, 0), (2, 17, 17, 16, 64, 0), (1, 3, 128, 48, 40, 0)]       # persistent kernel: C <= 48, W <= 128
    no = [(2, 20, 200, 48, 48, 0),      # wider than a 128-pixel tile
          (2, 20, 112, 40, 48, 0),      # C not a multiple of 16
          (2, 20, 112, 48, 72, 0),      # more than 64 output channels in the persistent kernel
          (2, 28, 28, 96, 96, 0), (2, 40, 40, 96, 96, 0),   # 64+ channels without the row-unit image flag: no kernel (round 6:
          (256, 14, 14, 192, 192, 0), (8, 16, 16, 64, 64, 1),  # the image-resident kernel of round 2 is retired)
          (2, 7, 7, 1280, 1280, 0)]
    for a in ok:
        d = cv.conv_small_desc(*a)
        assert d is not None, a
        assert (d.N, d.H, d.W, d.C, d.Cout, d.mode) == a
    for a in no:
        assert cv.conv_small_desc(*a) is None, a


def test_weight_gradient_workspace_planner():
    """hc_conv_wgrad_ws_bytes (host): split-K slabs are whole fp32 copies of the weight tensor, so the workspace is a
    multiple of it, grows with the number of splits the planner picks, and a degenerate problem needs none of a negative size."""
    import ctypes as C
    from holocron_amd import _lib
    lib = _lib.load()

    def ws(N, Cin, H, Cout, k, s):
        d = _lib.WgradDesc()
        OH = (H + 2 * (k // 2) - k) // s + 1
        d.N, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = N, H, H, Cin, OH, OH, Cout
        d.KH = d.KW = k
        d.stride, d.pad = s, k // 2
        return int(lib.hc_conv_wgrad_ws_bytes(C.byref(d)))

    for (N, Cin, H, Cout, k, s) in [(256, 48, 112, 48, 3, 1), (256, 48, 112, 48, 1, 1), (256, 96, 28, 192, 3, 2),
                                    (256, 192, 14, 192, 3, 1), (256, 1280, 7, 1280, 3, 1), (2, 16, 9, 32, 3, 1)]:
        b = ws(N, Cin, H, Cout, k, s)
        slab = 4 * Cout * Cin * k * k
        assert b >= slab and b % 4 == 0, (N, Cin, H, Cout, k, s, b)
        assert b < 64 * slab + (1 << 20) or b < (1 << 31), (b, slab)       # bounded: never more than a few dozen slabs
    # more images -> at least as many splits for the same layer
    assert ws(256, 192, 14, 192, 3, 1) >= ws(8, 192, 14, 192, 3, 1)


def test_row_unit_conv_planner_accepts_and_rejects():
    """hc_conv_small_supported with HC_CONV_SMALL_ROWS_IMAGE (host code): the row-unit kernel takes 192 channels on maps up to 16 pixels
    wide and 96 channels up to 32, any height (round 4: a predicate - the two 224 x 224 stages keep their tuned instantiations, every
    other width / height runs the family form), forward and data gradient; without the flag the same shapes resolve to the
    gather-conv."""
    from holocron_amd.ops import conv as cv
    R = cv.ROWS_IMAGE
    for a in [(256, 14, 14, 192, 192, R), (256, 14, 14, 192, 192, R | 1), (256, 28, 28, 96, 96, R), (3, 28, 28, 96, 96, R | 1),
              (5, 28, 14, 192, 192, R), (2, 56, 28, 96, 96, R),
              (4, 14, 14, 96, 96, R), (4, 21, 14, 192, 192, R), (4, 7, 14, 192, 192, R | 1), (4, 70, 14, 192, 192, R), (3, 16, 16, 192, 192, R),
              (3, 32, 32, 96, 96, R | 1), (2, 5, 9, 96, 96, R),          # the family form: other widths, odd unit counts, ragged heights
              (256, 112, 112, 48, 48, R), (256, 56, 56, 48, 48, R), (2, 8, 112, 48, 48, R)]:   # streaming 48-channel kernel: forward
        d = cv.conv_small_desc(*a)
        assert d is not None and d.mode == a[5], a
    for a in [(4, 28, 28, 192, 192, R), (4, 14, 17, 192, 192, R), (4, 14, 40, 96, 96, R), (4, 14, 14, 128, 128, R), (4, 14, 14, 64, 64, R),
              (4, 14, 14, 192, 96, R), (4, 14, 14, 192, 192, R | 2),
              (4, 112, 112, 48, 48, R | 1),     # its data gradient stays on the persistent kernel unless HC_CONV_ROWS48=2
              (4, 110, 112, 48, 48, R), (4, 28, 28, 48, 48, R), (4, 60, 56, 48, 48, R)]:
        assert cv.conv_small_desc(*a) is None, a
    assert cv.conv_small_desc(256, 14, 14, 192, 192, 0) is None            # (the image-resident kernel of round 2 is retired)
    assert cv.rows_image(192, "cpu").shape == (60, 192, 32) and cv.rows_image(96, "cpu").shape == (30, 96, 32)
    assert cv.rows_image(48, "cpu").shape == (20, 48, 32) and float(cv.rows_image(48, "cpu").abs().sum()) == 0.0   # zero-filled K padding


def test_fused_weight_gradient_planner():
    """hc_rep_wgrad_supported / _plan / _ws_bytes (host code): tile choice by channel counts, the rows-per-step and prefetch depth
    fit the LDS, grouping more blocks into a launch lowers the pixel split, the workspace is whole fp32 slabs, wide layers are left to
    the k-pipelined kernel."""
    import ctypes as C
    from holocron_amd.ops.conv import _WREP
    lib = _lib.load()

    def plan(N, cin, H, cout, s, jobs):
        d = _WREP._desc((N, cin, H, H, cout, s), jobs)
        out = (C.c_int32 * 8)()
        if not lib.hc_rep_wgrad_supported(C.byref(d)):
            return None
        assert lib.hc_rep_wgrad_plan(C.byref(d), out) == 0
        return list(out), int(lib.hc_rep_wgrad_ws_bytes(C.byref(d)))

    p192, ws192 = plan(256, 192, 14, 192, 1, 14)
    assert p192[0:2] == [6, 6] and p192[6] <= 160 * 1024 and p192[2] * 14 <= 128     # the 96 x 96 tile where both widths allow it
    p1, ws1 = plan(256, 192, 14, 192, 1, 1)
    assert p1[4] > p192[4]                                   # one block alone needs a deeper pixel split than 14 grouped ones
    assert ws192 % (4 * 192 * 10 * 192) == 0 and ws1 % (4 * 192 * 10 * 192) == 0
    p48, ws48 = plan(256, 48, 112, 48, 1, 1)
    assert p48[0:2] == [3, 3] and p48[6] <= 160 * 1024 and ws48 % (4 * 48 * 10 * 48) == 0
    assert plan(256, 96, 28, 192, 2, 1)[0][0:2] == [6, 6] and plan(256, 64, 16, 64, 1, 2)[0][0:2] == [4, 4]
    assert plan(256, 96, 28, 144, 1, 1)[0][0:2] == [6, 3]    # 96 | Cin, 48 | Cout only: the 96 x 48 tile of rounds 2-4
    assert plan(256, 1280, 7, 1280, 1, 1) is None and plan(256, 40, 14, 48, 1, 1) is None and plan(4, 48, 14, 48, 3, 1) is None
    assert _WREP._desc((4, 48, 14, 14, 48, 1), 17).njobs == 17 and plan(4, 48, 14, 48, 1, 17) is None     # more than 16 blocks per launch


def test_pack_and_stacked_conv_argument_checks():
    """argument validation that runs on the host before anything is launched"""
    import ctypes as C
    lib = _lib.load()
    w = torch.zeros(8)
    # row-unit images need Cout == Cin, a multiple of 48 (rows) and of 32 (k blocks)
    assert lib.hc_pack_conv_weight(w.data_ptr(), w.data_ptr(), 40, 40, 3, 3, 3, 0, 10, None) == 1
    assert lib.hc_pack_conv_weight(w.data_ptr(), w.data_ptr(), 96, 48, 3, 3, 4, 0, 10, None) == 1
    assert lib.hc_pack_conv_weight(w.data_ptr(), w.data_ptr(), 64, 64, 3, 3, 3, 0, 10, None) == 1
    assert lib.hc_pack_conv_weight(w.data_ptr(), w.data_ptr(), 48, 48, 3, 3, 5, 0, 10, None) == 1
    # stacked convolutions: split on a 4-channel boundary, second destination required, no epilogue extras
    d = _lib.ConvDesc()
    d.src0 = d.wpk = d.dst = w.data_ptr()
    d.N, d.IH, d.IW, d.srcC, d.OH, d.OW, d.Cout, d.T, d.nclass = 1, 4, 4, 16, 4, 4, 96, 9, 1
    d.co_split = 48
    assert lib.hc_conv_gather(C.byref(d), None) == 1          # dst2 missing
    d.dst2 = w.data_ptr()
    d.co_split = 46
    assert lib.hc_conv_gather(C.byref(d), None) == 1          # not a multiple of 4
    d.co_split = 48
    d.bias = w.data_ptr()
    assert lib.hc_conv_gather(C.byref(d), None) == 1          # bias with a split
    # the z-mask BN backward wants the forward coefficients and a known activation code
    assert lib.hc_rep_bwd_reduce_z(w.data_ptr(), None, 1, w.data_ptr(), w.data_ptr(), None, w.data_ptr(), 8, 8, None) == 1
    assert lib.hc_rep_bwd_apply_z(w.data_ptr(), w.data_ptr(), 2, w.data_ptr(), w.data_ptr(), None, w.data_ptr(), w.data_ptr(), w.data_ptr(),
                                  None, 8, 8, None) == 1


def test_deterministic_switch_resizes_the_replica_count():
    """... and the per-step zero arena forgets its high-water mark on every change of the replica count: sized for 32768 replicas it
    would otherwise clear gigabytes per step for the rest of the process"""
    from holocron_amd.nn.repblock_op import POOL
    lib = _lib.load()
    assert _lib.stat_replicas() == 128 and lib.hc_get_deterministic() == 0
    POOL.high = 4096
    _lib.set_deterministic(True)
    try:
        assert _lib.stat_replicas() == 32768 and lib.hc_get_deterministic() == 1
        assert POOL.high == 0 and POOL.buf is None
        POOL.high = 1 << 30
        _lib.set_deterministic(True)            # no change: nothing is reset
        assert POOL.high == 1 << 30
    finally:
        _lib.set_deterministic(False)
    assert _lib.stat_replicas() == 128 and POOL.high == 0


def test_optimizer_launch_groups_by_step_count():
    from holocron_amd.optim._multi_tensor import VGroups
    vg = VGroups()
    assert [vg.index(0, 5), vg.index(0, 5), vg.index(1, 5), vg.index(0, 3), vg.index(1, 5)] == [0, 0, 1, 2, 1]
    assert vg.keys == [(0, 5), (1, 5), (0, 3)] and len(vg) == 3


# ------------------------------------------------------------------ deferred RepBlock weight gradients
class _FakeQueue:
    """Drives ops.conv._RepWgradQueue on CPU tensors: the launch is replaced by `+= 1` on the queued gradient buffers."""

    def __init__(self, monkeypatch):
        import numpy as np
        self.launches = []
        q = cv._WREP
        monkeypatch.setattr(q, "support", {})
        monkeypatch.setattr(q, "supported", lambda key: True)

        def launch(key, jobs, accumulate=False):
            self.launches.append((key, len(jobs), accumulate))
            Cout, Cin = key[4], key[1]
            for (_, _, _, p3, p1) in jobs:
                for p, n in ((p3, Cout * Cin * 9), (p1, Cout * Cin)):
                    a = np.ctypeslib.as_array((ctypes.c_float * n).from_address(p))
                    if accumulate:
                        a += 1.0
                    else:
                        a[:] = 1.0
        monkeypatch.setattr(q, "launch", launch)
        q.jobs, q.armed, q.task = [], False, -1
        q.parked, q.cb_tasks = {}, set()


def _rep_like(w3, w1, x, fail=False):
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w3, w1):
            ctx.save_for_backward(x, w3, w1)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            x, w3, w1 = ctx.saved_tensors
            dw3, dw1 = cv.rep_block_wgrad(x, g, g, w3, w1, 1, defer=True)
            return g, dw3, dw1

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")
    y = Fn.apply(x, w3, w1)
    if fail:
        y = Fn.apply(Boom.apply(y), w3, w1)      # the pass raises AFTER a job has been queued
    return y.sum()


def test_deferred_wgrad_queue_recovers_from_an_aborted_backward(monkeypatch):
    fq = _FakeQueue(monkeypatch)
    w3 = torch.nn.Parameter(torch.zeros(16, 16, 3, 3))
    w1 = torch.nn.Parameter(torch.zeros(16, 16, 1, 1))
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match="boom"):
        _rep_like(w3, w1, x, fail=True).backward()
    assert cv._WREP.armed and cv._WREP.jobs          # the engine skipped the final callback: stale state is still there
    w3.grad = w1.grad = None
    fq.launches.clear()
    _rep_like(w3, w1, x).backward()                  # the retry must not inherit `armed` (it would train on zero gradients)
    assert not cv._WREP.armed and not cv._WREP.jobs
    # the stale job is PARKED under the dead pass's task id (it cannot be told from the outer pass of a re-entrant backward, whose jobs
    # must not be lost: ADVICE r3) and never launched; only the retry's own job runs
    assert len(fq.launches) == 1 and all(n == 1 for _, n, _ in fq.launches)
    assert torch.equal(w3.grad, torch.ones_like(w3)) and torch.equal(w1.grad, torch.ones_like(w1))
    assert cv._WREP.parked                           # ... until the next forward outside a backward pass drops it
    cv._WREP.note_forward()
    assert not cv._WREP.parked and not cv._WREP.cb_tasks


def test_retained_graph_retry_without_a_forward_is_not_counted_twice(monkeypatch):
    """backward raised, and the caller retries on the RETAINED graph - no forward in between, no zero_grad (gradient
    accumulation).  The dead pass's job must not be launched from inside the retry: its placeholders were adopted as .grad, the
    retry accumulates its own gradient onto them, and a launch of the stale job would add the failed micro-batch a second time."""
    fq = _FakeQueue(monkeypatch)
    w3 = torch.nn.Parameter(torch.zeros(16, 16, 3, 3))
    w1 = torch.nn.Parameter(torch.zeros(16, 16, 1, 1))
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    boom = {"on": True}

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w3, w1):
            ctx.save_for_backward(x, w3, w1)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            x, w3, w1 = ctx.saved_tensors
            dw3, dw1 = cv.rep_block_wgrad(x, g, g, w3, w1, 1, defer=True)
            return g, dw3, dw1

    class MaybeBoom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            if boom["on"]:
                raise RuntimeError("boom")
            return g
    loss = Fn.apply(MaybeBoom.apply(Fn.apply(x, w3, w1)), w3, w1).sum()
    with pytest.raises(RuntimeError, match="boom"):
        loss.backward(retain_graph=True)             # the last Fn queued one job, then the pass died
    assert cv._WREP.armed and len(cv._WREP.jobs) == 1
    boom["on"] = False
    fq.launches.clear()
    loss.backward()                                  # same graph, new graph task, no forward: both Fn nodes queue again
    assert not cv._WREP.armed and not cv._WREP.jobs
    assert sum(n for _, n, _ in fq.launches) == 2    # the retry's two jobs; the dead pass's parked job was not launched
    # w3 is used by two nodes: gradient = 2 (one per node), exactly once each
    assert torch.equal(w3.grad, 2 * torch.ones_like(w3)) and torch.equal(w1.grad, 2 * torch.ones_like(w1))


def test_aborted_backward_is_not_counted_twice_under_gradient_accumulation(monkeypatch):
    """a micro-batch whose backward raised and is retried WITHOUT zero_grad.  RepBlockFn.forward calls note_forward(): the
    queue of the dead pass is emptied by the retry's forward (no backward is running there), so the retry's gradient lands in the
    zero placeholders once - launching the stale jobs from inside the retry would add the failed micro-batch a second time."""
    fq = _FakeQueue(monkeypatch)
    w3 = torch.nn.Parameter(torch.zeros(16, 16, 3, 3))
    w1 = torch.nn.Parameter(torch.zeros(16, 16, 1, 1))
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match="boom"):
        _rep_like(w3, w1, x, fail=True).backward()
    assert cv._WREP.armed and cv._WREP.jobs
    cv._WREP.note_forward()                                                # what the retry's first RepBlock forward does
    assert not cv._WREP.armed and not cv._WREP.jobs
    fq.launches.clear()
    _rep_like(w3, w1, x).backward()
    assert torch.equal(w3.grad, torch.ones_like(w3)) and torch.equal(w1.grad, torch.ones_like(w1))
    assert len(fq.launches) == 1                                           # the retry's own job only: the dead pass's was dropped


def test_note_forward_leaves_a_running_pass_alone(monkeypatch):
    """The recomputation forward of a re-entrant checkpoint runs INSIDE a backward pass: note_forward() must not drop the outer jobs."""
    fq = _FakeQueue(monkeypatch)
    ws = [(torch.nn.Parameter(torch.zeros(16, 16, 3, 3)), torch.nn.Parameter(torch.zeros(16, 16, 1, 1))) for _ in range(2)]
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            before = len(cv._WREP.jobs)
            cv._WREP.note_forward()
            seen.append((before, len(cv._WREP.jobs), cv._WREP.armed))
            return g
    y = _rep_like_y(*ws[1], Probe.apply(_rep_like_y(*ws[0], x)))          # backward order: block 1 queues, Probe, block 0 queues
    y.sum().backward()
    assert seen == [(1, 1, True)]
    for w3, w1 in ws:
        assert torch.equal(w3.grad, torch.ones_like(w3)) and torch.equal(w1.grad, torch.ones_like(w1))


def _rep_like_y(w3, w1, x):
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w3, w1):
            ctx.save_for_backward(x, w3, w1)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            x, w3, w1 = ctx.saved_tensors
            dw3, dw1 = cv.rep_block_wgrad(x, g, g, w3, w1, 1, defer=True)
            return g, dw3, dw1
    return Fn.apply(x, w3, w1)


def test_deferred_wgrad_queue_survives_a_reentrant_backward(monkeypatch):
    """torch.utils.checkpoint(use_reentrant=True) runs a nested backward (a new graph task) inside the outer pass: the jobs the outer
    pass queued before it must still be launched (they were dropped and the zero placeholders stayed in .grad)."""
    from torch.utils.checkpoint import checkpoint
    fq = _FakeQueue(monkeypatch)
    ws = [(torch.nn.Parameter(torch.zeros(16, 16, 3, 3)), torch.nn.Parameter(torch.zeros(16, 16, 1, 1))) for _ in range(3)]
    x = torch.ones(2, 16, 4, 4, requires_grad=True)

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w3, w1):
            ctx.save_for_backward(x, w3, w1)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            x, w3, w1 = ctx.saved_tensors
            dw3, dw1 = cv.rep_block_wgrad(x, g, g, w3, w1, 1, defer=True)
            return g, dw3, dw1

    y = Fn.apply(x, *ws[0])                                         # backward order: block 2 (outer), block 1 (nested), block 0 (outer)
    y = checkpoint(lambda t: Fn.apply(t, *ws[1]), y, use_reentrant=True)
    y = Fn.apply(y, *ws[2])
    y.sum().backward()
    assert not cv._WREP.armed and not cv._WREP.jobs
    for w3, w1 in ws:
        assert torch.equal(w3.grad, torch.ones_like(w3)) and torch.equal(w1.grad, torch.ones_like(w1))


def test_deferred_wgrad_not_used_when_something_reads_gradients_inside_the_pass(monkeypatch):
    fq = _FakeQueue(monkeypatch)
    w3 = torch.nn.Parameter(torch.zeros(16, 16, 3, 3))
    w1 = torch.nn.Parameter(torch.zeros(16, 16, 1, 1))
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    seen = []
    h = w3.register_post_accumulate_grad_hook(lambda p: seen.append(float(p.grad.sum())))
    _rep_like(w3, w1, x).backward()
    assert seen == [float(w3.numel())]               # the hook saw the real gradient: the launch was immediate, not deferred
    assert fq.launches[0][2] is False
    h.remove()
    # a hook registered as flush-aware (parallel.GradReducer's) keeps deferral on
    w3.grad = w1.grad = None
    fq.launches.clear()
    h = w3.register_post_accumulate_grad_hook(lambda p: cv.flush_deferred_wgrads())
    cv.register_flush_aware_hook(w3, h)
    _rep_like(w3, w1, x).backward()
    assert fq.launches and all(acc for _, _, acc in fq.launches)
    assert torch.equal(w3.grad, torch.ones_like(w3))
    h.remove()
    # tensor hooks and the process-wide switch
    w3.grad = w1.grad = None
    fq.launches.clear()
    h = w1.register_hook(lambda g: g)
    _rep_like(w3, w1, x).backward()
    assert fq.launches[0][2] is False
    h.remove()
    w3.grad = w1.grad = None
    fq.launches.clear()
    cv.set_deferred_wgrads(False)
    try:
        _rep_like(w3, w1, x).backward()
        assert fq.launches[0][2] is False
    finally:
        cv.set_deferred_wgrads(True)


def test_optimizer_step_flushes_a_queue_the_pass_left_behind(monkeypatch):
    fq = _FakeQueue(monkeypatch)
    w3 = torch.nn.Parameter(torch.zeros(16, 16, 3, 3))
    w1 = torch.nn.Parameter(torch.zeros(16, 16, 1, 1))
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match="boom"):
        _rep_like(w3, w1, x, fail=True).backward()               # a job is queued and the engine never runs the final callback
    assert cv._WREP.jobs
    from holocron_amd.optim import AdaBelief
    opt = AdaBelief([torch.nn.Parameter(torch.zeros(4))])        # no gradients: step() returns right after the flush
    opt.step()
    assert not cv._WREP.jobs and len(fq.launches) == 1
    cv._WREP.armed, cv._WREP.task = False, -1


def test_copy_all_only_swallows_the_fused_path_refusals(monkeypatch):
    from holocron_amd import parallel

    def boom(dst, src):
        raise RuntimeError("HIP error: out of memory")
    monkeypatch.setattr(torch, "_foreach_copy_", boom)
    with pytest.raises(RuntimeError, match="out of memory"):
        parallel._copy_all([torch.zeros(2)], [torch.ones(2)])

    def refuse(dst, src):
        raise RuntimeError("_foreach_copy_: tensors must be on the same device")
    monkeypatch.setattr(torch, "_foreach_copy_", refuse)
    d = [torch.zeros(2)]
    parallel._copy_all(d, [torch.ones(2)])
    assert torch.equal(d[0], torch.ones(2))


def test_trainer_load_keeps_the_optimizer_state_through_reset_opt(tmp_path):
    """`load()` restored the optimizer state and the `_reset_opt()` at the top of fit_n_epochs threw it away."""
    from holocron_amd.trainer import ClassificationTrainer

    def make():
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 3))
        return m, torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(0)
    data = [(torch.randn(4, 3, 2, 2, generator=g), torch.randint(0, 3, (4,), generator=g)) for _ in range(3)]
    m, opt = make()
    tr = ClassificationTrainer(m, data, data[:1], torch.nn.CrossEntropyLoss(), opt, gpu=None, output_file=str(tmp_path / "c.pth"))
    tr.fit_n_epochs(1, 0.1, sched_type="cosine")
    tr.save(str(tmp_path / "full.pth"), with_optimizer=True)
    bufs = [opt.state[p]["momentum_buffer"].clone() for p in opt.param_groups[0]["params"]]
    m2, opt2 = make()
    tr2 = ClassificationTrainer(m2, data, data[:1], torch.nn.CrossEntropyLoss(), opt2, gpu=None, output_file=str(tmp_path / "d.pth"))
    tr2.load(torch.load(str(tmp_path / "full.pth"), weights_only=False))
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        tr2._reset_opt(0.05, norm_weight_decay=0.0)  # a different grouping (norm parameters split off) than the saved one
    assert len(opt2.param_groups) == 2 and opt2.param_groups[0]["lr"] == 0.05
    if not all(a is b for a, b in zip([p for grp in opt2.param_groups for p in grp["params"]], m2.parameters())):
        # same count, another order: moments must not be attached by position to parameters of other shapes
        assert any("another order" in str(w.message) for w in rec) and len(opt2.state) == 0
    # saved order = model.parameters() order; the regrouped optimizer holds the same tensors' state, matched by POSITION in the
    # saved flattening - which is only valid when the order is kept, so compare through the parameters themselves
    saved_order = list(m2.parameters())
    flat_now = [p for grp in opt2.param_groups for p in grp["params"]]
    assert len(flat_now) == len(saved_order)
    if all(a is b for a, b in zip(flat_now, saved_order)):
        for p, b in zip(saved_order, bufs):
            assert torch.equal(opt2.state[p]["momentum_buffer"], b)
    tr2._reset_opt(0.05)                             # one shot: the second reset starts fresh again (reference semantics)
    assert len(opt2.state) == 0
    # same grouping as saved: every momentum buffer comes back
    m3, opt3 = make()
    tr3 = ClassificationTrainer(m3, data, data[:1], torch.nn.CrossEntropyLoss(), opt3, gpu=None, output_file=str(tmp_path / "e.pth"))
    tr3.load(torch.load(str(tmp_path / "full.pth"), weights_only=False))
    tr3._reset_opt(0.02)
    for p, b in zip(opt3.param_groups[0]["params"], bufs):
        assert torch.equal(opt3.state[p]["momentum_buffer"], b)
    assert opt3.param_groups[0]["lr"] == 0.02


def test_zero_pool_keeps_retired_arenas_until_the_last_graph_holder_releases():
    """GraphedStep.release() freed every retired statistics arena of the process, also those another live GraphedStep's
    graphs still replay against."""
    from holocron_amd.nn.repblock_op import ZeroPool
    from holocron_amd import parallel
    pool = ZeroPool()
    pool.retired = [torch.zeros(4)]
    pool.graph_users = 2
    pool.release_retired()
    assert pool.graph_users == 1 and len(pool.retired) == 1       # one holder left: nothing is freed
    pool.release_retired()
    assert pool.graph_users == 0 and pool.retired == []
    pool.release_retired()                                          # a release without a capture is harmless
    assert pool.graph_users == 0
    gs = parallel.GraphedStep(lambda: None, torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1))
    from holocron_amd.nn.repblock_op import POOL
    users = POOL.graph_users
    gs.release()                                                    # never captured: does not touch the count
    assert POOL.graph_users == users


def test_multi_copy_launch_plan_covers_every_element_once():
    """parallel._copy_pieces (the piece table of hc_multi_copy): tensors are split in order into pieces of at most `piece` elements,
    at most `per_launch` pieces per launch, byte offsets follow the element sizes of either side."""
    from holocron_amd import parallel
    items = [(1000, 50000, 10, 4, 2), (2000, 60000, 0, 4, 2), (3000, 70000, 25, 4, 2), (4000, 80000, 8, 4, 2)]
    plan = parallel._copy_pieces(items, piece=8, per_launch=3)
    assert all(1 <= len(l) <= 3 for l in plan) and all(len(l) == 3 for l in plan[:-1])
    flat = [p for l in plan for p in l]
    assert flat == [(1000, 50000, 8), (1032, 50016, 2),
                    (3000, 70000, 8), (3032, 70016, 8), (3064, 70032, 8), (3096, 70048, 1),
                    (4000, 80000, 8)]
    assert parallel._copy_pieces([], 8, 3) == []


def test_squeeze_excite_mlp_routing_predicate():
    """nn/mbconv_op.se_mlp_fusable: the fused hc_se_mlp_* launches take exactly the layout the reference builds (rexnet.py:49-61:
    1x1 conv without bias, training-mode BatchNorm2d, ReLU / ReLU6, 1x1 conv); anything else stays on the generic units."""
    from torch import nn

    from holocron_amd.models.classification.rexnet import SEBlock
    from holocron_amd.nn.mbconv_op import se_mlp_fusable
    m = SEBlock(228, 12, nn.ReLU6(inplace=True), nn.BatchNorm2d).train()
    c1, bn, act, c2 = list(m.conv)[:4]
    assert se_mlp_fusable(c1, bn, act, c2)
    m.eval()
    assert not se_mlp_fusable(c1, bn, act, c2)                       # running statistics: the generic units
    m.train()
    assert not se_mlp_fusable(c1, bn, nn.SiLU(), c2)                 # an activation the kernels do not fuse
    assert not se_mlp_fusable(nn.Conv2d(228, 19, 1, bias=True), bn, act, c2)
    assert not se_mlp_fusable(c1, nn.BatchNorm2d(19, momentum=None), act, c2)
    wide = SEBlock(2400, 12, nn.ReLU6(), nn.BatchNorm2d).train()     # 200 reduced channels: above the kernels' 128
    assert not se_mlp_fusable(*list(wide.conv)[:4])



# ------------------------------------------------------------------ deferred, grouped weight gradients of the plain conv units (round 6)
def test_conv_unit_wgrad_queue_groups_same_shaped_layers(monkeypatch):
    """ops.conv._ConvWgradQueue on CPU tensors (the launch replaced by `+= njobs-th of 1`): same-shaped layers of one backward pass go
    out as ONE grouped launch at the end of the pass, other shapes as their own, gradients land in the tensors autograd adopted."""
    import numpy as np
    q = cv._WCONV
    launches = []
    monkeypatch.setattr(q, "support", {})
    monkeypatch.setattr(q, "supported", lambda key: True)

    def launch(key, jobs, accumulate=False):
        launches.append((key, len(jobs), accumulate))
        Cout, Cin, KH, KW = key[4], key[1], key[5], key[6]
        for (_, _, _, p, _) in jobs:
            a = np.ctypeslib.as_array((ctypes.c_float * (Cout * Cin * KH * KW)).from_address(p))
            a += 1.0
    monkeypatch.setattr(q, "launch", launch)
    q.jobs, q.armed, q.task, q.parked, q.cb_tasks = [], False, -1, {}, set()

    def unit(x, w):
        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                x, w = ctx.saved_tensors
                Cout, Cin, KH, KW = w.shape
                return g, cv.conv_wgrad_unit(x, g, w, Cin, Cout, KH, KW, 1, KH // 2)
        return Fn.apply(x, w)
    ws = [torch.nn.Parameter(torch.zeros(16, 16, 3, 3)) for _ in range(3)] + [torch.nn.Parameter(torch.zeros(16, 16, 1, 1))]
    x = torch.ones(2, 16, 4, 4, requires_grad=True)
    y = x
    for w in ws:
        y = unit(y, w)
    y.sum().backward()
    assert not q.armed and not q.jobs
    assert sorted((k[5], n, acc) for k, n, acc in launches) == [(1, 1, True), (3, 3, True)]
    for w in ws:
        assert torch.equal(w.grad, torch.ones_like(w))
    # a parameter that already has a gradient (accumulation), or a hook that reads gradients inside the pass: computed at once
    launches.clear()
    seen = []
    monkeypatch.setattr(cv, "conv_wgrad", lambda x, dy, Cin, Cout, KH, KW, s, p, **kw: (seen.append((Cout, KH)) or torch.ones(Cout, Cin, KH, KW)))
    y = unit(x, ws[0])
    y.sum().backward()
    assert seen == [(16, 3)] and not launches and torch.equal(ws[0].grad, 2 * torch.ones_like(ws[0]))
    # flush_deferred_wgrads() from inside a pass launches what is queued so far; the pass's later layers still arrive
    for w in ws:
        w.grad = None
    launches.clear()

    class Peek(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            cv.flush_deferred_wgrads()
            seen.append(("mid", float(ws[1].grad.sum())))
            return g
    y = unit(Peek.apply(unit(x, ws[0])), ws[1])
    y.sum().backward()
    assert ("mid", float(ws[1].numel())) in seen
    assert torch.equal(ws[0].grad, torch.ones_like(ws[0])) and torch.equal(ws[1].grad, torch.ones_like(ws[1]))
    assert len(launches) == 2 and not q.armed and not q.jobs and not q.cb_tasks


def test_gather_conv_accumulators_stay_in_registers(tmp_path):
    """Compile-time guard (hipcc cross-compiles without a GPU): no instantiation of the gather-conv kernel may use scratch memory.  Its
    epilogue is a fully unrolled loop over the accumulator quads; when the loop body outgrows the compiler's pragma-unroll budget the
    accumulators are indexed dynamically and move to scratch - the kernel still computes the right values, 1.5 x slower (round 6: the
    192-channel tile, found only by the headline's family timings)."""
    import re
    import subprocess
    from holocron_amd import build as hb
    src = os.path.join(hb.CSRC, "conv_gather.hip")
    cmd = [hb.HIPCC] + hb.COMMON + hb.SOURCES["conv_gather.hip"] + ["-c", src, "-o", str(tmp_path / "cg.o"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)
    assert len(names) == len(scratch) and len(names) >= 20
    bad = [(n[:60], s) for n, s in zip(names, scratch) if "conv_gather_kernel" in n and s != "0"]
    assert not bad, bad
