"""GPU: the MFMA conv kernels against torch-CPU fp32 on bf16-representable operands.

Tolerances: results stored as bf16 carry a rounding of 2^-9 relative per element (RMS ~1.1e-3),
so bf16 outputs are checked to rel-L2 <= 3e-3; fp32 outputs (statistics, weight gradients) to 2e-4.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _to_nchw_f32(y):
    return y.float().cpu().contiguous()


CONV_CASES = [
    # N, Cin, Cout, H, W, k, stride
    (2, 16, 48, 13, 13, 3, 1),     # BK=16, 64-wide channel tile with padding
    (2, 48, 48, 14, 9, 3, 2),
    (3, 32, 96, 9, 9, 3, 1),       # BK=32, 96 tile
    (2, 64, 192, 14, 14, 3, 1),    # BK=64, 192 tile
    (2, 192, 192, 7, 7, 1, 1),
    (2, 64, 128, 10, 10, 3, 2),    # 128 tile
    (1, 128, 320, 7, 7, 3, 1),     # two channel tiles + ragged
    (2, 96, 40, 5, 5, 1, 2),       # Cout % 32 != 0
    (5, 16, 255, 6, 6, 1, 1),      # Cout % 4 != 0 (YOLO head width): scalar store path
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride", CONV_CASES)
def test_conv_forward_matches_cpu(N, Cin, Cout, H, W, k, stride):
    from holocron_amd.ops import conv as cv
    torch.manual_seed(N * 1000 + Cin + Cout)
    x = bf16r(torch.randn(N, Cin, H, W))
    w = bf16r(torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5)
    b = torch.randn(Cout)
    ref = F.conv2d(x, w, b, stride, k // 2)
    from holocron_amd import _lib
    stats = torch.zeros(_lib.stat_replicas(), 2, Cout, device="cuda")
    out = cv.conv2d(x.cuda(), w.cuda(), b.cuda(), stride, k // 2, stats=stats)
    assert out.shape == ref.shape
    assert rel_l2(_to_nchw_f32(out), ref) < 3e-3
    # the statistics epilogue sees the fp32 accumulators BEFORE bias/rounding
    nob = F.conv2d(x, w, None, stride, k // 2).double()
    s = stats.cpu().double().sum(0)
    assert rel_l2(s[0], nob.sum((0, 2, 3))) < 2e-4 or (s[0] - nob.sum((0, 2, 3))).abs().max() < 1e-2
    assert rel_l2(s[1], (nob * nob).sum((0, 2, 3))) < 2e-4
    relu = cv.conv2d(x.cuda(), w.cuda(), b.cuda(), stride, k // 2, act=1)
    assert rel_l2(_to_nchw_f32(relu), F.relu(ref)) < 3e-3


BIG_TILE_CASES = [
    # N, Cin, Cout, H, W: shapes the efficiency predicate of the big-tile family (csrc/conv_gather.hip: hc_conv_gather) sends to the
    # eight-wave pipelined kernel - ragged last pixel tile, ragged last channel tile, the 128 x 512 tile, and a launch of several rounds
    (90, 64, 480, 19, 19),     # 256 x 256 tiles, 2 channel tiles (the second 224 of 256 rows), 127 pixel tiles with a ragged last one
    (31, 64, 112, 64, 64),     # 128 x 512 tiles, 112 of 128 channel rows, 248 tiles
    (16, 128, 128, 76, 76),    # YOLOv4's 128 @ 76 x 76 at batch 16: 181 tiles of 128 x 512, ragged last tile
    (40, 96, 256, 40, 41),     # 256 x 256, 257 tiles: one full round + one tile (efficiency 0.5: stays on the 128 x 128 form)
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", BIG_TILE_CASES)
def test_conv_forward_big_tile_family_matches_cpu(N, Cin, Cout, H, W):
    """3x3 forward + statistics + bias + ReLU through cv.conv2d at sizes where the launch-efficiency predicate picks the big-tile family
    (and one where it must not): against torch-CPU fp32 on bf16-representable operands, same bounds as the small cases."""
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    torch.manual_seed(Cin + Cout + H)
    x = bf16r(torch.randn(N, Cin, H, W))
    w = bf16r(torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout)
    ref = F.conv2d(x, w, b, 1, 1)
    stats = torch.zeros(_lib.stat_replicas(), 2, Cout, device="cuda")
    out = cv.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1, stats=stats)
    assert rel_l2(_to_nchw_f32(out), ref) < 3e-3
    # element-wise too: a mis-addressed ragged tile would hide in a norm over 10^7 elements
    assert float((_to_nchw_f32(out) - ref).abs().max()) < 0.05 * float(ref.abs().max())
    nob = (ref - b.view(1, -1, 1, 1)).double()
    s = stats.cpu().double().sum(0)
    assert (s[0] - nob.sum((0, 2, 3))).abs().max() < 2e-4 * float(nob.abs().sum((0, 2, 3)).max())
    assert rel_l2(s[1], (nob * nob).sum((0, 2, 3))) < 2e-4
    relu = cv.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1, act=1)
    assert rel_l2(_to_nchw_f32(relu), F.relu(ref)) < 3e-3


def test_conv_forward_stem_im2col():
    from holocron_amd.ops import conv as cv
    torch.manual_seed(3)
    x = bf16r(torch.rand(3, 3, 33, 31))
    w = bf16r(torch.randn(48, 3, 3, 3) * 0.2)
    out = cv.conv2d(x.cuda(), w.cuda(), None, 2, 1)
    assert rel_l2(_to_nchw_f32(out), F.conv2d(x, w, None, 2, 1)) < 3e-3


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride", [(2, 16, 48, 12, 12, 1), (2, 48, 96, 13, 10, 2), (2, 64, 192, 9, 9, 1),
                                                     (1, 192, 128, 8, 8, 2), (2, 32, 64, 7, 7, 2)])
def test_dual_branch_dgrad_matches_autograd(N, Cin, Cout, H, W, stride):
    from holocron_amd.ops import conv as cv
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, requires_grad=True)
    w3 = bf16r(torch.randn(Cout, Cin, 3, 3) / (Cout * 9) ** 0.5)
    w1 = bf16r(torch.randn(Cout, Cin, 1, 1) / Cout ** 0.5)
    y3, y1 = F.conv2d(x, w3, None, stride, 1), F.conv2d(x, w1, None, stride, 0)
    g3, g1 = bf16r(torch.randn_like(y3)), bf16r(torch.randn_like(y1))
    res = bf16r(torch.randn(N, Cin, H, W))
    (dx,) = torch.autograd.grad((y3 * g3).sum() + (y1 * g1).sum(), x)
    d = cv.dgrad_desc(N, Cin, H, W, Cout, [(3, 3, 1, 0, 0), (1, 1, 0, 1, 9)], stride)
    wp = torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device="cuda")
    cv.pack_weight(w3.cuda(), 1, out=wp, tap0=0, T=10)
    cv.pack_weight(w1.cuda(), 1, out=wp, tap0=9, T=10)
    out = cv.empty_cl(N, Cin, H, W, "cuda")
    cv.launch_conv(d, cv.to_cl_bf16(g3.cuda()), wp, out, src1=cv.to_cl_bf16(g1.cuda()), resid=cv.to_cl_bf16(res.cuda()))
    assert rel_l2(_to_nchw_f32(out), dx + res) < 3e-3


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride", [(2, 16, 48, 12, 12, 3, 1), (3, 48, 48, 11, 9, 3, 2), (2, 64, 192, 9, 9, 3, 1),
                                                       (2, 192, 64, 8, 8, 1, 1), (4, 32, 96, 7, 7, 1, 2), (2, 136, 200, 6, 6, 3, 1),
                                                       # 16 input channels: the transposing kernel's 16-channel tile (round 6; ReXNet's 16 -> 96 expansion)
                                                       (2, 16, 96, 20, 20, 1, 1), (3, 16, 32, 9, 9, 1, 2), (1, 16, 160, 10, 10, 3, 1), (2, 16, 144, 130, 7, 1, 1)])
def test_wgrad_matches_autograd(N, Cin, Cout, H, W, k, stride):
    from holocron_amd.ops import conv as cv
    torch.manual_seed(Cin * 3 + Cout)
    x = bf16r(torch.randn(N, Cin, H, W))
    w = torch.randn(Cout, Cin, k, k, requires_grad=True)
    y = F.conv2d(x, w, None, stride, k // 2)
    g = bf16r(torch.randn_like(y))
    (dw,) = torch.autograd.grad((y * g).sum(), w)
    out = cv.conv_wgrad(cv.to_cl_bf16(x.cuda()), cv.to_cl_bf16(g.cuda()), Cin, Cout, k, k, stride, k // 2)
    assert out.shape == dw.shape
    assert rel_l2(out.cpu(), dw) < 2e-4
    acc = cv.conv_wgrad(cv.to_cl_bf16(x.cuda()), cv.to_cl_bf16(g.cuda()), Cin, Cout, k, k, stride, k // 2,
                        out=out.clone(), accumulate=True)
    assert rel_l2(acc.cpu(), 2 * dw) < 2e-4


def test_wgrad_large_reduction_split_k():
    """many output pixels -> several K splits (slab reduce path)"""
    from holocron_amd.ops import conv as cv
    torch.manual_seed(0)
    N, Cin, Cout, H = 8, 48, 48, 56
    x = bf16r(torch.randn(N, Cin, H, H))
    w = torch.randn(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, None, 1, 1)
    g = bf16r(torch.randn_like(y))
    (dw,) = torch.autograd.grad((y * g).sum(), w)
    out = cv.conv_wgrad(cv.to_cl_bf16(x.cuda()), cv.to_cl_bf16(g.cuda()), Cin, Cout, 3, 3, 1, 1)
    assert rel_l2(out.cpu(), dw) < 2e-4


def test_layout_roundtrip():
    from holocron_amd.ops import conv as cv
    x = bf16r(torch.randn(2, 24, 5, 7))
    y = cv.to_cl_bf16(x.cuda())
    assert y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y.float().cpu(), x)


@pytest.mark.parametrize("N,C,Cout,H,W", [(2, 48, 48, 20, 112), (3, 48, 48, 9, 56), (2, 32, 48, 7, 30), (2, 16, 64, 5, 17),
                                           (1, 48, 40, 3, 128)])
def test_conv_small_forward_and_dgrad(N, C, Cout, H, W):
    """persistent LDS-resident-weight kernel for small channel counts: fused 3x3+1x1 forward with
    BN statistics, and the two-source data gradient with residual"""
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    torch.manual_seed(C + Cout + W)
    x = bf16r(torch.randn(N, C, H, W))
    w3 = bf16r(torch.randn(Cout, C, 3, 3) / (C * 9) ** 0.5)
    w1 = bf16r(torch.randn(Cout, C, 1, 1) / C ** 0.5)
    d = cv.conv_small_desc(N, H, W, C, Cout, 0)
    assert d is not None
    y3 = cv.empty_cl(N, Cout, H, W, "cuda")
    y1 = cv.empty_cl(N, Cout, H, W, "cuda")
    stats = torch.zeros(2, _lib.stat_replicas(), 2, Cout, device="cuda")
    cv.launch_conv_small_fwd(d, cv.to_cl_bf16(x.cuda()), cv.pack_weight(w3.cuda(), 0), cv.pack_weight(w1.cuda(), 0), y3, y1,
                             stats[0], stats[1])
    r3, r1 = F.conv2d(x, w3, None, 1, 1), F.conv2d(x, w1, None, 1, 0)
    assert rel_l2(_to_nchw_f32(y3), r3) < 3e-3 and rel_l2(_to_nchw_f32(y1), r1) < 3e-3
    s = stats.cpu().double().sum(1)
    for ref, st in ((r3, s[0]), (r1, s[1])):
        assert (st[0] - ref.double().sum((0, 2, 3))).abs().max() < 2e-2 + 2e-4 * ref.abs().sum((0, 2, 3)).max()
        assert rel_l2(st[1], (ref.double() ** 2).sum((0, 2, 3))) < 2e-4
    # data gradient of conv3x3(u) + conv1x1(u) for u with Cin_fwd = Cout (here: roles swapped)
    Cin_f, Cout_f = Cout, C          # forward conv maps Cin_f -> Cout_f; dgrad consumes dy with Cout_f channels
    if Cin_f % 16 == 0 and Cin_f <= 64:
        u = torch.randn(N, Cin_f, H, W, requires_grad=True)
        v3 = bf16r(torch.randn(Cout_f, Cin_f, 3, 3) / (Cout_f * 9) ** 0.5)
        v1 = bf16r(torch.randn(Cout_f, Cin_f, 1, 1) / Cout_f ** 0.5)
        o3, o1 = F.conv2d(u, v3, None, 1, 1), F.conv2d(u, v1, None, 1, 0)
        g3, g1 = bf16r(torch.randn_like(o3)), bf16r(torch.randn_like(o1))
        res = bf16r(torch.randn(N, Cin_f, H, W))
        (du,) = torch.autograd.grad((o3 * g3).sum() + (o1 * g1).sum(), u)
        wpd = torch.empty((Cin_f, 10, Cout_f), dtype=torch.bfloat16, device="cuda")
        cv.pack_weight(v3.cuda(), 1, out=wpd, tap0=0, T=10)
        cv.pack_weight(v1.cuda(), 1, out=wpd, tap0=9, T=10)
        dd = cv.conv_small_desc(N, H, W, Cout_f, Cin_f, 1)
        assert dd is not None
        dx = cv.empty_cl(N, Cin_f, H, W, "cuda")
        cv.launch_conv_small_dgrad(dd, cv.to_cl_bf16(g3.cuda()), cv.to_cl_bf16(g1.cuda()), wpd, dx, resid=cv.to_cl_bf16(res.cuda()))
        assert rel_l2(_to_nchw_f32(dx), du + res) < 3e-3


@pytest.mark.parametrize("pipe", ["1", "0"])
@pytest.mark.parametrize("grid", ["3", "7", "8", "16"])   # multiples of 8: the XCD-grouped tile order
def test_conv_small_many_tiles_per_workgroup(grid, pipe, monkeypatch):
    """The persistent kernel walks several tiles per workgroup (software-pipelined form: the epilogue of tile i-1 runs inside
    the MFMA stream of tile i).  HC_CONV_SMALL_GRID caps the grid so that small inputs exercise first / steady-state / last
    iterations, ragged last row tiles included; HC_CONV_SMALL_PIPE=0 is the non-pipelined kernel.  Both must match conv2d."""
    monkeypatch.setenv("HC_CONV_SMALL_GRID", grid)
    monkeypatch.setenv("HC_CONV_SMALL_PIPE", pipe)
    for args in [(2, 48, 48, 20, 112), (3, 48, 48, 9, 56), (2, 32, 48, 7, 30), (2, 16, 64, 5, 17), (1, 48, 40, 3, 128)]:
        test_conv_small_forward_and_dgrad(*args)




@pytest.mark.parametrize("N,Cin,Cout,H,k,jobs", [(2, 128, 128, 19, 3, 3), (2, 256, 128, 16, 1, 5), (1, 128, 256, 24, 3, 16), (3, 256, 128, 9, 3, 2)])
def test_grouped_wgrad_matches_single_launches(N, Cin, Cout, H, k, jobs):
    """hc_conv_wgrad_group (same-shaped layers in one launch pair, round 6) against hc_conv_wgrad layer by layer and against fp32
    autograd on the same bf16 operands; `beta` accumulates into the destination."""
    import ctypes as C
    import torch.nn.functional as F
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    lib = _lib.load()
    g = torch.Generator().manual_seed(Cin + Cout + H + k)
    xs = [bf16r(torch.randn((N, Cin, H, H), generator=g)) for _ in range(jobs)]
    dys = [bf16r(torch.randn((N, Cout, H, H), generator=g)) for _ in range(jobs)]
    key = (N, Cin, H, H, Cout, k, k, 1, k // 2)
    assert cv._WCONV.supported(key)
    xg = [cv.to_cl_bf16(t.cuda()) for t in xs]
    dg = [cv.to_cl_bf16(t.cuda()) for t in dys]
    outs = [torch.full((Cout, Cin, k, k), 0.5, dtype=torch.float32, device="cuda") for _ in range(jobs)]
    cv._WCONV.launch(key, [(a, b, None, o.data_ptr(), 0) for a, b, o in zip(xg, dg, outs)], accumulate=True)
    torch.cuda.synchronize()
    for j in range(jobs):
        single = cv.conv_wgrad(xg[j], dg[j], Cin, Cout, k, k, 1, k // 2)
        w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
        (ref,) = torch.autograd.grad((F.conv2d(xs[j], w, None, 1, k // 2) * dys[j]).sum(), w)
        got = outs[j].cpu() - 0.5
        assert rel_l2(got, ref) < 2e-4, (j, rel_l2(got, ref))
        assert rel_l2(got, single.cpu()) < 2e-5, (j, rel_l2(got, single.cpu()))
    # the same group twice is the same bits (fixed-order slab sums)
    outs2 = [torch.full((Cout, Cin, k, k), 0.5, dtype=torch.float32, device="cuda") for _ in range(jobs)]
    cv._WCONV.launch(key, [(a, b, None, o.data_ptr(), 0) for a, b, o in zip(xg, dg, outs2)], accumulate=True)
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2))


@pytest.mark.parametrize("shape", [(2, 3, 30, 30, 3, 1, 1, 32), (2, 3, 31, 29, 3, 2, 1, 32), (3, 3, 64, 48, 3, 2, 1, 32), (1, 3, 17, 608, 3, 1, 1, 32),
                                   (2, 3, 32, 32, 3, 1, 0, 32), (2, 3, 28, 28, 7, 2, 3, 152), (2, 4, 20, 24, 3, 1, 1, 40)])
def test_stem_im2col_matches_unfold(shape):
    """hc_im2col_small (the stems' column tensor: models/utils.py:73 with Cin = 3) against F.unfold, bit for bit: the LDS-staged form
    of the 3-channel 3 x 3 pad-1 stem (widths that are and are not multiples of four, odd heights, strides 1 / 2) and the gather
    form of everything else (no padding, 7 x 7, four channels).  k = (kh * KW + kw) * Cin + ci, zero beyond Cin * KH * KW."""
    import torch.nn.functional as F
    from holocron_amd.ops import conv as cv
    N, Cin, H, W, k, stride, pad, Kpad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g)
    col = cv.im2col_small(x.cuda(), k, k, stride, pad, Kpad)                       # logical [N, Kpad, OH, OW], NHWC bf16
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    assert tuple(col.shape) == (N, Kpad, OH, OW)
    u = F.unfold(x, k, padding=pad, stride=stride).view(N, Cin, k * k, OH, OW)     # [N, ci, tap, OH, OW]
    want = torch.zeros((N, Kpad, OH, OW))
    want[:, :Cin * k * k] = u.permute(0, 2, 1, 3, 4).reshape(N, k * k * Cin, OH, OW)
    assert torch.equal(col.float().cpu(), want.bfloat16().float()), shape
