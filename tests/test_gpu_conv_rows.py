"""Row-unit convolution kernel (csrc/conv_rows.hip: the 192 @ 14x14 and 96 @ 28x28 RepVGG-A0 stages) and its weight image
(hc_pack_conv_weight modes 3 / 4) against fp32 torch convolutions of the same bf16 operands.

Reference semantics: RepBlock.forward (holocron/models/classification/repvgg.py:71-73: conv3x3 + conv1x1 of one input) and
aten::convolution_backward's data gradient of both.  Bounds: one bf16 rounding of an fp32 result is 1.65e-3 rel-L2 on these
distributions (tests/test_gpu_fullsize_layers.py); the statistics are fp32 sums of fp32 accumulators (1e-5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-3


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def rows_index(rc, kc, tap, C):
    """The layout hc_pack_conv_weight modes 3 / 4 document (rows_image_index, csrc/rep_bn.hip), restated."""
    w, c = divmod(rc, 48)
    f, gq, i = ((c >> 2) & 1, c >> 3, c & 3) if c < 32 else (2, (c - 32) >> 2, c & 3)
    r = 48 * w + 16 * f + 4 * gq + i
    t, kk = divmod(kc, 32)
    hi, g, e = kk >> 4, (kk & 15) >> 2, kk & 3
    j = 8 * g + 4 * hi + e
    return ((tap * ((C + 31) // 32) + t) * C + r) * 32 + j


@pytest.mark.parametrize("C", [48, 96, 192])
def test_rows_image_layout(C):
    from holocron_amd.ops import conv as cv
    g = torch.Generator(device="cuda").manual_seed(0)
    w3 = torch.randn(C, C, 3, 3, device="cuda", generator=g)
    w1 = torch.randn(C, C, 1, 1, device="cuda", generator=g)
    for mode in (3, 4):
        img = cv.rows_image(C, "cuda")
        img.zero_()
        cv.pack_weight(w3, mode, out=img, tap0=0, T=10)
        cv.pack_weight(w1, mode, out=img, tap0=9, T=10)
        got = img.float().cpu().numpy().ravel()
        w3b, w1b = w3.bfloat16().float().cpu().numpy(), w1.bfloat16().float().cpu().numpy()
        rs = np.random.RandomState(1)
        for _ in range(400):
            co, ci, kh, kw = rs.randint(C), rs.randint(C), rs.randint(3), rs.randint(3)
            if mode == 3:
                assert got[rows_index(co, ci, kh * 3 + kw, C)] == w3b[co, ci, kh, kw]
                assert got[rows_index(co, ci, 9, C)] == w1b[co, ci, 0, 0]
            else:   # data gradient: rows = input channels, k = output channels, taps flipped
                assert got[rows_index(ci, co, (2 - kh) * 3 + (2 - kw), C)] == w3b[co, ci, kh, kw]
                assert got[rows_index(ci, co, 9, C)] == w1b[co, ci, 0, 0]


@pytest.mark.parametrize("C", [48, 96, 192, 240])
def test_rows_image_multi_pack_is_the_single_pack(C):
    """hc_pack_conv_weights_multi builds the row-unit images through LDS tiles (48 row channels x one k32 block x all taps per unit,
    16-byte loads and stores); hc_pack_conv_weight walks the elements.  Same bits, including the zero half of the last k block when
    C is not a multiple of 32."""
    from holocron_amd.ops import conv as cv
    from holocron_amd.nn.repblock_op import launch_pack_items
    g = torch.Generator(device="cuda").manual_seed(C)
    w3 = torch.randn(C, C, 3, 3, device="cuda", generator=g)
    w1 = torch.randn(C, C, 1, 1, device="cuda", generator=g)
    for mode in (3, 4):
        ref = cv.rows_image(C, "cuda"); ref.zero_()
        cv.pack_weight(w3, mode, out=ref, tap0=0, T=10)
        cv.pack_weight(w1, mode, out=ref, tap0=9, T=10)
        got = cv.rows_image(C, "cuda"); got.zero_()
        launch_pack_items([(w3, got, C, C, 3, 3, mode, 0, 10), (w1, got, C, C, 1, 1, mode, 9, 10)])
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (C, mode)
        got.fill_(1.0)            # the tiles own the padding of their pieces: stale values there must not survive a repack
        launch_pack_items([(w3, got, C, C, 3, 3, mode, 0, 10), (w1, got, C, C, 1, 1, mode, 9, 10)])
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (C, mode)


@pytest.mark.parametrize("N,C,H,W", [(2, 192, 14, None), (256, 192, 14, None), (3, 96, 28, None), (256, 96, 28, None), (5, 192, 28, None),
                                     (2, 96, 56, None), (3, 48, 112, None), (64, 48, 56, None),
                                     # the family form (round 4): any width up to 16 (192 channels) / 32 (96 channels) pixels, any height
                                     # - odd unit counts, a ragged last unit, maps narrower than a fragment
                                     (3, 192, 12, 12), (2, 192, 16, 16), (5, 192, 7, 7), (2, 192, 21, 14), (2, 192, 9, 13), (3, 192, 1, 3),
                                     (3, 96, 24, 24), (2, 96, 32, 32), (2, 96, 14, 14), (2, 96, 30, 17), (4, 96, 7, 28), (64, 192, 16, 16)])
def test_conv_rows_vs_fp32(N, C, H, W):
    """forward (3x3 + 1x1 + statistics) and data gradient (+ residual); W None: the tuned template size (H any multiple of 14)"""
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    if W is None:
        W = {192: 14, 96: 28, 48: H}[C]
    bf = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = bf(torch.randn(N, C, H, W, device="cuda", generator=g))
    dy3 = bf(torch.randn(N, C, H, W, device="cuda", generator=g))
    dy1 = bf(torch.randn(N, C, H, W, device="cuda", generator=g))
    w3 = bf(torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05)
    w1 = bf(torch.randn(C, C, 1, 1, device="cuda", generator=g) * 0.1)
    xc, d3c, d1c = cv.to_cl_bf16(x), cv.to_cl_bf16(dy3), cv.to_cl_bf16(dy1)
    wf, wd = cv.rows_image(C, "cuda"), cv.rows_image(C, "cuda")
    cv.pack_weight(w3, 3, out=wf, tap0=0, T=10); cv.pack_weight(w1, 3, out=wf, tap0=9, T=10)
    cv.pack_weight(w3, 4, out=wd, tap0=0, T=10); cv.pack_weight(w1, 4, out=wd, tap0=9, T=10)
    d = cv.conv_small_desc(N, H, W, C, C, cv.ROWS_IMAGE)
    dd = cv.conv_small_desc(N, H, W, C, C, cv.ROWS_IMAGE | 1)
    assert d is not None
    if C == 48 and dd is None:      # 48 channels: the streaming kernel takes the forward only by default (HC_CONV_ROWS48=2: see the
        dd = None                   # subprocess test below); the data gradient of these shapes is covered by test_gpu_conv.py
    else:
        assert dd is not None
    R = _lib.stat_replicas()
    y3, y1 = cv.empty_cl(N, C, H, W, "cuda"), cv.empty_cl(N, C, H, W, "cuda")
    stats = torch.zeros(2, R, 2, C, device="cuda")
    cv.launch_conv_small_fwd(d, xc, wf, None, y3, y1, stats[0], stats[1])
    dx = dx0 = None
    if dd is not None:
        dx = cv.empty_cl(N, C, H, W, "cuda")
        cv.launch_conv_small_dgrad(dd, d3c, d1c, wd, dx, resid=xc)
        dx0 = cv.empty_cl(N, C, H, W, "cuda")
        cv.launch_conv_small_dgrad(dd, d3c, d1c, wd, dx0, resid=None)
    torch.cuda.synchronize()
    r3, r1 = F.conv2d(x, w3, padding=1), F.conv2d(x, w1)
    assert rel(y3, r3) < TOL and rel(y1, r1) < TOL
    s3, s1 = stats[0].sum(0), stats[1].sum(0)
    for got, ref in ((s3[0], r3.sum((0, 2, 3))), (s3[1], (r3 * r3).sum((0, 2, 3))), (s1[0], r1.sum((0, 2, 3))), (s1[1], (r1 * r1).sum((0, 2, 3)))):
        assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5
    if dd is not None:
        rdx = F.conv_transpose2d(dy3, w3, padding=1) + F.conv_transpose2d(dy1, w1)
        assert rel(dx0, rdx) < TOL
        assert rel(dx, rdx + x) < TOL


def test_streaming_48_channel_kernel_with_its_data_gradient():
    """HC_CONV_ROWS48=2 (read once per process, hence a child process): forward, statistics and data gradient of the streaming
    48-channel kernel against fp32 torch convolutions (scripts/check_rows.py prints the rel-L2 errors)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HC_CONV_ROWS48="2", BL_N="5", BL_SHAPES="48,112;48,56")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_rows.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("48@")]
    assert len(lines) == 2, out.stdout
    for l in lines:
        m = re.search(r"y3 (\S+) y1 (\S+) stats (\S+) dx (\S+)", l)
        e3, e1, es, ed = (float(v) for v in m.groups())
        assert e3 < TOL and e1 < TOL and ed < TOL and es < 1e-5, l


def test_conv_rows_unsupported_shapes():
    from holocron_amd.ops import conv as cv
    for (N, H, W, C) in [(4, 28, 28, 192), (4, 14, 17, 192), (4, 14, 14, 128), (4, 14, 40, 96), (4, 14, 14, 64)]:
        assert cv.conv_small_desc(N, H, W, C, C, cv.ROWS_IMAGE) is None
    # ... and what the dispatch predicate takes since round 4 (it was a list of two shapes)
    for (N, H, W, C) in [(4, 14, 14, 96), (4, 21, 14, 192), (4, 7, 14, 192), (4, 16, 16, 192), (4, 32, 32, 96), (4, 5, 9, 96)]:
        assert cv.conv_small_desc(N, H, W, C, C, cv.ROWS_IMAGE) is not None and cv.conv_small_desc(N, H, W, C, C, cv.ROWS_IMAGE | 1) is not None
