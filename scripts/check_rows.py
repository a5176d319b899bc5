"""Row-unit conv (conv_rows.hip) against fp32 torch convolutions of the same bf16 inputs + timing.  BL_N / BL_SHAPES="C,H;..." """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from holocron_amd.ops import conv as cv
from holocron_amd import _lib

N = int(os.environ.get("BL_N", "256"))
shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ.get("BL_SHAPES", "192,14;96,28").split(";")]


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


R = _lib.stat_replicas()
bf = lambda t: t.to(torch.bfloat16).float()
for Cc, H in shapes:
    torch.manual_seed(0)
    x = bf(torch.randn(N, Cc, H, H, device="cuda"))
    dy3 = bf(torch.randn(N, Cc, H, H, device="cuda")); dy1 = bf(torch.randn(N, Cc, H, H, device="cuda"))
    w3 = bf(torch.randn(Cc, Cc, 3, 3, device="cuda") * 0.05); w1 = bf(torch.randn(Cc, Cc, 1, 1, device="cuda") * 0.1)
    xc, d3c, d1c = cv.to_cl_bf16(x), cv.to_cl_bf16(dy3), cv.to_cl_bf16(dy1)
    rows = os.environ.get("HC_CONV_ROWS", "1") != "0"
    if rows:
        wp3 = cv.rows_image(Cc, "cuda"); wp1 = None
        cv.pack_weight(w3, 3, out=wp3, tap0=0, T=10); cv.pack_weight(w1, 3, out=wp3, tap0=9, T=10)
    else:
        wp3, wp1 = cv.pack_weight(w3, 0), cv.pack_weight(w1, 0)
    d = cv.conv_small_desc(N, H, H, Cc, Cc, cv.ROWS_IMAGE if rows else 0)
    assert d is not None, (Cc, H)
    y3 = cv.empty_cl(N, Cc, H, H, "cuda"); y1 = cv.empty_cl(N, Cc, H, H, "cuda")
    stats = torch.zeros(2, R, 2, Cc, device="cuda")
    cv.launch_conv_small_fwd(d, xc, wp3, wp1, y3, y1, stats[0], stats[1])
    torch.cuda.synchronize()
    r3, r1 = F.conv2d(x, w3, padding=1), F.conv2d(x, w1)
    e3, e1 = rel(y3.float(), r3), rel(y1.float(), r1)
    s3 = stats[0].sum(0); s1 = stats[1].sum(0)
    es = max(rel(s3[0], r3.sum((0, 2, 3))), rel(s3[1], (r3 * r3).sum((0, 2, 3))), rel(s1[0], r1.sum((0, 2, 3))), rel(s1[1], (r1 * r1).sum((0, 2, 3))))
    if rows:
        wpd = cv.rows_image(Cc, "cuda")
        cv.pack_weight(w3, 4, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 4, out=wpd, tap0=9, T=10)
    else:
        wpd = torch.empty((Cc, 10, Cc), dtype=torch.bfloat16, device="cuda")
        cv.pack_weight(w3, 1, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 1, out=wpd, tap0=9, T=10)
    dd = cv.conv_small_desc(N, H, H, Cc, Cc, (cv.ROWS_IMAGE | 1) if rows else 1)
    if dd is None and rows:     # 48 channels: the row-unit form is forward only, the data gradient is conv_small's persistent kernel
        wpd = torch.empty((Cc, 10, Cc), dtype=torch.bfloat16, device="cuda")
        cv.pack_weight(w3, 1, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 1, out=wpd, tap0=9, T=10)
        dd = cv.conv_small_desc(N, H, H, Cc, Cc, 1)
    dx = cv.empty_cl(N, Cc, H, H, "cuda")
    cv.launch_conv_small_dgrad(dd, d3c, d1c, wpd, dx, resid=xc)
    torch.cuda.synchronize()
    rdx = F.conv_transpose2d(dy3, w3, padding=1) + F.conv_transpose2d(dy1, w1) + x
    ed = rel(dx.float(), rdx)
    flops = 2.0 * N * H * H * Cc * Cc * 10
    tf = timeit(lambda: cv.launch_conv_small_fwd(d, xc, wp3, wp1, y3, y1, stats[0], stats[1]))
    tn = timeit(lambda: cv.launch_conv_small_fwd(d, xc, wp3, wp1, y3, y1, None, None))
    td = timeit(lambda: cv.launch_conv_small_dgrad(dd, d3c, d1c, wpd, dx, resid=xc))
    print(f"{Cc}@{H} N={N}: y3 {e3:.2e} y1 {e1:.2e} stats {es:.2e} dx {ed:.2e} | fwd+stats {tf:6.1f} us ({flops / tf / 1e6:6.0f} TF) fwd {tn:6.1f} us "
          f"dgrad {td:6.1f} us ({flops / td / 1e6:6.0f} TF)", flush=True)
