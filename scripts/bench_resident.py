"""Micro-benchmark of the image-resident conv (192 channels @ 14x14, batch 256): forward 3x3+1x1 with statistics, data gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd.ops import conv as cv
from holocron_amd import _lib

N = int(os.environ.get("BL_N", "256"))
Cc, H = int(os.environ.get("BL_C", "192")), int(os.environ.get("BL_H", "14"))


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


R = _lib.stat_replicas()
x = cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda"))
dy3 = cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda")); dy1 = cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda"))
w3 = torch.randn(Cc, Cc, 3, 3, device="cuda"); w1 = torch.randn(Cc, Cc, 1, 1, device="cuda")
wp3, wp1 = cv.pack_weight(w3, 0), cv.pack_weight(w1, 0)
d = cv.conv_small_desc(N, H, H, Cc, Cc, 0)
y3 = cv.empty_cl(N, Cc, H, H, "cuda"); y1 = cv.empty_cl(N, Cc, H, H, "cuda")
stats = torch.zeros(2, R, 2, Cc, device="cuda")
flops = 2.0 * N * H * H * Cc * Cc * 10
us = timeit(lambda: cv.launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, stats[0], stats[1]))
print(f"resident fwd 3x3+1x1 +stats {Cc}@{H}: {us:8.1f} us {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
us = timeit(lambda: cv.launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, None, None))
print(f"resident fwd 3x3+1x1        {Cc}@{H}: {us:8.1f} us {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
wpd = torch.empty((Cc, 10, Cc), dtype=torch.bfloat16, device="cuda")
cv.pack_weight(w3, 1, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 1, out=wpd, tap0=9, T=10)
dd = cv.conv_small_desc(N, H, H, Cc, Cc, 1)
dx = cv.empty_cl(N, Cc, H, H, "cuda")
us = timeit(lambda: cv.launch_conv_small_dgrad(dd, dy3, dy1, wpd, dx, resid=x))
print(f"resident dgrad (+resid)     {Cc}@{H}: {us:8.1f} us {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
