"""Micro-benchmark of the single-branch conv -> BN -> activation passes (hc_bn_act_apply / _bwd_reduce / _bwd_apply, csrc/rep_bn.hip) at
the ReXNet batch-256 and YOLOv4 608^2 batch-16 tensor shapes: time and algorithmic TB/s (2 r + 1 w / 2 r / 2 r + 1 w)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd import _lib
from holocron_amd.ops import conv as cv
lib = _lib.load()
S = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


p = lambda t: t.data_ptr()
tot = [0.0, 0.0, 0.0]
for (N, Cc, H, act) in [(256, 96, 112, 5), (256, 160, 56, 5), (256, 240, 28, 5), (256, 448, 14, 5), (256, 1120, 7, 5),
                        (16, 64, 304, 4), (16, 128, 152, 4), (16, 256, 76, 4), (16, 512, 38, 4), (16, 1024, 19, 4)]:
    y, g, out, dy = (cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda")) for _ in range(4))
    npix = N * H * H
    mb = npix * Cc * 2 / 1e6
    R = _lib.stat_replicas()
    red = torch.zeros(R, 4, Cc, device="cuda")
    coef = torch.rand(4, Cc, device="cuda")
    bc = torch.rand(9, Cc, device="cuda")
    ua = timeit(lambda: lib.hc_bn_act_apply(p(y), p(coef), None, 0, None, None, p(out), Cc, npix, Cc, act, 0.1, S()))
    ur = timeit(lambda: lib.hc_bn_act_bwd_reduce(p(g), Cc, p(y), p(coef), None, None, p(red), npix, Cc, act, 0.1, S()))
    ub = timeit(lambda: lib.hc_bn_act_bwd_apply(p(g), Cc, p(y), p(coef), p(bc), None, None, p(dy), npix, Cc, act, 0.1, S()))
    tot[0] += ua; tot[1] += ur; tot[2] += ub
    print(f"N={N:3d} C={Cc:4d} H={H:3d} act={act} tensor {mb:6.1f} MB | apply {ua:7.1f} us ({2 * mb / ua:5.2f} TB/s) | bwd_reduce {ur:7.1f} us "
          f"({2 * mb / ur:5.2f} TB/s) | bwd_apply {ub:7.1f} us ({3 * mb / ub:5.2f} TB/s)")
print(f"sum: apply {tot[0]:.0f} us, bwd_reduce {tot[1]:.0f} us, bwd_apply {tot[2]:.0f} us")
