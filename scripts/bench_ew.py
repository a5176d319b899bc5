"""Micro-benchmark of the elementwise BN-fusion kernels on RepVGG-A0 bs256 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from holocron_amd import _lib
from holocron_amd.ops import conv as cv
lib = _lib.load()
S = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (Cc, H) in [(48, 112), (48, 56), (96, 28), (192, 14), (1280, 7)]:
    N = 256
    t = [cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda")) for _ in range(8)]
    npix = N * H * H
    mb = npix * Cc * 2 / 1e6
    red = torch.zeros(128, 4, Cc, device="cuda")
    coef = torch.rand(4, Cc, device="cuda"); bc = torch.rand(9, Cc, device="cuda")
    st = torch.zeros(128, 2, Cc, device="cuda")
    p = lambda x: x.data_ptr()
    us = timeit(lambda: lib.hc_rep_bwd_reduce(p(t[0]), p(t[1]), p(t[2]), p(t[3]), p(t[4]), p(red), npix, Cc, S()))
    print(f"C={Cc:4d} H={H:3d} tensor {mb:6.1f} MB | bwd_reduce {us:7.1f} us ({5*mb/us:5.2f} TB/s)", end="")
    us = timeit(lambda: lib.hc_rep_bwd_apply(p(t[0]), p(t[1]), p(t[2]), p(t[3]), p(t[4]), p(bc), p(t[5]), p(t[6]), p(t[7]), npix, Cc, S()))
    print(f" | bwd_apply {us:7.1f} us ({8*mb/us:5.2f} TB/s)", end="")
    us = timeit(lambda: lib.hc_rep_bwd_reduce_z(p(t[0]), p(coef), 1, p(t[2]), p(t[3]), p(t[4]), p(red), npix, Cc, S()))
    print(f" | reduce_z {us:7.1f} us ({4*mb/us:5.2f} TB/s)", end="")
    us = timeit(lambda: lib.hc_rep_bwd_apply_z(p(t[0]), p(coef), 1, p(t[2]), p(t[3]), p(t[4]), p(bc), p(t[5]), p(t[6]), p(t[7]), npix, Cc, S()))
    print(f" | apply_z {us:7.1f} us ({7*mb/us:5.2f} TB/s)", end="")
    us = timeit(lambda: lib.hc_rep_apply(p(t[0]), p(t[1]), p(t[2]), p(coef), p(t[5]), p(st), npix, Cc, 1, S()))
    print(f" | apply+stats {us:7.1f} us ({4*mb/us:5.2f} TB/s)", end="")
    us = timeit(lambda: lib.hc_rep_apply(p(t[0]), p(t[1]), p(t[2]), p(coef), p(t[5]), None, npix, Cc, 1, S()))
    print(f" | apply {us:7.1f} us ({4*mb/us:5.2f} TB/s)")
