"""Forward of the three front stride-2 RepBlock conv pairs at batch 256 (stem, 48@112 -> 48@56, 48@56 -> 96@28): time per launch
sequence and algorithmic GB/s.  HC_CONV_S2=0 routes them to the gather-conv path (im2col + stacked gather), HC_CONV_S2_R=1 picks the
smaller rows-per-workgroup variant of the stride-2 row kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd import _lib
from holocron_amd.nn import repblock_op as rb
from holocron_amd.ops import conv as cv


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N = 256
dev = torch.device("cuda:0")
for (cin, cout, H) in [(3, 48, 224), (48, 48, 112), (48, 96, 56)]:
    x = torch.rand((N, cin, H, H), device=dev)
    w3 = torch.randn((cout, cin, 3, 3), device=dev) * 0.1
    w1 = torch.randn((cout, cin, 1, 1), device=dev) * 0.1
    st = rb.RepState(2, False)
    geom = (N, cin, H, H, cout)
    direct = st.s2_desc(*geom) is not None
    stem = cin == 3
    src = x if (stem and direct) else (cv.im2col_small(x, 3, 3, 2, 1, rb.STEM_KPAD) if stem else cv.to_cl_bf16(x))
    stats = torch.zeros((2, _lib.stat_replicas(), 2, cout), device=dev)

    def run():
        s = src
        if stem and not direct:
            s = cv.im2col_small(x, 3, 3, 2, 1, rb.STEM_KPAD)
        rb.block_convs_forward(st, s, w3, w1, geom, stats, cin if stem else None)
    us = timeit(run)
    nbytes = x.numel() * (4 if stem else 2) + 2 * N * (H // 2) ** 2 * cout * 2
    print(f"{cin}->{cout}@{H}: {'conv_s2' if direct else 'gather '} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.0f} MB)")

for (cin, cout, H) in [(48, 48, 112), (48, 96, 56)]:
    dy3 = cv.to_cl_bf16(torch.randn((N, cout, H // 2, H // 2), device=dev))
    dy1 = cv.to_cl_bf16(torch.randn((N, cout, H // 2, H // 2), device=dev))
    w3 = torch.randn((cout, cin, 3, 3), device=dev) * 0.1
    w1 = torch.randn((cout, cin, 1, 1), device=dev) * 0.1
    st = rb.RepState(2, False)
    geom = (N, cin, H, H, cout)
    st.descs(*geom)
    us = timeit(lambda: rb.block_dgrad(st, dy3, dy1, None, w3, w1, geom))
    nbytes = N * H * H * cin * 2 + 2 * dy3.numel() * 2
    print(f"dgrad {cin}<-{cout}@{H}: {'conv_s2' if st.s2_dgrad else 'gather '} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.0f} MB)")

x = torch.rand((N, 3, 224, 224), device=dev)
dy3 = cv.to_cl_bf16(torch.randn((N, 48, 112, 112), device=dev))
dy1 = cv.to_cl_bf16(torch.randn((N, 48, 112, 112), device=dev))
w3 = torch.randn((48, 3, 3, 3), device=dev)
w1 = torch.randn((48, 3, 1, 1), device=dev)
st = rb.RepState(2, False)
geom = (N, 3, 224, 224, 48)
direct = st.s2_desc(*geom) is not None
col = None if direct else cv.im2col_small(x, 3, 3, 2, 1, rb.STEM_KPAD)
us = timeit(lambda: rb.block_wgrad(st, x if direct else col, dy3, dy1, w3, w1, geom, 3))
nbytes = x.numel() * 4 + 2 * dy3.numel() * 2
print(f"stem wgrad: {'direct' if direct else 'im2col GEMM (column tensor not counted)'} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.0f} MB)")
