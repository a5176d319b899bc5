"""The stem block fused with its BatchNorm passes (csrc/conv_s2.hip stem_fused_kernel) at batch 256: time and algorithmic TB/s of the
three launches (statistics, apply, one-pass backward) next to the unfused sequence they replace (conv -> y3 / y1, apply,
reduce, backward apply, stem weight gradient).  HC_STEM_GRID overrides the persistent grid."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd import _lib
from holocron_amd.nn import repblock_op as rb


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N = int(os.environ.get("N", "256"))
dev = torch.device("cuda:0")
lib = _lib.load()
x = torch.rand((N, 3, 224, 224), device=dev)
w3 = torch.randn((48, 3, 3, 3), device=dev) * 0.3
w1 = torch.randn((48, 3, 1, 1), device=dev) * 0.5
st = rb.RepState(2, False)
d = rb.stem_fused_desc(st, x, w3, w1, (N, 3, 224, 224, 48))
R = _lib.stat_replicas()
stats = torch.zeros((2, R, 2, 48), device=dev)
ostats = torch.zeros((R, 2, 48), device=dev)
coef = torch.rand((4, 48), device=dev) - 0.3
save = torch.rand((6, 48), device=dev) + 0.5
gam = torch.rand((2, 48), device=dev) + 0.5
out = torch.empty((N, 112, 112, 48), dtype=torch.bfloat16, device=dev)
g = torch.rand((N, 112, 112, 48), device=dev).to(torch.bfloat16)
dw3, dw1 = torch.empty((48, 3, 3, 3), device=dev), torch.empty((48, 3, 1, 1), device=dev)
dgb = torch.empty((4, 48), device=dev)
ws = torch.empty((int(lib.hc_stem_bwd_ws_bytes()),), dtype=torch.uint8, device=dev)
b = _lib.StemBwdDesc()
b.coef, b.g, b.save, b.gamma3, b.gamma1, b.w3, b.w1 = (t.data_ptr() for t in (coef, g, save, gam[0], gam[1], w3, w1))
b.dgamma3, b.dbeta3, b.dgamma1, b.dbeta1 = (dgb[i].data_ptr() for i in range(4))
b.dw3, b.dw1, b.ws, b.act, b.frozen, b.accumulate = dw3.data_ptr(), dw1.data_ptr(), ws.data_ptr(), 1, 0, 0
s = torch.cuda.current_stream().cuda_stream
img, T = x.numel() * 4, out.numel() * 2
cases = [
    ("stats    (image)", img, lambda: lib.hc_stem_stats(C.byref(d), stats[0].data_ptr(), stats[1].data_ptr(), s)),
    ("apply    (image -> out, + out stats)", img + T, lambda: lib.hc_stem_apply(C.byref(d), coef.data_ptr(), 1, out.data_ptr(), ostats.data_ptr(), s)),
    ("backward (image, g -> all gradients)", img + T, lambda: lib.hc_stem_bwd(C.byref(d), C.byref(b), s)),
]
tot = 0.0
for name, nbytes, fn in cases:
    assert fn() == 0
    us = timeit(fn)
    tot += us
    print(f"fused {name:<40} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.0f} MB)")
print(f"fused total {tot:.1f} us")
