"""Stem im2col (hc_im2col_small / _fp8) on the stem shapes of the benchmarked models: time per launch and a checksum of the column
tensor, for same-box A/Bs of two library builds (HC_LIB_PATH=... python scripts/bench_im2col.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd import _lib
from holocron_amd._lib import check, ptr, stream

dev = torch.device("cuda:0")
lib = _lib.load()
CASES = [("yolov4 16x3x608^2 k3 s1", 16, 608, 3, 1, 1, 32, False), ("rexnet 256x3x224^2 k3 s2", 256, 224, 3, 2, 1, 32, False),
         ("repvgg_a2 fp8 1024x3x224^2 k3 s2", 1024, 224, 3, 2, 1, 64, True), ("yolov1 4x3x448^2 k7 s2", 4, 448, 7, 2, 3, 152, False)]
for name, N, HW, k, s, pad, Kpad, fp8 in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, 3, HW, HW), device=dev, generator=g)
    OH = (HW + 2 * pad - k) // s + 1
    col = torch.empty((N, OH, OH, Kpad), dtype=torch.uint8 if fp8 else torch.bfloat16, device=dev)

    def run():
        if fp8:
            check(lib.hc_im2col_small_fp8(ptr(x), ptr(col), N, 3, HW, HW, OH, OH, k, k, s, pad, Kpad, 0.5, stream()), "im2col fp8")
        else:
            check(lib.hc_im2col_small(ptr(x), ptr(col), N, 3, HW, HW, OH, OH, k, k, s, pad, Kpad, stream()), "im2col")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = x.numel() * 4 + col.numel() * col.element_size()
    words = col.view(torch.uint8).view(-1).to(torch.int64)
    chk = int((words * (torch.arange(words.numel(), device=dev) % 251 + 1)).sum().item())
    print(f"{name:36s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  checksum {chk}")
