"""Micro-benchmark of the conv kernels on the RepVGG-A0 bs256 layer shapes (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from holocron_amd.ops import conv as cv

LAYERS = [  # Cin, Cout, H(in), stride
    (48, 48, 112, 1), (48, 48, 112, 2), (48, 48, 56, 1), (48, 96, 56, 2), (96, 96, 28, 1), (96, 192, 28, 2),
    (192, 192, 14, 1), (192, 1280, 14, 2), (1280, 1280, 7, 1),
]
which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
N = int(os.environ.get("BL_N", "256"))
if os.environ.get("BL_SHAPES"):   # "Cin,Cout,H,stride;..."
    LAYERS = [tuple(int(v) for v in t.split(",")) for t in os.environ["BL_SHAPES"].split(";")]
sel = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else range(len(LAYERS))


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for li in sel:
    Cin, Cout, H, s = LAYERS[li]
    OH = (H + 2 - 3) // s + 1
    x = cv.to_cl_bf16(torch.randn(N, Cin, H, H, device="cuda"))
    dy = cv.to_cl_bf16(torch.randn(N, Cout, OH, OH, device="cuda"))
    for k in (3, 1):
        flops = 2.0 * N * OH * OH * Cout * Cin * k * k
        if which == "wgrad":
            us = timeit(lambda: cv.conv_wgrad(x, dy, Cin, Cout, k, k, s, k // 2))
        elif which == "fwd":
            w = torch.randn(Cout, Cin, k, k, device="cuda")
            wp = cv.pack_weight(w, 0)
            d = cv.fwd_desc(N, Cin, H, H, Cout, k, k, s, k // 2)
            out = cv.empty_cl(N, Cout, OH, OH, "cuda")
            stats = torch.zeros(128, 2, Cout, device="cuda")
            us = timeit(lambda: cv.launch_conv(d, x, wp, out, stats=stats))
        elif which == "small":
            if k == 1 or s != 1 or Cin > 48:
                continue
            w3 = torch.randn(Cout, Cin, 3, 3, device="cuda"); w1 = torch.randn(Cout, Cin, 1, 1, device="cuda")
            wp3, wp1 = cv.pack_weight(w3, 0), cv.pack_weight(w1, 0)
            d = cv.conv_small_desc(N, H, H, Cin, Cout, 0)
            y3 = cv.empty_cl(N, Cout, OH, OH, "cuda"); y1 = cv.empty_cl(N, Cout, OH, OH, "cuda")
            stats = torch.zeros(2, 128, 2, Cout, device="cuda")
            flops = 2.0 * N * OH * OH * Cout * Cin * 10
            us = timeit(lambda: cv.launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, stats[0], stats[1]))
            print(f"small-fwd(3x3+1x1) {Cin}->{Cout} @{H}: {us:9.1f} us {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
            us = timeit(lambda: cv.launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, None, None))
            print(f"small-fwd no stats   {Cin}->{Cout} @{H}: {us:9.1f} us", flush=True)
            d.mode = 2
            us = timeit(lambda: cv.launch_conv_small_fwd(d, x, wp3, wp1, y3, y1, None, None))
            print(f"small-fwd no stats no stores {Cin}->{Cout} @{H}: {us:9.1f} us", flush=True)
            d.mode = 0
            wpd = torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device="cuda")
            cv.pack_weight(w3, 1, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 1, out=wpd, tap0=9, T=10)
            dd = cv.conv_small_desc(N, H, H, Cout, Cin, 1)
            dx = cv.empty_cl(N, Cin, H, H, "cuda")
            us = timeit(lambda: cv.launch_conv_small_dgrad(dd, dy, dy, wpd, dx, resid=x))
            print(f"small-dgrad          {Cin}->{Cout} @{H}: {us:9.1f} us {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
            continue
        else:  # dgrad (dual) measured once per layer
            if k == 1:
                continue
            w3 = torch.randn(Cout, Cin, 3, 3, device="cuda"); w1 = torch.randn(Cout, Cin, 1, 1, device="cuda")
            wp = torch.empty((Cin, 10, Cout), dtype=torch.bfloat16, device="cuda")
            cv.pack_weight(w3, 1, out=wp, tap0=0, T=10); cv.pack_weight(w1, 1, out=wp, tap0=9, T=10)
            d = cv.dgrad_desc(N, Cin, H, H, Cout, [(3, 3, 1, 0, 0), (1, 1, 0, 1, 9)], s)
            out = cv.empty_cl(N, Cin, H, H, "cuda")
            flops = 2.0 * N * OH * OH * Cout * Cin * 10
            us = timeit(lambda: cv.launch_conv(d, dy, wp, out, src1=dy, resid=x))
        print(f"{which:6s} {Cin:5d}->{Cout:5d} @{H:3d} s{s} k{k}: {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
