"""hc_spp_fwd / hc_spp_bwd on YOLOv4's SPP input (16 x 512 x 19 x 19): LDS-tiled kernels against the window walk (HC_SPP_TILE=0), us per
launch from a hipGraph of 20 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from holocron_amd import _lib  # noqa: E402
from holocron_amd._lib import check, ptr, stream  # noqa: E402
from holocron_amd.ops.conv import empty_cl, to_cl_bf16  # noqa: E402

lib = _lib.load()
N, Cc, H, W = 16, 512, 19, 19
x = to_cl_bf16(torch.randn((N, Cc, H, W), device="cuda"))
gr = to_cl_bf16(torch.randn((N, 4 * Cc, H, W), device="cuda"))
out = empty_cl(N, 4 * Cc, H, W, x.device)
idx = torch.zeros((3, N, H, W, Cc), dtype=torch.uint8, device=x.device)
dx = empty_cl(N, Cc, H, W, x.device)


def timed(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(n):
                fn()
    gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for mode in ("0", "1"):
    os.environ["HC_SPP_TILE"] = mode
    f = timed(lambda: check(lib.hc_spp_fwd(ptr(x), ptr(out), ptr(idx), N, H, W, Cc, stream()), "f"))
    b = timed(lambda: check(lib.hc_spp_bwd(ptr(gr), ptr(idx), ptr(dx), N, H, W, Cc, stream()), "b"))
    print(f"HC_SPP_TILE={mode}: fwd {f:.1f} us, bwd {b:.1f} us")
