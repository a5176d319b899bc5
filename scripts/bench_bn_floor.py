"""Floor of the small BatchNorm-activation passes: per-launch time of hc_bn_act_apply / hc_bn_act_bwd_reduce / hc_rep_bn_bwd_finalize-less
hc_bn_act_bwd_apply on the tensor sizes of a YOLOv4 step (batch 16), replayed from a hipGraph of 40 launches (hot: one buffer set;
cold: eight sets round-robin)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from holocron_amd import _lib  # noqa: E402
from holocron_amd._lib import check, ptr, stream  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
ACT, SLOPE = 4, 0.0


def run(npix, C, sets):
    bufs = []
    for _ in range(sets):
        y = torch.randn(npix, C, device=dev).bfloat16()
        g = torch.randn(npix, C, device=dev).bfloat16()
        out = torch.empty_like(y)
        bufs.append((y, g, out))
    coef = torch.rand(4 * C, device=dev)
    bcoef = torch.rand(3 * C, device=dev)
    red = torch.zeros(int(lib.hc_get_stat_replicas()) * 4 * C, device=dev)
    res = {}

    def time_graph(fn, n=40):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for i in range(n):
                    fn(i)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * n)

    def apply(i):
        y, g, out = bufs[i % sets]
        check(lib.hc_bn_act_apply(ptr(y), ptr(coef), None, 0, None, None, ptr(out), C, npix, C, ACT, SLOPE, stream()), "apply")

    def reduce(i):
        y, g, out = bufs[i % sets]
        check(lib.hc_bn_act_bwd_reduce(ptr(g), C, ptr(y), ptr(coef), None, None, ptr(red), npix, C, ACT, SLOPE, stream()), "reduce")

    def bapply(i):
        y, g, out = bufs[i % sets]
        check(lib.hc_bn_act_bwd_apply(ptr(g), C, ptr(y), ptr(coef), ptr(bcoef), None, None, ptr(out), npix, C, ACT, SLOPE, stream()), "bapply")

    res["apply"] = time_graph(apply)
    res["bwd_reduce"] = time_graph(reduce)
    res["bwd_apply"] = time_graph(bapply)
    return res


print(f"{'npix':>8s} {'C':>5s} {'MB':>7s} sets | apply us  TB/s | reduce us TB/s | bwd_apply us TB/s")
for npix, C in [(5776, 64), (5776, 512), (5776, 1024), (23104, 256), (23104, 512), (92416, 128), (92416, 256), (369664, 64), (369664, 128)]:
    mb = npix * C * 2 / 1e6
    for sets in (1, 8):
        r = run(npix, C, sets)
        print(f"{npix:8d} {C:5d} {mb:7.1f} {sets:4d} | {r['apply']:7.1f} {2 * mb / r['apply']:6.2f} | {r['bwd_reduce']:7.1f} {2 * mb / r['bwd_reduce']:6.2f} | "
              f"{r['bwd_apply']:7.1f} {3 * mb / r['bwd_apply']:6.2f}")
