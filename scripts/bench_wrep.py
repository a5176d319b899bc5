"""Per-shape timing of the fused RepBlock weight-gradient kernel against the two-launch path it replaces, at the repvgg_a0
batch-256 shapes with the group sizes of the real step.  usage: python scripts/bench_wrep.py [--iters 20]
(The planner's experiment knobs of rounds 2-4 - HC_WREP_PF / _R / _TILE / _HV / _PV / _LDS - were removed in round 5: its choices are what it ships.)"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holocron_amd import _lib  # noqa: E402
from holocron_amd.ops import conv as cv  # noqa: E402

GROUPS = [  # Cin, H, Cout, stride, blocks of this shape in repvgg_a0
    (48, 112, 48, 1, 1), (48, 112, 48, 2, 1), (48, 56, 48, 1, 2), (48, 56, 96, 2, 1), (96, 28, 96, 1, 4), (96, 28, 192, 2, 1),
    (192, 14, 192, 1, 14),
]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-old", action="store_true")
    a = ap.parse_args()
    N = a.batch
    lib = _lib.load()
    dev = torch.device("cuda:0")
    print(f"{'shape':<28}{'jobs':>5}{'plan MR,NR,R,PF,nsplit,grid,smem,NSLOT':>44}{'fused us':>10}{'/block':>8}{'TFLOP/s':>9}{'GB/s':>8}{'old us/blk':>11}")
    tot_new = tot_old = 0.0
    for cin, H, cout, s, nb in GROUPS:
        key = (N, cin, H, H, cout, s)
        OH = (H - 1) // s + 1
        jobs = []
        for _ in range(nb):
            x = torch.randn((N, cin, H, H), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dy3 = torch.randn((N, cout, OH, OH), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dy1 = torch.randn((N, cout, OH, OH), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dw3 = torch.empty((cout, cin, 3, 3), device=dev)
            dw1 = torch.empty((cout, cin, 1, 1), device=dev)
            jobs.append((x, dy3, dy1, dw3, dw1))
        plan = (C.c_int32 * 8)()
        ok = cv._WREP.supported(key) and lib.hc_rep_wgrad_plan(C.byref(cv._WREP._desc(key, nb)), plan) == 0
        flops = 2.0 * N * OH * OH * cout * 10 * cin * nb
        nbytes = nb * (2 * N * H * H * cin + 4 * N * OH * OH * cout)
        t_new = float("nan")
        if ok:
            ptrs = [(j[0], j[1], j[2], j[3].data_ptr(), j[4].data_ptr()) for j in jobs]
            t_new = timed(lambda: cv._WREP.launch(key, ptrs), a.iters)
        t_old = float("nan")
        if not a.no_old:
            def old():
                for (x, dy3, dy1, dw3, dw1) in jobs:
                    cv.conv_wgrad(x, dy3, cin, cout, 3, 3, s, 1, out=dw3)
                    cv.conv_wgrad(x, dy1, cin, cout, 1, 1, s, 0, out=dw1)
            t_old = timed(old, max(3, a.iters // 4))
        tot_new += t_new
        tot_old += t_old
        print(f"{str((cin, H, cout, s)):<28}{nb:>5}{str(list(plan)):>44}{t_new:>10.1f}{t_new / nb:>8.1f}{flops / t_new / 1e6:>9.0f}"
              f"{nbytes / t_new / 1e3:>8.0f}{t_old / nb:>11.1f}")
    print(f"sum fused {tot_new:.0f} us, two-launch path {tot_old:.0f} us per step (these layers)")


if __name__ == "__main__":
    main()
