"""Per-layer roofline of the RepVGG-A0 training step at batch 256 (SURVEY.md §8d "report both": the whole-step MFMA fraction
is in bench.py, this is achieved / min(roof_MFMA, roof_HBM) per block shape and per operation).

For each of the 10 distinct block shapes one RepBlock runs forward + backward on its own; the conv launches are timed with
HIP events on the launch stream (holocron_amd.ops.conv.PROFILE, the same instrumentation bench.py uses), the BatchNorm /
ReLU passes as (block total - conv launches).  Algorithmic work per operation (bf16 activations, packed bf16 weights):

  fwd   : 2 N OH OW Cout Cin 10 FLOP ; read x, write y3 + y1, read weights             (3x3 and 1x1 fused)
  dgrad : same FLOP                  ; read dy3 + dy1 (+ dx_id), write dx, read weights
  wgrad : same FLOP                  ; read x, dy3, dy1, write fp32 dW
  BN    : 0 FLOP                     ; fwd (3 + id) A, bwd reduce (4 + id) A, bwd apply (6 + 2 id) A,  A = bytes of one output map

roof time = max(FLOP / 2.5e15, bytes / 8e12)  (MI355X_MICROARCH.md: dense bf16 MFMA, HBM3E spec); frac = roof time / measured.
Prints a markdown table (copy to profiles/)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from holocron_amd.models.classification.repvgg import RepBlock  # noqa: E402
from holocron_amd.nn.repblock_op import POOL  # noqa: E402
from holocron_amd.ops import conv as cv  # noqa: E402

MFMA, HBM = 2.5e15, 8.0e12
N = int(os.environ.get("LR_N", "256"))
SHAPES = [  # Cin, Cout, H(in), stride, identity, count in repvgg_a0
    (3, 48, 224, 2, False, 1), (48, 48, 112, 1, True, 1), (48, 48, 112, 2, False, 1), (48, 48, 56, 1, True, 2),
    (48, 96, 56, 2, False, 1), (96, 96, 28, 1, True, 4), (96, 192, 28, 2, False, 1), (192, 192, 14, 1, True, 14),
    (192, 1280, 14, 2, False, 1), (1280, 1280, 7, 1, True, 1),
]
ITERS = 5


def ev():
    return torch.cuda.Event(enable_timing=True)


def main():
    dev = torch.device("cuda:0")
    rows, tot = [], {"meas": 0.0, "roof": 0.0}
    print("| block (x count) | op | GFLOP | MB | t_MFMA us | t_HBM us | bound | measured us | achieved | frac of roof |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for Cin, Cout, H, s, ident, count in SHAPES:
        torch.manual_seed(0)
        blk = RepBlock(Cin, Cout, s, ident).to(dev).train()
        OH = (H + 2 - 3) // s + 1
        stem = Cin % 16 != 0
        x = torch.rand((N, Cin, H, H), device=dev)
        if not stem:
            x = cv.to_cl_bf16(x).requires_grad_(True)
        g = cv.to_cl_bf16(torch.rand((N, Cout, OH, OH), device=dev))
        acc = {"fwd": [], "dgrad": [], "wgrad": [], "bn_fwd": [], "bn_bwd": []}
        for it in range(ITERS + 2):
            for p in blk.parameters():
                p.grad = None
            if x.requires_grad:
                x.grad = None
            cv.PROFILE = []
            e = [ev() for _ in range(3)]
            POOL.begin(dev)
            e[0].record()
            out = blk(x)
            e[1].record()
            POOL.end()
            nf = len(cv.PROFILE)
            out.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            prof, cv.PROFILE = cv.PROFILE, None
            if it < 2:
                continue
            us = lambda a, b: a.elapsed_time(b) * 1e3
            t_f = sum(us(p[2], p[3]) for p in prof[:nf])
            t_w = sum(us(p[2], p[3]) for p in prof[nf:] if p[0] == "conv_wgrad")
            t_d = sum(us(p[2], p[3]) for p in prof[nf:] if p[0] != "conv_wgrad")
            acc["fwd"].append(t_f)
            acc["dgrad"].append(t_d)
            acc["wgrad"].append(t_w)
            acc["bn_fwd"].append(us(e[0], e[1]) - t_f)
            acc["bn_bwd"].append(us(e[1], e[2]) - t_d - t_w)
        med = {k: statistics.median(v) for k, v in acc.items()}
        A_in, A_out = 2.0 * N * Cin * H * H, 2.0 * N * Cout * OH * OH
        if stem:
            A_in = 4.0 * N * Cin * H * H       # the stem reads the fp32 NCHW image
        Wb = 2.0 * 10 * Cin * Cout
        fl = 2.0 * N * OH * OH * Cout * Cin * 10
        idf = 1 if ident else 0
        ops = [("fwd 3x3+1x1", fl, A_in + 2 * A_out + Wb, med["fwd"]),
               ("dgrad", 0.0 if stem else fl, 0.0 if stem else 2 * A_out + idf * A_in + A_in + Wb, med["dgrad"]),
               ("wgrad 3x3+1x1", fl, A_in + 2 * A_out + 2 * Wb, med["wgrad"]),
               ("BN+ReLU fwd", 0.0, (3 + idf) * A_out, med["bn_fwd"]),
               ("BN bwd (reduce+apply)", 0.0, (10 + 3 * idf) * A_out, med["bn_bwd"])]
        for name, F, B, t in ops:
            if B == 0.0:
                continue
            tm, th = F / MFMA * 1e6, B / HBM * 1e6
            roof = max(tm, th)
            ach = f"{F / t / 1e6:.0f} TFLOP/s" if tm >= th else f"{B / t / 1e6:.2f} TB/s"
            print(f"| {Cin}@{H} -> {Cout}@{OH} (x{count}) | {name} | {F / 1e9:.1f} | {B / 1e6:.0f} | {tm:.1f} | {th:.1f} | "
                  f"{'MFMA' if tm >= th else 'HBM'} | {t:.1f} | {ach} | {roof / t:.2f} |")
            tot["meas"] += t * count
            tot["roof"] += roof * count
        del blk, x, g, out
        torch.cuda.empty_cache()
    print(f"\nwhole net (blocks x count, batch {N}): measured {tot['meas'] / 1e3:.2f} ms in isolation, roofline "
          f"{tot['roof'] / 1e3:.2f} ms -> {tot['roof'] / tot['meas']:.2f} of the per-operation roofline "
          f"(fusing further, e.g. BN passes into the convs, would lower the roofline itself)")


if __name__ == "__main__":
    main()
