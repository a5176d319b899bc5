"""Secondary measurement: mobileone_s0 bf16 training step (fwd + CE + bwd + AdaBelief), synthetic 224 x 224, per-GPU
batch 256, data parallel over RCCL with the contract of bench.py (scripts/_train_bench.py).  Prints one JSON line (rank 0).

    python scripts/bench_mobileone.py --gpus 1 --steps 10 --warmup 3
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import _train_bench as tb  # noqa: E402


def main():
    ap = tb.add_common_args(argparse.ArgumentParser())
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arch", default="mobileone_s0")
    a = ap.parse_args()
    import holocron_amd as h

    def build():
        return h.models.__dict__[a.arch](num_classes=1000)

    def make_batch(rank, dev):
        g = torch.Generator(device=dev).manual_seed(rank)
        return (torch.rand((a.batch, 3, 224, 224), device=dev, generator=g), torch.randint(0, 1000, (a.batch,), device=dev, generator=g))

    def loss_of(model, x, t):
        return h.nn.functional.cross_entropy(model(x).float(), t)

    tb.run(a, build, make_batch, loss_of, f"images/sec fwd+bwd+AdaBelief, {a.arch} bs{a.batch}/GPU 224^2",
           f"{a.arch} bf16 train step (fwd + CE + bwd + AdaBelief), synthetic 224^2, bs={a.batch} per MI355X (SURVEY 8f-1), "
           "random-init weights, 1000 classes", cpu_baseline=None)


if __name__ == "__main__":
    main()
