"""Secondary measurement (SURVEY.md §8d the next-row model of §8f): mobileone_s0 bf16 training step (fwd + CE + bwd + AdaBelief), synthetic
224 x 224, per-GPU batch 256, on one MI355X.  Prints one JSON line.

    python scripts/bench_rexnet.py --batch 256 --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import holocron_amd as h  # noqa: E402
from _train_bench import timed_training  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--arch", default="mobileone_s0")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = h.models.__dict__[a.arch](num_classes=1000).to(dev).train()
    opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0)
    x = torch.rand((a.batch, 3, 224, 224), device=dev)
    t = torch.randint(0, 1000, (a.batch,), device=dev)

    dt, loss, mode = timed_training(m, opt, x, t, lambda out, tgt: F.cross_entropy(out.float(), tgt), a.steps, a.warmup,
                                    use_graph=not a.no_graph)
    print(json.dumps({"metric": "images/sec train step (fwd+CE+bwd+AdaBelief), mobileone_s0 224^2", "value": a.batch / dt,
                      "unit": "img/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "dtype": "bf16",
                      "data": "synthetic", "config": {"workload": f"mobileone_s0 224^2 bs{a.batch}"},
                      "loss": loss, "mode": mode,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
