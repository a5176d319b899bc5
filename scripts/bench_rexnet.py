"""BASELINE.json configs[2]: rexnet1_0x bf16 training step (fwd + CE + bwd + AdaBelief; depthwise + 1x1 conv path), synthetic
224 x 224, 256 images per GPU (512 over the 2 GPUs the config names), data parallel over RCCL.  Prints one JSON line (rank 0).

    python scripts/bench_rexnet.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 scripts/bench_rexnet.py --gpus 2
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import _train_bench as tb  # noqa: E402


def main():
    ap = tb.add_common_args(argparse.ArgumentParser())
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (configs[2]: 512 over 2 GPUs)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    a = ap.parse_args()
    import holocron_amd as h

    def build():
        return h.models.rexnet1_0x()

    def make_batch(rank, dev):
        g = torch.Generator(device=dev).manual_seed(rank)
        return (torch.rand((a.batch, 3, 224, 224), device=dev, generator=g),
                torch.randint(0, 1000, (a.batch,), device=dev, generator=g))

    def loss_of(model, x, t):
        return h.nn.functional.cross_entropy(model(x).float(), t)      # the same criterion as two HIP launches (round 4)

    def cpu_baseline():
        """oracle.rexnet.forward (the reference's rexnet1_0x restated on torch-CPU fp32) + CE + autograd + the oracle's AdaBelief."""
        from oracle import rexnet as orx
        from oracle.optim import adabelief_step
        torch.manual_seed(0)
        sd = {k: v.detach().clone() for k, v in h.models.rexnet1_0x().state_dict().items()}
        keys = [k for k, v in sd.items() if v.dtype.is_floating_point and not ("running" in k)]
        g = torch.Generator().manual_seed(0)
        x = torch.rand((a.cpu_batch, 3, 224, 224), generator=g)
        t = torch.randint(0, 1000, (a.cpu_batch,), generator=g)
        state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in keys}
        step = [0]

        def one():
            work = dict(sd)
            params = {k: sd[k].detach().requires_grad_(True) for k in keys}
            work.update(params)
            loss = F.cross_entropy(orx.forward(work, x, training=True), t)
            grads = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
            step[0] += 1
            with torch.no_grad():
                for k, gr in zip(keys, grads):
                    if gr is not None:
                        adabelief_step(sd[k], gr, state[k][0], state[k][1], step[0], 1e-3, 0.95, 0.99, 1e-6, 0.0)
            return a.cpu_batch
        torch.set_flush_denormal(True)
        best, trial, host, default = tb.best_threads_run(one)
        t0 = time.perf_counter()
        n = sum(one() for _ in range(3))
        dt = time.perf_counter() - t0
        torch.set_num_threads(default)
        return {"value": n / dt, "unit": "images/sec", "cores": best, "host_threads": host, "kind": "port",
                "threads_tried": {str(k): round(v, 2) for k, v in trial.items()},
                "sample": f"oracle rexnet1_0x train step (torch-CPU fp32), batch {a.cpu_batch}, 3 timed iterations"}

    tb.run(a, build, make_batch, loss_of, "images/sec fwd+bwd+AdaBelief, rexnet1_0x bs256/GPU 224^2",
           "rexnet1_0x bf16 train step (fwd+CE+bwd+AdaBelief), synthetic 224^2, bs=256 per MI355X (BASELINE.json configs[2]), "
           "random-init weights, 1000 classes", train_gflop_per_img=2.39, cpu_baseline=cpu_baseline,
           # >= 30 MB/img of algorithmic HBM bytes forward (bf16, BN / activation fused), ~3.5 x for a training step
           extra=lambda sec, batch: {"hbm_floor_ms": 30e6 * 3.5 * batch / 6.29e12 * 1e3}, traffic_key="rexnet")


if __name__ == "__main__":
    main()
