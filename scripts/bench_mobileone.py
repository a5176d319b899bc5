"""Secondary measurement: mobileone_s0 bf16 training step (fwd + CE + bwd + AdaBelief), synthetic 224 x 224, per-GPU
batch 256, data parallel over RCCL with the contract of bench.py (scripts/_train_bench.py).  Prints one JSON line (rank 0).

    python scripts/bench_mobileone.py --gpus 1 --steps 10 --warmup 3
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import _train_bench as tb  # noqa: E402


def main():
    ap = tb.add_common_args(argparse.ArgumentParser())
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arch", default="mobileone_s0")
    ap.add_argument("--cpu-batch", type=int, default=16)
    a = ap.parse_args()
    import holocron_amd as h

    def build():
        return h.models.__dict__[a.arch](num_classes=1000)

    def make_batch(rank, dev):
        g = torch.Generator(device=dev).manual_seed(rank)
        return (torch.rand((a.batch, 3, 224, 224), device=dev, generator=g), torch.randint(0, 1000, (a.batch,), device=dev, generator=g))

    def loss_of(model, x, t):
        return h.nn.functional.cross_entropy(model(x).float(), t)

    def cpu_baseline():
        """oracle.mobileone.forward (the reference's MobileOne restated on torch-CPU fp32) + CE + autograd + the oracle's AdaBelief."""
        import torch.nn.functional as F
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import mobileone as omo
        from oracle.optim import adabelief_step
        torch.manual_seed(0)
        sd = {k: v.detach().clone() for k, v in build().state_dict().items()}
        keys = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
        g = torch.Generator().manual_seed(0)
        x = torch.rand((a.cpu_batch, 3, 224, 224), generator=g)
        t = torch.randint(0, 1000, (a.cpu_batch,), generator=g)
        state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in keys}
        step = [0]

        def one():
            work = dict(sd)
            params = {k: sd[k].detach().requires_grad_(True) for k in keys}
            work.update(params)
            loss = F.cross_entropy(omo.forward(work, x, training=True), t)
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
                "sample": f"oracle {a.arch} train step (torch-CPU fp32), batch {a.cpu_batch}, 3 timed iterations"}

    tb.run(a, build, make_batch, loss_of, f"images/sec fwd+bwd+AdaBelief, {a.arch} bs{a.batch}/GPU 224^2",
           f"{a.arch} bf16 train step (fwd + CE + bwd + AdaBelief), synthetic 224^2, bs={a.batch} per MI355X (SURVEY 8f-1), "
           "random-init weights, 1000 classes", cpu_baseline=cpu_baseline if a.arch == "mobileone_s0" else None, traffic_key="mobileone")


if __name__ == "__main__":
    main()
