"""Stub of bench.py's launch contract for the CPU test of `parallel.ensure_ranks` (tests/test_self_launch.py): same argument
handling, gloo instead of RCCL, one all-reduce as the "step", ONE JSON line from rank 0 with n_gpus = the ranks that really ran."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=0)
    a = ap.parse_args()
    from holocron_amd.parallel import ensure_ranks
    ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    total = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.init_process_group("gloo")
        for _ in range(a.steps):
            t = torch.tensor([float(rank + 1)])
            dist.all_reduce(t)
            total = t
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "steps": a.steps, "sum_of_ranks": float(total.item())}))


if __name__ == "__main__":
    main()
