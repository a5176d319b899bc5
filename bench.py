#!/usr/bin/env python
"""Headline benchmark: RepVGG-A0 bf16 training step (fwd + bwd + AdaBelief), synthetic 3x224x224,
per-GPU batch 256 (BASELINE.json configs[1]); weak scaling over GPUs with RCCL gradient all-reduce.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` is whole-job images/sec with inputs resident in HBM.
`roofline` is measured live with HIP events on the launch stream in an extra instrumented step
(not part of the timed region) for the dominant kernel family; `cpu_baseline` times the CPU oracle
(the reference's algorithm restated in torch-CPU fp32) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
WGRAD_STREAM_DEFAULT = "1"   # +2.2 % same-box (16.29k -> 16.65k img/s); HC_WGRAD_STREAM=0 switches it off
TRAIN_GFLOP_PER_IMG = 16.88  # conv fwd+dgrad+wgrad (stem dgrad excluded), 2*MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(batch, iters):
    """The reference algorithm on the host cores: oracle train step (fp32), bounded sample."""
    from oracle import repvgg as orv
    torch.set_flush_denormal(True)
    nb, a, b = orv.ARCH["repvgg_a0"]
    ch = orv.widths(orv.PLANES, a, b)
    sd = orv.init_state(nb, ch, num_classes=10, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((batch, 3, 224, 224), generator=g)
    t = torch.randint(0, 10, (batch,), generator=g)
    opt = {}
    orv.train_step(sd, opt, x, t, nb, ch)  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        orv.train_step(sd, opt, x, t, nb, ch)
    dt = time.perf_counter() - t0
    return {"value": batch * iters / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement of the reference) repvgg_a0 train step, batch {batch}, "
                      f"1 warm-up + {iters} timed iterations"}


def main():
    args = parse()
    # Only the JSON line may reach stdout: libraries that print from C (RCCL's version banner sits in the
    # stdio buffer until exit, i.e. after our line) are sent to stderr, the result goes to the real fd 1.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import torch.distributed as dist
    import holocron_amd as h
    from holocron_amd import parallel
    from holocron_amd.ops import conv as cv

    # HC_FORCE_DIST=1 drives the multi-GPU code path (process group, reducer, two-graph step) on one rank:
    # the 8-GPU runs are the driver's, this is how that path is exercised on a 1-GPU box
    force_dist = os.environ.get("HC_FORCE_DIST", "0") == "1"
    distributed = world > 1 or force_dist
    if world > 1:
        parallel.init_process_group_from_env("nccl")
    elif force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)

    torch.manual_seed(0)
    model = h.models.repvgg_a0(num_classes=10).to(dev).train()
    if distributed:
        parallel.broadcast_parameters(model)
    opt = h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)
    reducer, cut = None, None
    if distributed:
        # Backward in two pieces: the last block + head hold 66 % of the parameters (the 1280 x 1280 x 3 x 3 conv) and their
        # gradients come first; their bucket ends exactly at the cut, so its all-reduce runs behind the rest of backward.
        rear_mod = model.features[-1][-1]
        rear = {id(p) for p in rear_mod.parameters()} | {id(p) for p in model.head.parameters()}
        front_last = next(p for p in reversed(list(model.parameters())) if id(p) not in rear)
        reducer = parallel.GradReducer(model.parameters(), bucket_mb=64.0, comm_dtype=torch.bfloat16, force=force_dist,
                                       new_bucket_at=[front_last])
        cut = parallel.BackwardCut(rear_mod)

    g = torch.Generator(device=dev).manual_seed(rank)
    x = torch.rand((args.batch, 3, 224, 224), device=dev, generator=g)
    t = torch.randint(0, 10, (args.batch,), device=dev, generator=g)
    loss_buf = torch.zeros((), device=dev)

    def seg_fwd_bwd():
        opt.zero_grad(set_to_none=True)
        logits = model(x)
        loss = torch.nn.functional.cross_entropy(logits, t, label_smoothing=0.1)
        loss.backward()
        loss_buf.copy_(loss.detach())

    segments = [seg_fwd_bwd] if cut is None else [seg_fwd_bwd, cut.continue_backward]

    def fwd_bwd():
        for seg in segments:
            seg()

    def step():          # eager: per-bucket all-reduces issued from the autograd hooks, overlapped with backward
        fwd_bwd()
        if reducer is not None:
            reducer.finalize()
        opt.step()

    # ---- warm-up (eager) ----------------------------------------------------------------------
    n_eager_warm = max(2, args.warmup) if not args.no_graph else args.warmup
    for _ in range(n_eager_warm):
        step()
    torch.cuda.synchronize()

    # ---- hipGraph capture of the whole step (one graph; two around the all-reduce when distributed) ----
    # weight gradients on a second stream (parallel branch of the captured graph): HC_WGRAD_STREAM=0/1
    wgrad_side = os.environ.get("HC_WGRAD_STREAM", WGRAD_STREAM_DEFAULT) == "1" and not args.no_graph
    gstep = None
    graph_note = "eager" + (", bucketed all-reduce overlapped with backward" if distributed else "")
    if not args.no_graph:
        ok, why = 1, ""
        try:
            cv.set_wgrad_side_stream(wgrad_side)
            gstep = parallel.GraphedStep(segments, opt, reducer)
            gstep.capture()
            # replay must keep training: loss finite and parameters moving
            before = model.head.weight.detach().clone()
            gstep.run()
            torch.cuda.synchronize()
            if not torch.isfinite(loss_buf).item() or torch.equal(before, model.head.weight.detach()):
                raise RuntimeError("graph replay did not train")
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {str(e)[:80]}"
            torch.cuda.synchronize()
        if distributed:          # every rank replays or none does (the two modes issue different collectives)
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if ok and not int(flag.item()):
                ok, why = 0, "capture failed on another rank"
        if ok:
            graph_note = ("weight gradients on a second stream; " if wgrad_side else "") + ("hipGraph replay of the full step" if not distributed else
                          "hipGraphs with eager RCCL all-reduces of the bf16 gradient between them (fwd + bwd of last block/head | "
                          f"all-reduce {sum(t.numel() for t in gstep.spans[0]) * 2 / 1e6:.1f} MB behind: rest of bwd | all-reduce "
                          f"{sum(t.numel() for t in gstep.spans[-1]) * 2 / 1e6:.1f} MB | unpack + AdaBelief)")
        else:
            if gstep is not None:
                gstep.release()
            gstep = None
            cv.set_wgrad_side_stream(False)
            if reducer is not None:
                reducer.set_overlap(True)
            graph_note += f" (graph capture failed: {why})"

    def run_step():
        if gstep is not None:
            gstep.run()
        else:
            step()

    for _ in range(args.warmup):
        run_step()

    # ---- timed region ---------------------------------------------------------------------------
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss_buf.item())

    # ---- roofline: instrumented eager step, HIP events on the launch stream -----------------------
    roof = None
    cv.set_wgrad_side_stream(False)
    if reducer is not None:
        reducer.set_overlap(False)   # the instrumented step below runs on rank 0 only: no collectives, no hooks
    if rank == 0:
        cv.PROFILE = []
        fwd_bwd()
        opt.step()
        torch.cuda.synchronize()
        fam = {}
        for name, flops, e0, e1, nbytes in cv.PROFILE:
            f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
            f[0] += flops
            f[1] += e0.elapsed_time(e1) * 1e-3
            f[2] += 1
            f[3] += nbytes
        cv.PROFILE = None
        dom = max(fam, key=lambda k: fam[k][1])
        fl, sec, n, alg_bytes = fam[dom]
        # HBM bytes per launch from the PMC passes over this same command (scripts/pmc_step.sh; the counters cannot be read
        # from inside the process), committed next to the kernel-trace summary
        traffic = None
        pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_step_traffic.json")
        if args.batch == 256 and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                traffic = json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": MFMA_BF16_PEAK / 1e12,
                "unit": "TFLOP/s", "frac": fl / sec / MFMA_BF16_PEAK, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_step_traffic.json)",
                "algorithmic_bytes_per_launch": alg_bytes / n,
                "launches_per_step": n, "avg_launch_ms": sec / n * 1e3,
                "families": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] * 1e3, "launches": v[2]}
                             for k, v in fam.items()}}

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    imgs = args.batch * world * args.steps
    ms = dt / args.steps * 1e3
    out = {
        "metric": "images/sec fwd+bwd+AdaBelief, RepVGG-A0 bs256/GPU 224^2",
        "value": imgs / dt,
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "repvgg_a0 bf16 train step (fwd+bwd+AdaBelief), synthetic 224^2, bs=256 per MI355X "
                               "(BASELINE.json configs[1]), random-init weights, 10 classes, CE label_smoothing 0.1",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "mode": graph_note,
                   "final_loss": final_loss},
        "mfma_fraction_whole_step": TRAIN_GFLOP_PER_IMG * 1e9 * imgs / dt / MFMA_BF16_PEAK / world,
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_iters)
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
