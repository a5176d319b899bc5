"""Shared driver of the secondary training benches with the contract of bench.py:

    python scripts/bench_X.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/bench_X.py --gpus N ...

one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment), weak scaling (fixed per-GPU batch), gradients averaged
by `parallel.GradReducer` over RCCL (fp32 on the links), K timed steps between barrier + synchronize, MAX over ranks, ONE JSON line
from rank 0 with `roofline` (dominant instrumented conv family, HIP events on the launch stream in one extra eager step) and
`cpu_baseline` (the oracle = the reference's algorithm restated on torch-CPU, bounded sample).  The step is replayed from
hipGraphs when it captures: one graph on one rank; at N > 1 graph(forward + backward + pack) | eager all-reduce | graph(unpack +
optimizer) - the same GraphedStep bench.py uses."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15
HBM_PEAK = 8.0e12


def add_common_args(ap):
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm-dtype", choices=["fp32", "bf16"], default="fp32")
    return ap


def best_threads_run(fn, counts=(8, 16, 32, 64)):
    """fn() -> units processed; picks the torch thread count with the highest rate on one trial each."""
    host = os.cpu_count() or 1
    default = torch.get_num_threads()
    trial = {}
    for n in sorted({c for c in counts + (default,) if c <= host}):
        torch.set_num_threads(n)
        fn()
        t0 = time.perf_counter()
        units = fn()
        trial[n] = units / (time.perf_counter() - t0)
    best = max(trial, key=trial.get)
    torch.set_num_threads(best)
    return best, trial, host, default


def run(a, build_model, make_batch, loss_of, metric, workload, train_gflop_per_img=None, cpu_baseline=None, extra=None, traffic_key=None):
    """build_model() -> nn.Module (CPU); make_batch(rank, device) -> (x, target); loss_of(model, x, target) -> scalar loss."""
    from holocron_amd.parallel import ensure_ranks
    ensure_ranks(a.gpus, os.path.abspath(sys.argv[0]), sys.argv[1:])   # --gpus N starts N ranks when no launcher did
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)                      # RCCL prints its banner from C at exit: only the JSON line may reach stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import holocron_amd as h
    from holocron_amd import parallel
    from holocron_amd.ops import conv as cv
    # HC_FORCE_DIST=1 drives the N > 1 code path (process group, reducer, graph | all-reduce | graph) on ONE rank: the multi-GPU
    # runs are the driver's, this is how that path is exercised on a 1-GPU box
    force_dist = os.environ.get("HC_FORCE_DIST", "0") == "1"
    distributed = world > 1 or force_dist
    if world > 1:
        parallel.init_process_group_from_env("nccl")
    elif force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1)
    torch.manual_seed(0)
    model = build_model().to(dev).train()
    if distributed:
        parallel.broadcast_parameters(model)
    opt = h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0)
    reducer = None
    if distributed:
        reducer = parallel.GradReducer(model.parameters(), bucket_mb=128.0, force=force_dist,
                                       comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else torch.float32)
    x, target = make_batch(rank, dev)
    batch = x.shape[0]
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        loss = loss_of(model, x, target)
        loss.backward()
        loss_buf.copy_(loss.detach())

    def eager_step():
        fwd_bwd()
        if reducer is not None:
            reducer.finalize()
        opt.step()

    for _ in range(max(2, a.warmup)):
        eager_step()
    torch.cuda.synchronize()
    gstep, note = None, "eager" + (", bucketed all-reduce overlapped with backward" if distributed else "")
    if not a.no_graph:
        ok, why = 1, ""
        try:
            gstep = parallel.GraphedStep(fwd_bwd, opt, reducer)
            gstep.capture()
            probe = next(p for p in model.parameters() if p.dim() >= 2)
            before = probe.detach().clone()
            gstep.run()
            torch.cuda.synchronize()
            if not torch.isfinite(loss_buf).item() or torch.equal(before, probe.detach()):
                raise RuntimeError("graph replay did not train")
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {str(e)[:80]}"
            torch.cuda.synchronize()
        if distributed:
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if ok and not int(flag.item()):
                ok, why = 0, "capture failed on another rank"
        if ok:
            note = "hipGraph replay of the full step" if not distributed else \
                f"hipGraph(fwd + bwd + pack) | RCCL all-reduce ({a.comm_dtype}) | hipGraph(unpack + AdaBelief)"
        else:
            if gstep is not None:
                gstep.release()
            gstep = None
            if reducer is not None:
                reducer.set_overlap(True)
            note += f" (graph capture failed: {why})"

    def run_step():
        if gstep is not None:
            gstep.run()
        else:
            eager_step()

    for _ in range(a.warmup):
        run_step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
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

    roof = None
    if reducer is not None:
        reducer.set_overlap(False)
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
        if fam:
            dom = max(fam, key=lambda k: fam[k][1])
            fl, sec, n, nb = fam[dom]
            if nb / HBM_PEAK > fl / MFMA_BF16_PEAK:
                roof = {"bound": "hbm", "kernel": dom, "achieved": nb / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": nb / sec / HBM_PEAK}
            else:
                roof = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": fl / sec / MFMA_BF16_PEAK}
            # HBM bytes per launch of the dominant family from the PMC passes over this same command (scripts/pmc_families.sh; the
            # counters cannot be read from inside the process): this round's committed file when there is one, else null
            traffic = traffic_src = None
            if traffic_key is not None:
                import json as _json
                for rnd in ("r06", "r05", "r04", "r03"):
                    pf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{rnd}_pmc_{traffic_key}_traffic.json")
                    if traffic is None and os.path.exists(pf):
                        with open(pf) as fh:
                            traffic = _json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
                        if traffic is not None:
                            traffic_src = f"profiles/{rnd}_pmc_{traffic_key}_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; committed, not live)"
            roof.update({"traffic": traffic, "traffic_source": traffic_src, "launches_per_step": n, "avg_launch_ms": sec / n * 1e3,
                         "algorithmic_flops_per_launch": fl / n, "algorithmic_bytes_per_launch": nb / n,
                         "families": {k: {"tflops": v[0] / v[1] / 1e12, "gbps": v[3] / v[1] / 1e9, "ms_per_step": v[1] * 1e3,
                                          "launches": v[2]} for k, v in fam.items()}})
    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    imgs = batch * world * a.steps
    out = {"metric": metric, "value": imgs / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic",
           "config": {"workload": workload, "global_batch": batch * world, "parallelism": f"dp{world}", "mode": note,
                      "final_loss": final_loss},
           "roofline": roof, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    if train_gflop_per_img is not None:
        out["mfma_fraction_whole_step"] = train_gflop_per_img * 1e9 * imgs / dt / MFMA_BF16_PEAK / world
    if extra:
        out.update(extra(dt / a.steps, batch))
    if world == 1 and not a.no_cpu_baseline and cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
