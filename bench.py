#!/usr/bin/env python
"""Headline benchmark: RepVGG-A0 bf16 training step (fwd + bwd + AdaBelief), synthetic 3x224x224,
per-GPU batch 256 (BASELINE.json configs[1]); weak scaling over GPUs with RCCL gradient all-reduce.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      (starts its own N ranks: parallel.ensure_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same job; N must match)

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
# A finished group of deferred RepBlock weight gradients (e.g. the 192-channel stage's) goes to a second stream as soon as backward
# moves on to another block shape, instead of waiting for the end-of-pass flush: it then runs beside the HBM-bound passes of the
# 96- / 48-channel stages.  Same box, three pairs: 10.80 / 10.80 / 10.80 ms without, 10.69 / 10.79 / 10.68 with (read at import).
os.environ.setdefault("HC_WREP_SIDE", "1")

MFMA_BF16_PEAK = 2.5e15   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12         # HBM3E spec, same guide (6.29 TB/s measured copy)
WGRAD_STREAM_DEFAULT = "1"   # +2.2 % same-box (16.29k -> 16.65k img/s); HC_WGRAD_STREAM=0 switches it off
TRAIN_GFLOP_PER_IMG = 16.88  # conv fwd+dgrad+wgrad (stem dgrad excluded), 2*MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: ~2.5 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--profile-steps", type=int, default=8, help="instrumented eager steps behind the roofline object (untimed)")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--arch", default="repvgg_a0", help="the headline is repvgg_a0 (BASELINE.json configs[1]); other RepVGG variants for "
                                                        "side measurements (their line says so and carries no MFMA fraction)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary BASELINE configs (C3 rexnet1_0x, C4 yolov4 train / eval, C5 repvgg_a2 fp8) that rank 0 "
                         "runs after the headline's timed region at N = 1 and reports under `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--deterministic", action="store_true",
                    help="bit-reproducible steps (holocron_amd.set_deterministic: single-writer statistics slots); for A/B checks of the "
                         "step's code paths, not for the headline number")
    ap.add_argument("--loss-tail", type=int, default=0,
                    help="after the timed region run this many more steps and report each one's loss (config.loss_tail): the loss "
                         "trajectory of two runs with the same flags must agree bit for bit under --deterministic")
    ap.add_argument("--comm-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="dtype of the gradient all-reduce at N > 1 (fp32 = the reference's gradient precision; bf16 halves the "
                         "bytes on xGMI but moves an AdaBelief update by ~2e-2, tests/test_parallel_gloo.py)")
    return ap.parse_args()


def cpu_baseline(batch, iters):
    """The reference algorithm on the host cores: oracle train step (fp32), bounded sample (~20 s).  torch's default of one thread
    per hardware thread oversubscribes a batch-32 step on a 128 / 256-thread host (round 1: 7.3 img/s on 128 threads against 18.6 on
    8), so the thread count is chosen first: one timed step at each of a few counts, the best one runs the sample."""
    from oracle import repvgg as orv
    torch.set_flush_denormal(True)
    nb, a, b = orv.ARCH["repvgg_a0"]
    ch = orv.widths(orv.PLANES, a, b)
    sd = orv.init_state(nb, ch, num_classes=10, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((batch, 3, 224, 224), generator=g)
    t = torch.randint(0, 10, (batch,), generator=g)
    opt = {}
    host = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    orv.train_step(sd, opt, x, t, nb, ch)  # warm-up (allocator, oneDNN primitives)
    trial = {}
    for n in sorted({c for c in (8, 16, 32, 64, default_threads) if c <= host}):
        torch.set_num_threads(n)
        orv.train_step(sd, opt, x, t, nb, ch)
        t0 = time.perf_counter()
        orv.train_step(sd, opt, x, t, nb, ch)
        trial[n] = batch / (time.perf_counter() - t0)
    best = max(trial, key=trial.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(iters):
        orv.train_step(sd, opt, x, t, nb, ch)
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": batch * iters / dt, "unit": "images/sec", "cores": best, "host_threads": host, "kind": "port",
            "threads_tried": {str(k): round(v, 2) for k, v in trial.items()},
            "sample": f"oracle (torch-CPU fp32 restatement of the reference) repvgg_a0 train step, batch {batch}, "
                      f"1 warm-up + {iters} timed iterations on the best of the tried thread counts"}


SECONDARY = [   # (key, BASELINE.json config, script under scripts/, extra arguments)
    ("rexnet1_0x_train_bs256", "configs[2]", "bench_rexnet.py", []),
    ("yolov4_train_bs16_608", "configs[3]", "bench_yolov4.py", []),
    ("yolov4_eval_bs16_608", "configs[3] (eval: forward + decode + NMS)", "bench_yolov4.py", ["--eval"]),
    ("repvgg_a2_fp8_infer_bs1024", "configs[4]", "bench_repvgg_fp8.py", []),
]


def run_secondary(steps, timeout_s=420):
    """The other single-GPU BASELINE configs, each by its own driver under scripts/ (same timing contract as this file: W untimed
    steps, K timed steps between synchronisations, one JSON line) in a fresh process AFTER the headline's timed region, so that the
    driver's record carries them.  CPU baselines are skipped here (they are in the per-config lines committed
    under profiles/); a failed or slow config is reported as such, never silently dropped."""
    import subprocess
    out = {}
    for key, cfg, script, extra in SECONDARY:
        cmd = [sys.executable, os.path.join(ROOT, "scripts", script), "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if r.returncode != 0 or line is None:
                out[key] = {"baseline_config": cfg, "error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                continue
            j = json.loads(line)
            roof = j.get("roofline") or {}
            out[key] = {"baseline_config": cfg, "metric": j.get("metric"), "value": j.get("value"), "unit": j.get("unit"),
                        "ms_per_step": j.get("ms_per_step"), "steps": j.get("steps"), "dtype": j.get("dtype"),
                        "mode": (j.get("config") or {}).get("mode"),
                        "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")},
                        "mfma_fraction_whole_step": j.get("mfma_fraction_whole_step", j.get("mfma_fraction")),
                        "wall_s": round(time.perf_counter() - t0, 1)}
        except subprocess.TimeoutExpired:
            out[key] = {"baseline_config": cfg, "error": f"timeout after {timeout_s} s"}
        except Exception as e:  # noqa: BLE001
            out[key] = {"baseline_config": cfg, "error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def main():
    args = parse()
    # `bench.py --gpus N` IS the N-rank job: with no launcher around it this process starts the N ranks itself (one per GPU) and
    # exits with their status; under torchrun WORLD_SIZE must equal --gpus (a mismatch is a mis-launch, not a 1-GPU measurement)
    from holocron_amd.parallel import ensure_ranks
    ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
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

    if args.deterministic:
        h.set_deterministic(True)
    torch.manual_seed(0)
    model = getattr(h.models, args.arch)(num_classes=10).to(dev).train()
    headline = args.arch == "repvgg_a0"
    if distributed:
        parallel.broadcast_parameters(model)
    opt = h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)
    reducer, cut, cut2 = None, None, None
    if distributed:
        # Backward in two pieces: the last block + head hold 66 % of the parameters (the 1280 x 1280 x 3 x 3 conv) and their
        # gradients come first; their bucket ends exactly at the cut, so its all-reduce runs behind the rest of backward.
        rear_mod = model.features[-1][-1]
        rear = {id(p) for p in rear_mod.parameters()} | {id(p) for p in model.head.parameters()}
        # The first cut sits in the MIDDLE of the 192-channel stage (features[3][8]), not right behind the last block: the weight
        # gradient of the 1280 x 1280 block (0.63 ms) runs on the side stream and must be joined before its graph ends, so the first
        # graph needs main-stream work for it to hide behind (one-rank cost of the N > 1 path, same box: 11.10 ms with the cut behind
        # the last block, 11.01 behind the last stage, 10.90 here, 10.95 at features[3][11]; single graph 10.56).  The first bucket
        # grows from 65.6 to 86 MB and still has the remaining ~6 ms of backward to be reduced behind.  HC_BENCH_CUT0="stage:block"
        # moves it, "" puts it back behind the last block.
        c0 = os.environ.get("HC_BENCH_CUT0", "3:8" if len(model.features[3]) > 8 else "")
        if c0:
            si, bi = (int(v) for v in c0.split(":"))
            flat = [(i, j) for i, st in enumerate(model.features) for j in range(len(st))]
            rear_mod = model.features[si][bi]
            rear = {id(p) for (i, j) in flat if (i, j) >= (si, bi) for p in model.features[i][j].parameters()} | {id(p) for p in model.head.parameters()}
        front_last = next(p for p in reversed(list(model.parameters())) if id(p) not in rear)
        comm = torch.bfloat16 if args.comm_dtype == "bf16" else torch.float32
        # ... and in three: a second cut in front of the 192-channel stage.  The middle bucket (that stage + the first 1280-channel
        # block, 8 M parameters) is then reduced behind the backward of the 48 / 96-channel stages (the HBM-bound third of the pass),
        # and what stays exposed is their own 0.55 M parameters (2.2 MB in fp32) instead of a third of the model.
        mid_mod = model.features[3][0]
        mid = {id(p) for st in model.features[3:] for p in st.parameters()} | rear
        early_last = next((p for p in reversed(list(model.parameters())) if id(p) not in mid), None)
        bounds = [front_last] + ([early_last] if early_last is not None else [])
        reducer = parallel.GradReducer(model.parameters(), bucket_mb=256.0, comm_dtype=comm, force=force_dist, new_bucket_at=bounds)
        cut = parallel.BackwardCut(rear_mod)
        if len(bounds) == 2:
            cut2 = parallel.BackwardCut(mid_mod)

    g = torch.Generator(device=dev).manual_seed(rank)
    x = torch.rand((args.batch, 3, 224, 224), device=dev, generator=g)
    t = torch.randint(0, 10, (args.batch,), device=dev, generator=g)
    loss_buf = torch.zeros((), device=dev)

    def seg_fwd_bwd():
        opt.zero_grad(set_to_none=True)
        logits = model(x)
        # the criterion of references/classification/train.py:194 as two HIP launches (torch composes it from ~25 small kernels);
        loss = h.nn.functional.cross_entropy(logits, t, label_smoothing=0.1)
        loss.backward()
        loss_buf.copy_(loss.detach())

    segments = [seg_fwd_bwd] if cut is None else [seg_fwd_bwd, cut.continue_backward] + ([cut2.continue_backward] if cut2 is not None else [])

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
            mb = [sum(t.numel() * t.element_size() for t in sp) / 1e6 for sp in gstep.spans] if distributed else []
            graph_note = ("weight gradients on a second stream; " if wgrad_side else "") + ("hipGraph replay of the full step" if not distributed else
                          f"hipGraphs with eager RCCL all-reduces of the {args.comm_dtype} gradient between them (fwd + bwd down to the first cut | "
                          + " | ".join(f"all-reduce {m:.1f} MB" + (" behind: next part of bwd" if i + 1 < len(mb) else " (exposed)")
                                       for i, m in enumerate(mb)) + " | unpack + AdaBelief)")
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
    loss_tail = []
    for _ in range(args.loss_tail):          # untimed: every rank runs them (they hold collectives at N > 1)
        run_step()
        loss_tail.append(float(loss_buf.item()))

    # ---- roofline: instrumented eager step, HIP events on the launch stream -----------------------
    roof = None
    cv.set_wgrad_side_stream(False)
    if reducer is not None:
        reducer.set_overlap(False)   # the instrumented step below runs on rank 0 only: no collectives, no hooks
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters())
        nprof = max(1, args.profile_steps)
        fam = {}
        step_ms = 0.0
        for _ in range(nprof):
            cv.PROFILE = []
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            fwd_bwd()
            with cv.profiled("optimizer", 0.0, 28.0 * nparam):       # AdaBelief: p r/w, g r, m r/w, s r/w in fp32
                opt.step()
            s1.record()
            torch.cuda.synchronize()
            step_ms += s0.elapsed_time(s1)
            for name, flops, e0, e1, nbytes in cv.PROFILE:
                f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
                f[0] += flops
                f[1] += e0.elapsed_time(e1) * 1e-3
                f[2] += 1
                f[3] += nbytes
        cv.PROFILE = None
        for f in fam.values():          # per instrumented step
            f[0], f[1], f[2], f[3] = f[0] / nprof, f[1] / nprof, f[2] // nprof, f[3] / nprof
        covered_ms = sum(f[1] for f in fam.values()) * 1e3
        # the dominant family over ALL the time of the step (round 2 only saw the conv launches: VERDICT r2 weak #6)
        dom = max(fam, key=lambda k: fam[k][1])
        fl, sec, n, alg_bytes = fam[dom]
        # which roof bounds the family: its algorithmic FLOPs at the dense bf16 MFMA peak against its algorithmic bytes at the HBM peak
        t_mfma, t_hbm = fl / MFMA_BF16_PEAK, alg_bytes / HBM_PEAK
        # HBM bytes per launch from the PMC passes over this same command (scripts/pmc_step.sh; the counters cannot be read from
        # inside the process): taken from this round's committed file when there is one, else null
        traffic, traffic_src = None, None
        for cand in ("r06_pmc_step_traffic.json", "r05_pmc_step_traffic.json", "r04_pmc_step_traffic.json", "r03_pmc_step_traffic.json", "r02_pmc_step_traffic.json"):
            pmc_file = os.path.join(ROOT, "profiles", cand)
            if headline and args.batch == 256 and os.path.exists(pmc_file) and traffic is None:
                with open(pmc_file) as fh:
                    traffic = json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_src = "profiles/" + cand + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; a committed measurement of this command, not read live)"
        if t_hbm > t_mfma:
            roof = {"bound": "hbm", "kernel": dom, "achieved": alg_bytes / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": alg_bytes / sec / HBM_PEAK}
        else:
            roof = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": fl / sec / MFMA_BF16_PEAK}
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes / n,
                     "algorithmic_flops_per_launch": fl / n, "launches_per_step": n, "avg_launch_ms": sec / n * 1e3,
                     "instrumented_steps": nprof, "instrumented_step_ms": step_ms / nprof, "covered_ms_per_step": covered_ms,
                     "families": {k: {"tflops": v[0] / v[1] / 1e12, "gbps": v[3] / v[1] / 1e9, "ms_per_step": v[1] * 1e3,
                                      "launches": v[2],
                                      "bound": ("latency" if v[0] == 0 and v[3] == 0 else
                                                "hbm" if v[3] / HBM_PEAK > v[0] / MFMA_BF16_PEAK else "mfma"),
                                      "frac": max(v[3] / HBM_PEAK, v[0] / MFMA_BF16_PEAK) / v[1]}
                                  for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}})

    # north_star's own metric: conv FLOPs / conv-family time against the dense bf16 MFMA peak, over every family that executes
    # convolution MACs (forward, data gradient, weight gradient; BN / packing / optimizer launches are not in it)
    conv_frac = None
    if rank == 0 and roof is not None:
        cf = {k: v for k, v in fam.items() if v[0] > 0.0}
        c_fl, c_sec = sum(v[0] for v in cf.values()), sum(v[1] for v in cf.values())
        if c_sec > 0:
            conv_frac = {"frac": c_fl / c_sec / MFMA_BF16_PEAK, "tflops": c_fl / c_sec / 1e12, "peak_tflops": MFMA_BF16_PEAK / 1e12,
                         "conv_ms_per_step": c_sec * 1e3, "conv_gflop_per_step": c_fl / 1e9, "families": sorted(cf),
                         "source": "HIP events around every conv launch of the instrumented eager steps (same as roofline.families)"}

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    imgs = args.batch * world * args.steps
    ms = dt / args.steps * 1e3
    out = {
        "metric": "images/sec fwd+bwd+AdaBelief, RepVGG-A0 bs256/GPU 224^2" if headline else f"images/sec fwd+bwd+AdaBelief, {args.arch} bs{args.batch}/GPU 224^2",
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
        "config": {"workload": ("repvgg_a0 bf16 train step (fwd+bwd+AdaBelief), synthetic 224^2, bs=256 per MI355X "
                                "(BASELINE.json configs[1]), random-init weights, 10 classes, CE label_smoothing 0.1") if headline else
                               f"{args.arch} bf16 train step (fwd+bwd+AdaBelief), synthetic 224^2, bs={args.batch} per MI355X (side measurement, "
                               "not a BASELINE config), random-init weights, 10 classes, CE label_smoothing 0.1",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "mode": graph_note,
                   "final_loss": final_loss, **({"loss_tail": loss_tail, "deterministic": bool(args.deterministic)} if loss_tail else {})},
        "mfma_fraction_whole_step": (TRAIN_GFLOP_PER_IMG * 1e9 * imgs / dt / MFMA_BF16_PEAK / world) if headline else None,
        "conv_mfma_fraction": conv_frac,
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_iters)
    if world == 1 and headline and not args.no_secondary and not force_dist:
        # release this process's device memory first: the secondary drivers are separate processes on the same GPU
        del gstep, model, opt, x
        torch.cuda.empty_cache()
        out["secondary"] = run_secondary(args.secondary_steps)
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
