"""BASELINE.json configs[3]: darknet53-CSP + YOLOv4 head (nc = 80), CIoU loss, synthetic 608 x 608, 16 images per GPU (128 over the
8 GPUs the config names), data parallel over RCCL: forward + four-part loss + backward + AdaBelief.  `--eval` times forward +
decode + NMS on one GPU instead.  Prints one JSON line (rank 0).

    python scripts/bench_yolov4.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 scripts/bench_yolov4.py --gpus 8
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import _train_bench as tb  # noqa: E402

TRAIN_GFLOP_PER_IMG = 385.0   # 3 x 2 x 64.195 GMAC (fwd + dgrad + wgrad) at 608^2, nc=80


def targets(batch, g, dev):
    out = []
    for i in range(batch):       # the reference's own recipe (tests/test_models_detection.py:40-42), 1..8 boxes per image
        k = 1 + i % 8
        b = torch.rand((k, 4), generator=g)
        b[:, :2] *= b[:, 2:]
        b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.02).clamp(max=0.999)
        out.append({"boxes": b.to(dev), "labels": torch.randint(0, 80, (k,), generator=g).to(dev)})
    return out


def main():
    ap = tb.add_common_args(argparse.ArgumentParser())
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (configs[3]: 128 over 8 GPUs)")
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    a = ap.parse_args()
    # The step is replayed from a hipGraph again (round 3): the ragged ground truth is packed ONCE into device tensors
    # (models.detection.yolov4.PackedTargets) instead of on every forward - the per-forward packing became pageable memcpy nodes that
    # re-read freed host buffers on replay (the memory fault of round 2); packing under capture now raises instead.
    import holocron_amd as h
    from holocron_amd.models.detection.yolov4 import PackedTargets

    if a.eval:
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        m = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev).eval()
        x = torch.rand((a.batch, 3, a.size, a.size), generator=torch.Generator().manual_seed(1)).to(dev)
        with torch.no_grad():
            for _ in range(a.warmup):
                out = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out = m(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        # what the post-processing saw: 22 743 predictors per 608^2 image (3 x (19^2 + 38^2 + 76^2), yolov4.py:302-336); with random-init
        # weights about half of them pass the objectness threshold, so the NMS works on thousands of candidates per image and scale
        npred = 3 * sum((a.size // s) ** 2 for s in (32, 16, 8))
        kept = [int(o["boxes"].shape[0]) for o in out]
        line = {"metric": "images/sec eval fwd+decode+NMS, YOLOv4 608^2 nc=80", "value": a.batch / dt, "unit": "images/sec",
                "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "bf16",
                "data": "synthetic", "config": {"workload": f"yolov4 {a.size}^2 bs{a.batch} eval: forward + decode + score filter + NMS "
                                                            "(BASELINE.md section 3 row C4), random-init weights, 80 classes"},
                "predictors_per_image": npred, "detections_per_image_mean": sum(kept) / len(kept), "detections_img0": kept[0]}
        # ---- roofline: one instrumented eval pass (HIP events on the launch stream around every conv / BatchNorm launch family, as
        # in the training benches) + the post-processing as its own family: decode reads the padded logits of every predictor once
        # (2 B x (5 + nc)) and writes boxes / scores / labels; the batched NMS builds an n x n / 64-word suppression bitmap per
        # (image, scale) problem from its n candidates (yolov4.py:302-336)
        from holocron_amd.ops import conv as cv
        import importlib
        ymod = importlib.import_module("holocron_amd.models.detection.yolov4")      # (the package attribute `yolov4` is the constructor)
        real_pp = ymod.post_process_scales
        pp = {}

        def timed_pp(layers, outs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real_pp(layers, outs)
            e1.record()
            pp["ev"] = (e0, e1)
            return r
        ymod.post_process_scales = timed_pp
        cv.PROFILE = []
        try:
            with torch.no_grad():
                out = m(x)
            torch.cuda.synchronize()
            prof = cv.PROFILE
        finally:
            cv.PROFILE = None
            ymod.post_process_scales = real_pp
        fam = {}
        for name, flops, e0, e1, nbytes in prof:
            f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
            f[0] += flops; f[1] += e0.elapsed_time(e1) * 1e-3; f[2] += 1; f[3] += nbytes
        if "ev" in pp:
            kept_total = sum(int(o["boxes"].shape[0]) for o in out)
            dec_bytes = a.batch * npred * ((5 + 80) * 2.0 + 4 * 4 + 4 + 8)           # logits read; boxes, score, label written
            # candidates per problem are not kept by post_process_scales: bound the bitmap by the predictors that passed the threshold
            # on average (detections_per_image is a lower bound of the candidates, npred the upper one) - reported as bytes of the
            # decode only, the NMS part is latency-bound index work (bit-exact integer decisions)
            fam["post_process"] = [0.0, pp["ev"][0].elapsed_time(pp["ev"][1]) * 1e-3, 1, dec_bytes]
        MF, HB = 2.5e15, 8.0e12
        if fam:
            dom = max(fam, key=lambda k: fam[k][1])
            fl, sec, nl, nb = fam[dom]
            t_m, t_h = fl / MF, nb / HB
            bound = "mfma" if t_m >= t_h else "hbm"
            # HBM-side bytes per launch of the dominant family from the committed PMC passes of this command (scripts/pmc_families.sh)
            traffic = traffic_src = None
            pf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_pmc_yolov4_eval_traffic.json")
            if os.path.exists(pf):
                with open(pf) as fh:
                    traffic = json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_src = "profiles/r06_pmc_yolov4_eval_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; committed, not live)"
            line["roofline"] = {"bound": bound, "kernel": dom,
                                "achieved": (fl / sec / 1e12) if bound == "mfma" else (nb / sec / 1e9),
                                "peak": MF / 1e12 if bound == "mfma" else HB / 1e9, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                "frac": max(t_m, t_h) / sec, "traffic": traffic, "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": nb / nl, "launches_per_pass": nl, "avg_launch_ms": sec / nl * 1e3,
                                "covered_ms_per_pass": sum(v[1] for v in fam.values()) * 1e3,
                                "families": {k: {"ms_per_pass": v[1] * 1e3, "launches": v[2], "tflops": v[0] / v[1] / 1e12 if v[1] else 0.0,
                                                 "gbps": v[3] / v[1] / 1e9 if v[1] else 0.0,
                                                 "frac": max(v[0] / MF, v[3] / HB) / v[1] if v[1] else 0.0}
                                             for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
                                "note": "instrumented eager eval pass (HIP events per launch family); post_process = decode + score filter + "
                                        "sorts + batched NMS incl. its two host waits, priced against the decode's bytes only (the NMS is "
                                        "latency-bound integer work)"}
        if not a.no_cpu_baseline:
            # the oracle's detect(): the reference's eval path restated on torch-CPU fp32 incl. the restated torchvision NMS
            from oracle import yolov4 as ov
            sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
            xc = x[: a.cpu_batch].float().cpu()
            torch.set_flush_denormal(True)

            def one():
                with torch.no_grad():
                    ov.detect(sd, xc, ov.CSP53, 80, ov.Cfg(act="mish", drop=(0.1, 7), training=False))
                return a.cpu_batch
            best, trial, host, default = tb.best_threads_run(one, counts=(16, 32, 64))
            t0 = time.perf_counter()
            n = sum(one() for _ in range(2))
            cdt = time.perf_counter() - t0
            torch.set_num_threads(default)
            line["cpu_baseline"] = {"value": n / cdt, "unit": "images/sec", "cores": best, "host_threads": host, "kind": "port",
                                    "threads_tried": {str(k): round(v, 3) for k, v in trial.items()},
                                    "sample": f"oracle yolov4 608^2 eval forward + decode + NMS (torch-CPU fp32), batch {a.cpu_batch}, "
                                              "2 timed iterations"}
        print(json.dumps(line))
        return

    def build():
        return h.models.detection.yolov4(pretrained_backbone=False, num_classes=80)

    def make_batch(rank, dev):
        g = torch.Generator().manual_seed(1 + rank)
        return torch.rand((a.batch, 3, a.size, a.size), generator=g).to(dev), PackedTargets(targets(a.batch, g, dev), dev)

    def loss_of(model, x, t):
        return sum(v.sum() for v in model(x, t).values())

    def cpu_baseline():
        """oracle.yolov4.train_losses (the reference's YOLOv4 restated on torch-CPU fp32) + autograd + the oracle's AdaBelief."""
        from oracle import yolov4 as ov
        from oracle.optim import adabelief_step
        torch.manual_seed(0)
        sd = {k: v.detach().clone() for k, v in build().state_dict().items()}
        keys = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
        g = torch.Generator().manual_seed(0)
        x = torch.rand((a.cpu_batch, 3, a.size, a.size), generator=g)
        t = targets(a.cpu_batch, g, "cpu")
        state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in keys}
        step = [0]

        def one():
            work = dict(sd)
            params = {k: sd[k].detach().requires_grad_(True) for k in keys}
            work.update(params)
            losses, _ = ov.train_losses(work, x, t, ov.CSP53, 80, ov.Cfg(act="mish", drop=(0.1, 7), training=True))
            grads = torch.autograd.grad(sum(v.sum() for v in losses.values()), [params[k] for k in keys], allow_unused=True)
            step[0] += 1
            with torch.no_grad():
                for k, gr in zip(keys, grads):
                    if gr is not None:
                        adabelief_step(sd[k], gr, state[k][0], state[k][1], step[0], 1e-3, 0.95, 0.99, 1e-6, 0.0)
            return a.cpu_batch
        torch.set_flush_denormal(True)
        best, trial, host, default = tb.best_threads_run(one, counts=(16, 32, 64))
        t0 = time.perf_counter()
        n = sum(one() for _ in range(2))
        dt = time.perf_counter() - t0
        torch.set_num_threads(default)
        return {"value": n / dt, "unit": "images/sec", "cores": best, "host_threads": host, "kind": "port",
                "threads_tried": {str(k): round(v, 3) for k, v in trial.items()},
                "sample": f"oracle yolov4 608^2 train step (torch-CPU fp32), batch {a.cpu_batch}, 2 timed iterations"}

    tb.run(a, build, make_batch, loss_of, "images/sec fwd+loss+bwd+AdaBelief, YOLOv4 (CSP-darknet53) bs16/GPU 608^2 nc=80",
           f"yolov4 bf16 train step (fwd + 4 losses + bwd + AdaBelief), synthetic {a.size}^2, bs={a.batch} per MI355X "
           "(BASELINE.json configs[3]), random-init weights, 80 classes", train_gflop_per_img=TRAIN_GFLOP_PER_IMG if a.size == 608 else None,
           cpu_baseline=cpu_baseline, traffic_key="yolov4")


if __name__ == "__main__":
    main()
