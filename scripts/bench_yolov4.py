"""Secondary measurement: YOLOv4 (CSP-Darknet-53 + PAN/SPP neck + head, nc=80) training step
at 608 x 608 on one MI355X: forward + four-part loss + backward + AdaBelief, synthetic data.  Prints one JSON line.

    python scripts/bench_yolov4.py --batch 16 --steps 5 --warmup 2 [--eval]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import holocron_amd as h  # noqa: E402

TRAIN_GFLOP_PER_IMG = 385.0   # 3 x 2 x 64.195 GMAC (fwd + dgrad + wgrad) at 608^2, nc=80


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--eval", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((a.batch, 3, a.size, a.size), generator=g).to(dev)
    target = []
    for i in range(a.batch):
        k = 1 + i % 8
        b = torch.rand((k, 4), generator=g)
        b[:, :2] *= b[:, 2:]
        b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.02).clamp(max=0.999)
        target.append({"boxes": b.to(dev), "labels": torch.randint(0, 80, (k,), generator=g).to(dev)})
    if a.eval:
        m.eval()

        def step():
            with torch.no_grad():
                return m(x)
    else:
        m.train()
        opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0)

        def step():
            opt.zero_grad(set_to_none=True)
            losses = m(x, target)
            sum(v.sum() for v in losses.values()).backward()
            opt.step()
            return losses
    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    rec = {"metric": "images/sec " + ("eval fwd+decode+NMS" if a.eval else "train step (fwd+loss+bwd+AdaBelief)") + ", YOLOv4 608^2 nc=80",
           "value": a.batch / dt, "unit": "img/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3,
           "dtype": "bf16", "data": "synthetic", "config": {"workload": f"yolov4 {a.size}^2 bs{a.batch}"},
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}
    if not a.eval:
        rec["mfma_frac"] = TRAIN_GFLOP_PER_IMG * 1e9 * a.batch / dt / 2.5e15 if a.size == 608 else None
        rec["losses"] = {k: float(v.sum()) for k, v in out.items()}
    else:
        rec["detections_img0"] = int(out[0]["boxes"].shape[0])
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
