"""Per-launch table of one instrumented training step (HIP events on the launch stream around every profiled launch, as in the
benches' roofline objects), grouped by (family, shape): count, total us, TFLOP/s against the dense bf16 MFMA peak, GB/s of algorithmic
bytes.  Tells which layer shapes a model's step spends its time on - the input of every kernel decision in DESIGN.md.

    python scripts/layer_table.py --model yolov4 [--batch 16]      # also: repvgg_a0, rexnet1_0x, mobileone_s0
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov4")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    import holocron_amd as h
    from holocron_amd.ops import conv as cv
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if a.model == "yolov4":
        import bench_yolov4 as by
        from holocron_amd.models.detection.yolov4 import PackedTargets
        batch = a.batch or 16
        m = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev).train()
        g = torch.Generator().manual_seed(0)
        x = torch.rand((batch, 3, 608, 608), generator=g).to(dev)
        tg = PackedTargets(by.targets(batch, g, dev), dev)

        def loss_of():
            return sum(v.sum() for v in m(x, tg).values())
    else:
        batch = a.batch or 256
        nc = 1000 if a.model.startswith("rexnet") else 10
        m = getattr(h.models, a.model)(num_classes=nc).to(dev).train()
        x = torch.rand((batch, 3, 224, 224), device=dev)
        t = torch.randint(0, nc, (batch,), device=dev)

        def loss_of():
            return h.nn.functional.cross_entropy(m(x).float(), t)
    opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0)

    def step():
        opt.zero_grad(set_to_none=True)
        loss_of().backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rows = {}
    total = 0.0
    for _ in range(a.steps):
        cv.PROFILE, cv.PROFILE_TAGS = [], []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
        assert len(cv.PROFILE) == len(cv.PROFILE_TAGS), (len(cv.PROFILE), len(cv.PROFILE_TAGS))
        for (fam, fl, s0, s1, nb), tag in zip(cv.PROFILE, cv.PROFILE_TAGS):
            r = rows.setdefault((fam, tag), [0, 0.0, 0.0, 0.0])
            r[0] += 1
            r[1] += s0.elapsed_time(s1) * 1e3
            r[2] += fl
            r[3] += nb
    cv.PROFILE = cv.PROFILE_TAGS = None
    n = a.steps
    print(f"# {a.model} batch {batch}: instrumented eager step {total / n:.2f} ms; rows = launches per step x avg us (sorted by time per step)")
    print(f"{'family':<16} {'shape':<58} {'n':>4} {'us each':>8} {'us/step':>9} {'TF/s':>7} {'%peak':>6} {'GB/s':>7}")
    fam_tot = {}
    covered = 0.0
    for (fam, tag), (cnt, us, fl, nb) in sorted(rows.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{fam:<16} {tag:<58} {cnt / n:>4.0f} {us / cnt:>8.1f} {us / n:>9.1f} {fl / us / 1e6:>7.0f} {fl / us / 1e6 / 2500 * 100:>6.1f} {nb / us / 1e3:>7.0f}")
    for (fam, tag), (cnt, us, fl, nb) in rows.items():
        f = fam_tot.setdefault(fam, [0, 0.0, 0.0, 0.0])
        f[0] += cnt; f[1] += us; f[2] += fl; f[3] += nb
        covered += us
    print("# families")
    for fam, (cnt, us, fl, nb) in sorted(fam_tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{fam:<16} {'':<58} {cnt / n:>4.0f} {us / cnt:>8.1f} {us / n:>9.1f} {fl / us / 1e6:>7.0f} {fl / us / 1e6 / 2500 * 100:>6.1f} {nb / us / 1e3:>7.0f}")
    print(f"# covered {covered / n / 1e3:.2f} ms of {total / n:.2f} ms")


if __name__ == "__main__":
    main()
