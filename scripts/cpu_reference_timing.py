"""Authoring-container measurement for BASELINE.md: the REFERENCE ITSELF (imported from /root/reference through tests/_refshim.py)
next to the oracle port, same repvgg_a0 training step (fp32, CE label_smoothing 0.1, AdaBelief lr 1e-3 betas (0.95, 0.99) eps 1e-6),
same batch and thread count - so that the "port" number bench.py reports on the GPU box is anchored on the reference's own code.
Only runs where /root/reference exists.   usage: python scripts/cpu_reference_timing.py [--batch 32] [--iters 3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    import _refshim
    ref = _refshim.load_reference()
    torch.set_flush_denormal(True)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((a.batch, 3, 224, 224), generator=g)
    t = torch.randint(0, 10, (a.batch,), generator=g)
    torch.manual_seed(0)
    model = ref.models.classification.repvgg_a0(pretrained=False).train()
    opt = ref.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)

    def ref_step():
        opt.zero_grad()
        crit(model(x), t).backward()
        opt.step()
    from oracle import repvgg as orv
    nb, aa, bb = orv.ARCH["repvgg_a0"]
    ch = orv.widths(orv.PLANES, aa, bb)
    sd = orv.init_state(nb, ch, num_classes=10, seed=0)
    ostate = {}

    def port_step():
        orv.train_step(sd, ostate, x, t, nb, ch)
    out = {}
    for name, fn in (("reference", ref_step), ("port", port_step)):
        fn()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        out[name] = a.batch * a.iters / (time.perf_counter() - t0)
    print(f"threads {torch.get_num_threads()} (host {os.cpu_count()}), batch {a.batch}, {a.iters} timed iterations: "
          f"reference {out['reference']:.2f} img/s, oracle port {out['port']:.2f} img/s")


if __name__ == "__main__":
    main()
