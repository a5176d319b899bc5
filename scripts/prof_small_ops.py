"""Which aten ops launch the small torch kernels (fill / copy / neg / add / reduce) of an eager headline step, by phase of the step
and by input shape (torch.profiler; kernels are attributed to the innermost aten op that launched them)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function

import holocron_amd as h

dev = torch.device("cuda:0")
torch.manual_seed(0)
MODEL = os.environ.get("MODEL", "repvgg_a0")
USE_HC_LOSS = os.environ.get("HC_LOSS", "1") == "1"
if MODEL == "yolov4":
    from holocron_amd.models.detection.yolov4 import PackedTargets
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_yolov4 import targets
    m = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand((16, 3, 608, 608), generator=g).to(dev)
    t = PackedTargets(targets(16, g, dev), dev)
else:
    m = getattr(h.models, MODEL)(num_classes=10 if MODEL.startswith("repvgg") else 1000).to(dev).train()
    x = torch.rand(256, 3, 224, 224, device=dev)
    t = torch.randint(0, 10, (256,), device=dev)
opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6)
loss_buf = torch.zeros((), device=dev)


def step():
    with record_function("PH_zero"):
        opt.zero_grad(set_to_none=True)
    with record_function("PH_fwd"):
        logits = m(x, t) if MODEL == "yolov4" else m(x)
    with record_function("PH_loss"):
        if MODEL == "yolov4":
            loss = sum(v.sum() for v in logits.values())
        elif USE_HC_LOSS:
            loss = h.nn.functional.cross_entropy(logits, t, label_smoothing=0.1)
        else:
            loss = torch.nn.functional.cross_entropy(logits.float(), t, label_smoothing=0.1)
    with record_function("PH_bwd"):
        loss.backward()
    with record_function("PH_copy"):
        loss_buf.copy_(loss.detach())
    with record_function("PH_opt"):
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
NSTEP = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(NSTEP):
        step()
    torch.cuda.synchronize()
evs = list(prof.events())
phases = sorted([(e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.startswith("PH_")])


def phase_of(e):
    for s, t_, n in phases:
        if s <= e.time_range.start <= t_:
            return n
    return "autograd-thread"      # backward nodes run on the engine's thread: their ops start inside PH_bwd's wall time anyway


agg, tim = collections.Counter(), collections.Counter()
nk = 0
for ev in evs:
    if not ev.kernels or ev.name.startswith("PH_"):
        continue
    nk += len(ev.kernels)
    kn = ",".join(sorted({k.name.replace("void ", "").replace("at::native::", "").split("(")[0][:42] for k in ev.kernels}))
    key = (phase_of(ev), ev.name[:40], str(ev.input_shapes)[:60] if os.environ.get("SHAPES", "1") == "1" else "", kn[:90])
    agg[key] += len(ev.kernels)
    tim[key] += sum(k.duration for k in ev.kernels)
print(f"kernel launches per step: {nk / NSTEP:.1f}")
print(f"{'phase':<16} {'op':<40} {'launches/step':>13} {'us/step':>9}  shapes | kernels")
for key, n in sorted(agg.items(), key=lambda kv: (kv[0][0], -tim[kv[0]])):
    hc = not ("at::" in key[3] or "elementwise" in key[3] or "Memcpy" in key[3] or "Memset" in key[3] or "reduce_kernel" in key[3] or "Cijk" in key[3]
              or "softmax" in key[3] or "nll" in key[3])
    if hc and os.environ.get("ALL", "0") != "1":
        continue
    print(f"{key[0]:<16} {key[1]:<40} {n / NSTEP:>13.1f} {tim[key] / NSTEP:>9.1f}  {key[2]} | {key[3]}")
