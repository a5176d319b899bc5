"""Where do the small fill / device-to-device copy launches of a RepVGG-A0 training step come from?  One eager step under
torch.profiler with Python stacks; prints every aten::copy_ / zero_ / fill_ / zeros / clone call site (first holocron_amd / bench
frame of its stack) with its count and device time."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import holocron_amd as h  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = h.models.repvgg_a0(num_classes=10).to(dev).train()
opt = h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)
x = torch.rand((256, 3, 224, 224), device=dev)
t = torch.randint(0, 10, (256,), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(x), t, label_smoothing=0.1)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::clone", "aten::zeros_like", "aten::ones_like", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::add_", "aten::mul_")
sites = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in WANT or ev.device_time_total <= 0:
        continue
    where = "?"
    for fr in ev.stack:
        if ("holocron_amd" in fr or "find_small" in fr) and "torch/" not in fr:
            where = fr.replace(ROOT + "/", "")
            break
    if where == "?" and ev.stack:
        where = "autograd/engine: " + ev.stack[0][-80:]
    k = (ev.name, where)
    sites[k][0] += 1
    sites[k][1] += ev.device_time_total
for (name, where), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {us / max(n, 1):7.1f} us  {name:18s} {where}")
