"""Where do the small torch kernels of a training step come from?  One eager step (repvgg_a0 by default, `yolov4` as first argument for
the YOLOv4 608^2 batch-16 step, `yolov4_eval` for one eval pass) under torch.profiler with Python stacks: every op that launches a fill / copy / torch elementwise
kernel, grouped by its innermost holocron_amd / bench frame."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import holocron_amd as h
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "repvgg_a0"
loss_buf = torch.zeros((), device=dev)
if which == "yolov4":
    import bench_yolov4 as by_
    from holocron_amd.models.detection.yolov4 import PackedTargets
    model = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev).train()
    g_ = torch.Generator().manual_seed(0)
    x = torch.rand((16, 3, 608, 608), generator=g_).to(dev)
    tg = PackedTargets(by_.targets(16, g_, dev), dev)

    def loss_of():
        return sum(v.sum() for v in model(x, tg).values())
elif which == "yolov4_eval":
    model = h.models.detection.yolov4(pretrained_backbone=False, num_classes=80).to(dev).eval()
    g_ = torch.Generator().manual_seed(0)
    x = torch.rand((16, 3, 608, 608), generator=g_).to(dev)
else:
    nc_ = 1000 if which.startswith("rexnet") else 10
    model = getattr(h.models, which)(num_classes=nc_).to(dev).train()
    x = torch.rand((256, 3, 224, 224), device=dev)
    t = torch.randint(0, nc_, (256,), device=dev)

    def loss_of():
        return h.nn.functional.cross_entropy(model(x), t, label_smoothing=0.1)
opt = None if which == "yolov4_eval" else h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)


def step():
    if which == "yolov4_eval":
        with torch.no_grad():
            model(x)
        return
    opt.zero_grad(set_to_none=True)
    loss = loss_of()
    loss.backward()
    loss_buf.copy_(loss.detach())
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by = collections.Counter()
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == "CUDA" if hasattr(ev.device_type, "name") else False:
        continue
    ks = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not ks:
        continue
    names = ",".join(sorted({k.name[:48] for k in ks}))
    if not any(s in names for s in ("Fill", "copy", "Memcpy", "Memset", "elementwise", "reduce_kernel", "at::")):
        continue
    where = "?"
    for fr in ev.stack or []:
        if root in fr and "site-packages" not in fr and "dist-packages" not in fr:
            where = fr.replace(root + "/", "")
            break
    if where == "?":
        # no Python stack (ops issued from inside the autograd engine): name the enclosing profiler ranges instead
        chain, par = [], getattr(ev, "cpu_parent", None)
        while par is not None and len(chain) < 4:
            chain.append(par.name[:60])
            par = getattr(par, "cpu_parent", None)
        where = " < ".join(chain) if chain else "?"
    by[(ev.name, names, where)] += 1
for (name, names, where), n in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {names:50s} {where}")
