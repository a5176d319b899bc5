import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
import holocron_amd as h
from holocron_amd.nn import functional as Fh
from holocron_amd.ops.conv import to_cl_bf16
from oracle import yolo as oy

def bf(t): return t.to(torch.bfloat16).float()
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-12))

g = torch.Generator().manual_seed(8)
N, H, W, nc = 2, 76, 76, 80
anchors = torch.tensor([[12, 16], [19, 36], [40, 28]], dtype=torch.float32) / 608
x = bf(torch.randn((N, 255, H, W), generator=g))
tg = []
for k in (5, 8):
    b = torch.rand((k, 4), generator=g); b[:, :2] *= b[:, 2:]; b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.02).clamp(max=0.999)
    tg.append({"boxes": b, "labels": torch.randint(0, nc, (k,), generator=g)})
xr = x.clone().requires_grad_(True)
ref = oy.compute_losses(xr, tg, anchors, nc, 1.2)
(dref,) = torch.autograd.grad(sum(v.sum() for v in ref.values()), xr)
layer = h.models.detection.YoloLayer(anchors.clone(), num_classes=nc, scale_xy=1.2).cuda().train()
xp = to_cl_bf16(torch.cat([x, torch.zeros((N, 1, H, W))], 1).cuda()).requires_grad_(True)
losses = layer(xp, [{k: v.cuda() for k, v in t.items()} for t in tg])
print({k: (float(v.sum()), float(ref[k].sum())) for k, v in losses.items()})
sum(v.sum() for v in losses.values()).backward()
got = xp.grad.float().cpu()
d = (got[:, :255] - dref).abs()
print("grad maxabs", float(dref.abs().max()), "maxdiff", float(d.max()), "rel", rel(got[:, :255], dref))
bad = (d > 1.6e-2 * dref.abs() + 1e-7)
print("bad", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()[:10]
for i in idx: print(i.tolist(), float(got[tuple(i)]), float(dref[tuple(i)]))

gm = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "yolo.pt"))["model"]
from oracle import yolov4 as ov
from test_oracle_golden import _yolo_golden_state
m = _yolo_golden_state(gm)
sd = {k: v.clone() for k, v in m.state_dict().items()}
names = [n for n, _ in m.named_parameters()]
leaves = {n: sd[n].requires_grad_(True) for n in names}
cfg = ov.Cfg(act="mish", drop=(gm["drop_p"], 7), noise=gm["noise"], training=True, emulate_bf16=True)
elosses, elogits = ov.train_losses(sd, gm["x"], gm["target"], gm["layout"], gm["num_classes"], cfg)
egrads = dict(zip(names, torch.autograd.grad(sum(v.sum() for v in elosses.values()), list(leaves.values()))))
for mod in m.modules():
    if isinstance(mod, h.nn.DropBlock2d): mod.p = gm["drop_p"]
m = m.cuda().train()
draws = list(gm["noise"])
Fh._noise = lambda shape, device: draws.pop(0).to(device)
losses = m(gm["x"].cuda(), [{k: v.cuda() for k, v in t.items()} for t in gm["target"]])
print({k: (float(v.sum()), float(elosses[k].sum()), float(gm["losses"][k].sum())) for k, v in losses.items()})
sum(v.sum() for v in losses.values()).backward()
params = dict(m.named_parameters())
worst = []
for n in names:
    got = params[n].grad.float().cpu(); e = egrads[n]
    worst.append((float(F.cosine_similarity(got.flatten(), e.flatten(), dim=0)), rel(got, e), n))
worst.sort()
for w in worst[:25]: print(w)
print("median cos", worst[len(worst)//2][0])
for n, v in gm["running"].items():
    if n.endswith("running_mean") and ("stages.4" in n or "head3" in n or "stem" in n): print(n, "vs emu", round(rel(m.state_dict()[n].cpu(), sd[n].detach()), 5))
