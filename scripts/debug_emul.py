import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import holocron_amd as h
from oracle import repvgg as orv
def rel(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
cases = torch.load("tests/golden/repblock.pt", weights_only=False)
for c in cases:
    cin, cout, stride, ident = c["cfg"]
    sd = {"blk." + k: v.clone() for k, v in c["state"].items()}
    x = c["x"]
    with torch.no_grad():
        e = orv.rep_block_bf16(orv.bf16r(x), sd, "blk", stride, ident, True)
    blk = h.models.RepBlock(cin, cout, stride, ident); blk.load_state_dict(c["state"]); blk = blk.cuda().train()
    with torch.no_grad():
        o = blk(x.cuda()).float().cpu()
    d = (o - e).abs()
    print(c["cfg"], "rel vs emul", rel(o, e), "rel vs fp32 ref", rel(o, c["out"]), "n_diff", int((d > 0).sum()), "/", d.numel(), "max", float(d.max()))
g = torch.load("tests/golden/repvgg_small.pt", weights_only=False)
cfg = g["cfg"]; ch = orv.widths(cfg["planes"], 1, 1)
sd = {k: v.clone() for k, v in g["state"].items()}
taps = {}
with torch.no_grad():
    el = orv.forward(sd, g["x"], cfg["num_blocks"], ch, training=True, taps=taps, emulate_bf16=True)
m = h.models.RepVGG(**cfg); m.load_state_dict(g["state"]); m = m.cuda().train()
xx = g["x"].cuda()
with torch.no_grad():
    for si, stage in enumerate(m.features):
        for bi, blk in enumerate(stage):
            xx = blk(xx)
            k = f"features.{si}.{bi}"
            d = (xx.float().cpu() - taps[k]).abs()
            print(k, tuple(xx.shape), "rel", rel(xx.float().cpu(), taps[k]), "ndiff", int((d > 0).sum()), "/", d.numel(), "max", float(d.max()))
