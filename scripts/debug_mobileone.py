import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import holocron_amd as h
from oracle import mobileone as omo
g = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mobileone.pt"))
for c in g["blocks"]:
    cin, cout, K, stride = c["cfg"]
    blk = h.models.MobileOneBlock(cin, cout, K, stride); blk.load_state_dict(c["state"]); blk = blk.cuda().train()
    x = c["x"].cuda().requires_grad_(True)
    out = blk(x); (out.float() * c["r"].cuda()).sum().backward()
    sd = {"b." + k: v.clone() for k, v in c["state"].items()}
    names = list(c["dparams"]); leaves = [sd["b." + n].requires_grad_(True) for n in names]
    xe = c["x"].clone().requires_grad_(True)
    oe = omo.block(xe, sd, "b", stride, True, emu=True)
    ge = torch.autograd.grad((oe * c["r"]).sum(), [xe] + leaves)
    params = dict(blk.named_parameters())
    for n, gg in zip(names, ge[1:]):
        if gg.dim() == 4 and tuple(gg.shape[1:]) == (1, 1, 1):
            got = params[n].grad.float().cpu().flatten(); w = c["state"][n].flatten()
            i = int(got.abs().argmax())
            print(c["cfg"], n, "max|got|", float(got.abs().max()), "emu", float(gg.abs().max()), "ref", float(c["dparams"][n].abs().max()),
                  "w at argmax", float(w[i]), "emu there", float(gg.flatten()[i]), "min|w|", float(w.abs().min()))
