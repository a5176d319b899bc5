"""Debug helper: intermediate tensors of one RepBlock step, HIP vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import holocron_amd as h
from holocron_amd.nn import repblock_op as ro
from holocron_amd.ops import conv as cv

cases = torch.load("tests/golden/repblock.pt", weights_only=False)
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = cases[idx]
cin, cout, stride, ident = c["cfg"]
print("cfg", c["cfg"], "x", tuple(c["x"].shape))
sd = c["state"]
x = c["x"].clone().requires_grad_(True)
w3 = sd["branches.0.0.weight"].clone().requires_grad_(True); w1 = sd["branches.1.0.weight"].clone().requires_grad_(True)
y3 = F.conv2d(x, w3, None, stride, 1); y1 = F.conv2d(x, w1, None, stride, 0)
y3.retain_grad(); y1.retain_grad()
def bn(y, p):
    return F.batch_norm(y, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
z = bn(y3, "branches.0.1") + bn(y1, "branches.1.1")
if ident: z = z + bn(x, "branches.2")
out = F.relu(z)
(out * c["r"]).sum().backward()
print("oracle out vs golden", (out - c["out"]).abs().max().item())

def rel(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
blk = h.models.RepBlock(cin, cout, stride, ident); blk.load_state_dict(sd); blk = blk.cuda().train()
# monkeypatch to capture intermediates
saved = {}
orig_bwd_apply = None
xg = c["x"].cuda().requires_grad_(True)
o = blk(xg)
print("out rel", rel(o.float().cpu(), out))
fn = o.grad_fn
src, gy3, gy1, gout, save, *_ = fn.saved_tensors
print("y3 rel", rel(gy3.float().cpu(), y3), "y1 rel", rel(gy1.float().cpu(), y1))
mean3 = y3.mean((0,2,3)); var3 = y3.var((0,2,3), unbiased=False)
print("save mean3 err", (save[0].cpu()-mean3).abs().max().item(), "invstd3 rel", rel(save[1].cpu(), (var3+1e-5).rsqrt()))
(o.float() * c["r"].cuda()).sum().backward()
e = (xg.grad.float().cpu() - x.grad)
print("dx rel", rel(xg.grad.float().cpu(), x.grad), "max abs err", e.abs().max().item(), "ref rms", x.grad.pow(2).mean().sqrt().item())
print("err rms per channel", e.pow(2).mean((0,2,3)).sqrt()[:8])
print("err rms per row", e.pow(2).mean((0,1,3)).sqrt())
print("err rms per col", e.pow(2).mean((0,1,2)).sqrt())
print("dw3 rel", rel(blk.branches[0][0].weight.grad.cpu(), w3.grad), "dw1 rel", rel(blk.branches[1][0].weight.grad.cpu(), w1.grad))
# direct check: recompute dy3/dy1 on GPU pieces
lib = h._lib.load()
