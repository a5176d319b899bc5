import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
import holocron_amd as h
from holocron_amd import _lib
lib = _lib.load()
# wrap every entry point with a trace print
for name in list(_lib.SIGNATURES):
    fn = getattr(lib, name)
    def mk(fn, name):
        def w(*a):
            print("CALL", name, flush=True)
            r = fn(*a)
            torch.cuda.synchronize()
            print("  ok", name, r, flush=True)
            return r
        return w
    setattr(lib, name, mk(fn, name))
g = torch.load("tests/golden/darknet.pt", weights_only=False)
c = g["resblocks"][0]
planes = c["planes"]
blk = h.models.ResBlock(planes, planes // 2, torch.nn.LeakyReLU(0.1, inplace=True), torch.nn.BatchNorm2d)
blk.load_state_dict(c["state"]); blk = blk.cuda().train()
x = c["x"].cuda().requires_grad_(True)
out = blk(x)
print("fwd done", flush=True)
(out.float() * c["r"].cuda()).sum().backward()
print("bwd done")
