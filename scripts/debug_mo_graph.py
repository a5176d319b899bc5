import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import holocron_amd as h
dev = torch.device("cuda:0")
for (cin, cout, K, stride) in [(48, 48, 4, 1), (48, 128, 2, 2), (3, 48, 4, 2)]:
    torch.manual_seed(0)
    m = h.models.MobileOneBlock(cin, cout, K, stride).to(dev).train()
    x = torch.rand((8, cin, 32, 32), device=dev, requires_grad=True)
    r = torch.rand((8, cout, 32 // stride, 32 // stride), device=dev)
    holder = {}
    def step():
        for p in m.parameters(): p.grad = None
        x.grad = None
        out = m(x)
        (out.float() * r).sum().backward()
        holder["out"] = out.detach()
    for _ in range(2): step()
    torch.cuda.synchronize()
    ref = {"out": holder["out"].detach().float().clone(), "dx": x.grad.float().clone(), **{n: p.grad.float().clone() for n, p in m.named_parameters()}}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for it in range(2):
        g.replay(); torch.cuda.synchronize()
        got = {"out": holder["out"].detach().float(), "dx": x.grad.float(), **{n: p.grad.float() for n, p in m.named_parameters()}}
        bad = []
        for k, v in ref.items():
            e = float((got[k] - v).norm() / (v.norm() + 1e-12))
            if not (e < 1e-3): bad.append((k, e))
        print((cin, cout, K, stride), "replay", it, "mismatches:", bad[:8], flush=True)
