"""One training step of repvgg_a0 at odd batch / image sizes with the row-unit kernels on and off (separate processes via env):
prints loss and a few gradient norms so that the two runs can be compared."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import holocron_amd as h

for (bs, size) in [(3, 224), (8, 192), (5, 448), (2, 112)]:
    torch.manual_seed(0)
    m = h.models.repvgg_a0(num_classes=10).cuda().train()
    x = torch.rand(bs, 3, size, size, device="cuda")
    t = torch.randint(0, 10, (bs,), device="cuda")
    loss = torch.nn.functional.cross_entropy(m(x), t)
    loss.backward()
    torch.cuda.synchronize()
    gs = [float(p.grad.norm()) for n, p in m.named_parameters() if n.endswith("branches.0.0.weight")]
    print(f"bs={bs} size={size} loss={float(loss):.4f} gnorm first/mid/last={gs[0]:.4f} {gs[len(gs) // 2]:.4f} {gs[-1]:.4f}", flush=True)
