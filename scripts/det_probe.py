"""Locate run-to-run differences in deterministic mode: two identical fwd+bwd passes of repvgg_a0 bs256, per-module outputs and
per-parameter gradients compared bitwise."""
import sys, torch
sys.path.insert(0, ".")
import holocron_amd as h

dev = torch.device("cuda:0")
h.set_deterministic(True)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand((256, 3, 224, 224), device=dev, generator=g)
t = torch.randint(0, 10, (256,), device=dev, generator=g)
torch.manual_seed(0)
m = h.models.repvgg_a0(num_classes=10).to(dev).train()
state = {k: v.clone() for k, v in m.state_dict().items()}


def run():
    m.load_state_dict(state)
    outs = {}
    hooks = []
    for n, mod in m.named_modules():
        if n and len(list(mod.children())) == 0 or type(mod).__name__ == "RepBlock":
            hooks.append(mod.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o.detach().float().clone()) if torch.is_tensor(o) else None))
    m.zero_grad(set_to_none=True)
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, t, label_smoothing=0.1)
    loss.backward()
    for hk in hooks:
        hk.remove()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in m.named_buffers()}
    torch.cuda.synchronize()
    return outs, grads, bufs, loss.detach().clone()


a = run()
b = run()
print("loss", a[3].item(), b[3].item(), torch.equal(a[3], b[3]))
for what, da, db in (("out", a[0], b[0]), ("grad", a[1], b[1]), ("buf", a[2], b[2])):
    bad = [(n, (da[n].float() - db[n].float()).abs().max().item()) for n in da if not torch.equal(da[n], db[n])]
    print(what, "differing:", len(bad), "of", len(da))
    for n, e in bad[:12]:
        print("   ", n, e)
    if what == "grad" and bad:
        print("    last:", bad[-6:])
