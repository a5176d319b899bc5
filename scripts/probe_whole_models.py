"""Measurement behind the whole-model parity bounds (tests/test_gpu_rexnet.py, tests/test_gpu_mobileone.py): the HIP model, the
bf16-emulating oracle and the fp32 reference (tests/golden/whole_models.pt) on the 16 x 128 x 128 fixture."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2  # noqa: E402

import holocron_amd as h  # noqa: E402
from oracle import mobileone as omo, rexnet as orx  # noqa: E402

g = torch.load(os.path.join(ROOT, "tests", "golden", "whole_models.pt"))
for name in ("rexnet1_0x", "mobileone_s0"):
    gm = g[name]
    x = (gm["x8"].float() / 255.0).to(torch.bfloat16).float()
    torch.manual_seed(gm["seed"])
    m = getattr(h.models, name)(num_classes=10, **gm["kwargs"])
    names = [n for n, _ in m.named_parameters()]
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    leaves = [sd[n].requires_grad_(True) for n in names]
    lo = orx.forward(sd, x, training=True, emulate_bf16=True) if name.startswith("rex") else omo.forward(sd, x, training=True, emu=True)
    loss_o = torch.nn.functional.cross_entropy(lo, gm["target"])
    go = dict(zip(names, torch.autograd.grad(loss_o, leaves, allow_unused=True)))
    m = m.cuda().train()
    lg = m(x.cuda())
    loss_g = torch.nn.functional.cross_entropy(lg.float(), gm["target"].cuda())
    loss_g.backward()
    gg = {n: p.grad.float().cpu() for n, p in m.named_parameters()}
    print(name, "logits hip-vs-emu", rel_l2(lg.float().cpu(), lo.detach()), "hip-vs-ref", rel_l2(lg.float().cpu(), gm["logits"]),
          "emu-vs-ref", rel_l2(lo.detach(), gm["logits"]), "loss", float(loss_g), float(loss_o), float(gm["loss"]))
    errs = sorted(((rel_l2(gg[n], go[n]), n) for n in names if gm["grad_abs_max"][n] > 1e-6 and go[n] is not None), reverse=True)
    print("  grads hip-vs-emu: worst", [(round(a, 4), b) for a, b in errs[:8]], "median", round(errs[len(errs) // 2][0], 4),
          "frac<6e-2", sum(1 for a, _ in errs if a < 6e-2) / len(errs), "n", len(errs))
    cos = sorted(((float(torch.nn.functional.cosine_similarity(gg[n].flatten(), go[n].flatten(), dim=0)), n) for n in names
                  if gm["grad_abs_max"][n] > 1e-6 and go[n] is not None))
    print("  cosine worst", [(round(a, 4), b) for a, b in cos[:5]])
    nr = sorted(((abs(float(gg[n].norm()) - gm["grad_norms"][n]) / (gm["grad_norms"][n] + 1e-12), n) for n in names if gm["grad_abs_max"][n] > 1e-6), reverse=True)
    print("  grad-norm hip-vs-ref worst", [(round(a, 3), b) for a, b in nr[:5]], "median", round(nr[len(nr) // 2][0], 4))
