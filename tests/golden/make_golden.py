"""Generate the golden fixtures by running the REFERENCE itself (import of /root/reference through
tests/_refshim.py).  Only runnable in the authoring container; the .pt files it writes are committed
and are what travels to the GPU box.

    python tests/golden/make_golden.py

Inputs/weights that feed bf16 kernels are rounded to bf16-representable fp32 values so that the only
GPU-vs-reference differences are accumulation order and the bf16 rounding of intermediates.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _refshim  # noqa: E402

ref = _refshim.load_reference()
torch.set_num_threads(4)


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rand_boxes(n, g):
    xy = torch.rand((n, 2), generator=g) * 0.6
    wh = torch.rand((n, 2), generator=g) * 0.35 + 0.02
    return torch.cat([xy, xy + wh], dim=1)


def gen_boxes():
    g = torch.Generator().manual_seed(11)
    from ref_holocron.ops import boxes as rb
    b1, b2 = rand_boxes(7, g), rand_boxes(5, g)
    # the reference's own KAT fixture (tests/test_ops.py:9-14)
    kat = torch.tensor([[0, 0, 100, 100], [50, 50, 100, 100], [50, 50, 150, 150], [100, 100, 200, 200]], dtype=torch.float32)
    out = {"b1": b1, "b2": b2, "kat_boxes": kat}
    for tag, (x, y) in {"rand": (b1, b2), "kat": (kat, kat)}.items():
        out[tag] = {
            "iou": rb.box_iou(x, y), "giou": rb.box_giou(x, y), "penalty": rb.iou_penalty(x, y),
            "diou": rb.diou_loss(x, y), "ciou": rb.ciou_loss(x, y), "arc": rb.aspect_ratio_consistency(x, y),
        }
    save("boxes.pt", out)


def gen_functional():
    g = torch.Generator().manual_seed(5)
    Fr = ref.nn.functional
    x = (torch.rand((4, 3, 8, 8), generator=g) * 8 - 4).requires_grad_(True)
    y = Fr.hard_mish(x)
    r = torch.rand(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * r).sum(), x)
    out = {"hard_mish": {"x": x.detach(), "y": y.detach(), "r": r, "dx": gx}}
    cases = []
    for (shape, K, w, ign, gamma, red) in [((6,), 5, False, -100, 2.0, "mean"), ((2, 4, 4), 7, True, 3, 2.0, "mean"),
                                            ((2, 4, 4), 7, True, 3, 1.5, "sum"), ((3, 5), 4, False, -100, 0.0, "none")]:
        xs = (shape[0], K) + tuple(shape[1:])
        x = (torch.randn(xs, generator=g) * 2).requires_grad_(True)
        t = torch.randint(0, K, shape, generator=g)
        weight = torch.rand((K,), generator=g) + 0.5 if w else None
        loss = Fr.focal_loss(x, t, weight, ign, red, gamma)
        rr = torch.rand(loss.shape, generator=g) if red == "none" else torch.tensor(1.0)
        (gx,) = torch.autograd.grad((loss * rr).sum(), x)
        cases.append({"x": x.detach(), "target": t, "weight": weight, "ignore_index": ign, "gamma": gamma,
                      "reduction": red, "loss": loss.detach(), "r": rr, "dx": gx})
    out["focal"] = cases
    save("functional.pt", out)


def gen_optim():
    g = torch.Generator().manual_seed(7)
    out = {"adabelief": [], "lars": []}
    p0 = torch.randn((1000,), generator=g)
    grads = [torch.randn((1000,), generator=g) * (0.5 ** i) for i in range(4)]
    for kw in [dict(lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0, amsgrad=False),
               dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False),
               dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True)]:
        p = torch.nn.Parameter(p0.clone())
        opt = ref.optim.AdaBelief([p], **kw)
        traj = []
        for gr in grads:
            p.grad = gr.clone()
            opt.step()
            traj.append(p.detach().clone())
        st = opt.state[p]
        out["adabelief"].append({"kw": kw, "p0": p0, "grads": grads, "traj": traj, "exp_avg": st["exp_avg"].clone(),
                                 "exp_avg_sq": st["exp_avg_sq"].clone()})
    for kw in [dict(lr=1e-2, momentum=0.0, weight_decay=0.0), dict(lr=1e-2, momentum=0.9, weight_decay=1e-3),
               dict(lr=5e-3, momentum=0.9, weight_decay=1e-3, nesterov=True), dict(lr=1e-2, momentum=0.9, dampening=0.1)]:
        p = torch.nn.Parameter(p0.clone())
        opt = ref.optim.LARS([p], **kw)
        traj, gafter = [], []
        for gr in grads:
            p.grad = gr.clone()
            opt.step()
            traj.append(p.detach().clone())
            gafter.append(p.grad.clone())
        out["lars"].append({"kw": kw, "p0": p0, "grads": grads, "traj": traj, "grad_after": gafter})
    save("optim.pt", out)


def _randomize_bn(m, g):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = torch.rand(mod.weight.shape, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.2
            mod.running_mean.data = torch.randn(mod.bias.shape, generator=g) * 0.1
            mod.running_var.data = torch.rand(mod.bias.shape, generator=g) + 0.5


def gen_repblock():
    g = torch.Generator().manual_seed(3)
    RepBlock = ref.models.classification.repvgg.RepBlock
    cases = []
    for (cin, cout, stride, ident, hw) in [(16, 16, 1, True, 12), (16, 32, 2, False, 12), (48, 48, 1, True, 9),
                                           (32, 64, 2, False, 11), (3, 16, 2, False, 16)]:
        torch.manual_seed(cin * 100 + cout)
        blk = RepBlock(cin, cout, stride, ident)
        ref.nn.init.init_module(blk, "relu")
        _randomize_bn(blk, g)
        for p in blk.parameters():
            if p.dim() == 4:
                p.data = bf16r(p.data)
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = bf16r(torch.randn((2, cin, hw, hw), generator=g)).requires_grad_(True)
        blk.train()
        out = blk(x)
        r = bf16r(torch.randn(out.shape, generator=g))
        params = [p for p in blk.parameters()]
        grads = torch.autograd.grad((out * r).sum(), [x] + params)
        names = [n for n, _ in blk.named_parameters()]
        sd1 = {k: v.clone() for k, v in blk.state_dict().items()}
        blk.eval()
        with torch.no_grad():
            out_eval = blk(x)
            blk.reparametrize()
            out_rep = blk(x)
        cases.append({"cfg": (cin, cout, stride, ident), "state": sd0, "x": x.detach(), "r": r, "out": out.detach(),
                      "dx": grads[0], "dparams": dict(zip(names, grads[1:])), "state_after": sd1,
                      "out_eval": out_eval, "out_rep": out_rep,
                      "rep_weight": blk.branches.weight.detach().clone(), "rep_bias": blk.branches.bias.detach().clone()})
    save("repblock.pt", cases)


def gen_repvgg_small():
    g = torch.Generator().manual_seed(9)
    torch.manual_seed(21)
    RepVGG = ref.models.classification.repvgg.RepVGG
    cfg = dict(num_blocks=[1, 1, 2, 1, 1], planes=[16, 16, 32, 64, 64], width_multiplier=1, final_width_multiplier=1,
               num_classes=10)
    m = RepVGG(**cfg)
    _randomize_bn(m, g)
    for p in m.parameters():
        if p.dim() == 4:
            p.data = bf16r(p.data)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = bf16r(torch.rand((4, 3, 64, 64), generator=g))
    t = torch.randint(0, 10, (4,), generator=g)
    m.train()
    opt = ref.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    steps = []
    for _ in range(2):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, t)
        loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        opt.step()
        steps.append({"logits": logits.detach().clone(), "loss": loss.detach().clone(), "grads": grads,
                      "state_after": {k: v.clone() for k, v in m.state_dict().items()}})
    m.eval()
    with torch.no_grad():
        ev = m(x)
        m.reparametrize()
        ev_rep = m(x)
    save("repvgg_small.pt", {"cfg": cfg, "state": sd0, "x": x, "target": t, "steps": steps, "eval_logits": ev,
                             "eval_logits_rep": ev_rep})


def gen_darknet():
    """DarkNet ResBlock unit + two training steps of a small DarknetV3 (reference modules)."""
    g = torch.Generator().manual_seed(17)
    dk = ref.models.classification.darknetv3
    cases = []
    for (planes, hw) in [(32, 10), (64, 7)]:
        torch.manual_seed(planes)
        blk = dk.ResBlock(planes, planes // 2, torch.nn.LeakyReLU(0.1, inplace=True), torch.nn.BatchNorm2d)
        ref.nn.init.init_module(blk, "leaky_relu")
        _randomize_bn(blk, g)
        for p in blk.parameters():
            if p.dim() == 4:
                p.data = bf16r(p.data)
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = bf16r(torch.randn((2, planes, hw, hw), generator=g)).requires_grad_(True)
        blk.train()
        out = blk(x)
        r = bf16r(torch.randn(out.shape, generator=g))
        params = list(blk.parameters())
        grads = torch.autograd.grad((out * r).sum(), [x] + params)
        cases.append({"planes": planes, "state": sd0, "x": x.detach(), "r": r, "out": out.detach(), "dx": grads[0],
                      "dparams": dict(zip([n for n, _ in blk.named_parameters()], grads[1:])),
                      "state_after": {k: v.clone() for k, v in blk.state_dict().items()}})
    torch.manual_seed(33)
    layout = [(32, 1), (64, 2)]
    m = dk.DarknetV3(layout, num_classes=10, stem_channels=16)
    _randomize_bn(m, g)
    for p in m.parameters():
        if p.dim() == 4:
            p.data = bf16r(p.data)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = bf16r(torch.rand((4, 3, 32, 32), generator=g))
    t = torch.randint(0, 10, (4,), generator=g)
    m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, t)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    save("darknet.pt", {"resblocks": cases, "layout": layout, "stem": 16, "state": sd0, "x": x, "target": t,
                        "logits": logits.detach(), "loss": loss.detach(), "grads": grads,
                        "state_after": {k: v.clone() for k, v in m.state_dict().items()}})


def gen_nms():
    """torchvision.ops.nms is absent: these vectors come from the restated algorithm (oracle/tv_ops.py),
    plus the two situations the reference's own tests pin (tests/test_models_detection.py:158-163: disjoint
    boxes all kept in score order; :229-233: identical boxes -> one kept)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.tv_ops import nms
    g = torch.Generator().manual_seed(13)
    cases = []
    grid = torch.tensor([[i * 10.0, j * 10.0, i * 10.0 + 5, j * 10.0 + 5] for i in range(7) for j in range(7)])
    sc = torch.rand((49,), generator=g)
    cases.append({"boxes": grid, "scores": sc, "thr": 0.7, "keep": nms(grid, sc, 0.7), "pinned_by": "reference test (disjoint)"})
    same = torch.tensor([[0.2, 0.2, 0.6, 0.6]]).repeat(9, 1)
    sc = torch.full((9,), 0.25)
    cases.append({"boxes": same, "scores": sc, "thr": 0.7, "keep": nms(same, sc, 0.7), "pinned_by": "reference test (identical)"})
    for n, thr in [(200, 0.5), (777, 0.7), (1500, 0.3)]:
        b = rand_boxes(n, g)
        sc = (torch.rand((n,), generator=g) * 20).round() / 20      # many exact ties
        cases.append({"boxes": b, "scores": sc, "thr": thr, "keep": nms(b, sc, thr), "pinned_by": "restatement"})
    save("nms.pt", cases)


if __name__ == "__main__":
    gen_boxes()
    gen_functional()
    gen_optim()
    gen_repblock()
    gen_repvgg_small()
    gen_darknet()
    gen_nms()
