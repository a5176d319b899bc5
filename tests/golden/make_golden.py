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

from _inputs import optim2_inputs, yolo12_image  # noqa: E402

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


def gen_losses():
    """poly-1 / dice losses and DropBlock of the reference (functional.py:465-613).  DropBlock's uniform noise
    is captured (torch.rand is wrapped while the reference runs) so that the same noise can be replayed."""
    g = torch.Generator().manual_seed(21)
    Fr = ref.nn.functional
    poly = []
    for (shape, K, soft, w, ign, eps, red) in [
            ((6,), 5, False, False, -100, 2.0, "mean"), ((2, 4, 4), 7, False, True, 3, 2.0, "mean"),
            ((2, 4, 4), 7, False, True, 3, 1.0, "sum"), ((3, 5), 4, False, False, -100, 2.0, "none"),
            ((8,), 6, True, False, -100, 2.0, "mean"), ((8,), 6, True, True, 2, 2.0, "mean"),
            ((8,), 6, True, True, 2, 0.5, "sum"), ((8,), 6, True, False, 1, 2.0, "none"),
            ((2, 3, 3), 5, True, False, 4, 2.0, "mean")]:
        xs = (shape[0], K) + tuple(shape[1:])
        x = (torch.randn(xs, generator=g) * 2).requires_grad_(True)
        if soft:
            t = torch.softmax(torch.randn(xs, generator=g), dim=1)
        else:
            t = torch.randint(0, K, shape, generator=g)
        weight = torch.rand((K,), generator=g) + 0.5 if w else None
        loss = Fr.poly_loss(x, t, eps, weight, ign, red)
        rr = torch.rand(loss.shape, generator=g) if red == "none" else torch.tensor(1.0)
        (gx,) = torch.autograd.grad((loss * rr).sum(), x)
        poly.append({"x": x.detach(), "target": t, "weight": weight, "ignore_index": ign, "eps": eps, "reduction": red,
                     "loss": loss.detach(), "r": rr, "dx": gx})
    dice = []
    for (xs, w, gamma, eps) in [((2, 4, 9, 9), False, 1.0, 1e-8), ((3, 5, 16), True, 2.0, 1e-8), ((2, 3, 4, 4), True, 0.5, 1e-3),
                                ((1, 2, 50), False, 1.0, 1e-8)]:
        x = torch.softmax(torch.randn(xs, generator=g), dim=1).requires_grad_(True)
        t = torch.zeros(xs).scatter_(1, torch.randint(0, xs[1], (xs[0], 1) + tuple(xs[2:]), generator=g), 1.0)
        weight = torch.rand((xs[1],), generator=g) + 0.5 if w else None
        loss = Fr.dice_loss(x, t, weight, gamma, eps)
        (gx,) = torch.autograd.grad(loss, x)
        dice.append({"x": x.detach(), "target": t, "weight": weight, "gamma": gamma, "eps": eps, "loss": loss.detach(), "dx": gx})
    drop = []
    real_rand = torch.rand
    for (xs, p, bs, inplace) in [((2, 3, 16, 16), 0.2 * 9, 3, False), ((2, 8, 19, 19), 0.1 * 49 * 4, 7, True),
                                 ((1, 4, 8, 8), 0.5 * 25, 5, False), ((2, 2, 6, 6), 1e-9, 3, False), ((1, 2, 5, 5), 9.0, 3, False)]:
        x = torch.randn(xs, generator=g)
        noise = real_rand((xs[0],) + tuple(xs[2:]), generator=g)
        torch.rand = lambda *a, **k: noise.clone()
        try:
            xin = x.clone().requires_grad_(True)
            y = Fr.dropblock2d(xin * 1.0, p, bs, inplace, True)
        finally:
            torch.rand = real_rand
        r = real_rand(y.shape, generator=g)
        (gx,) = torch.autograd.grad((y * r).sum(), xin)
        drop.append({"x": x, "noise": noise, "drop_prob": p, "block_size": bs, "inplace": inplace, "y": y.detach(), "r": r, "dx": gx})
    save("losses.pt", {"poly": poly, "dice": dice, "dropblock": drop})


def _rand_targets(n_per_image, nc, g, same_cell=False):
    tg = []
    for k in n_per_image:
        b = torch.rand((k, 4), generator=g)
        b[:, :2] *= b[:, 2:]                      # the reference tests' recipe (tests/test_models_detection.py:40-42)
        b = torch.cat([b[:, :2], b[:, 2:].clamp(min=0.05)], 1)
        b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.02).clamp(max=0.999)
        if same_cell and k >= 2:
            b[1] = b[0] + torch.tensor([0.001, 0.002, 0.003, 0.001])
        tg.append({"boxes": b, "labels": torch.randint(0, nc, (k,), generator=g)})
    return tg


class _RecordRand:
    """Wrap torch.rand while the reference runs so that DropBlock's noise (functional.py:482) can be replayed."""

    def __init__(self, g):
        self.g, self.real, self.draws = g, torch.rand, []

    def __enter__(self):
        def rand(*size, **kw):
            kw.pop("device", None)
            t = self.real(*size, generator=self.g, **kw)
            self.draws.append(t.clone())
            return t
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.rand = self.real


def gen_yolo():
    """YoloLayer (decode / targets / losses / post-process) on random logits, and one training step of a reduced
    YOLOv4 whose weights are reproducible from the seed (the mirror's constructor consumes the RNG identically)."""
    import importlib
    ry = importlib.import_module("ref_holocron.models.detection.yolov4")
    g = torch.Generator().manual_seed(29)
    layer_cases = []
    anchors = torch.tensor([[36, 75], [76, 55], [72, 146]], dtype=torch.float32) / 608
    for (N, H, W, nc, counts, scale_xy, same) in [(2, 5, 5, 4, [2, 3], 1.2, False), (3, 7, 6, 3, [1, 0, 4], 1.1, True),
                                                  (2, 4, 4, 1, [0, 0], 1.05, False), (1, 9, 9, 20, [6], 1.2, True)]:
        layer = ry.YoloLayer(anchors.clone(), num_classes=nc, scale_xy=scale_xy)
        x = (torch.randn((N, 3 * (5 + nc), H, W), generator=g) * 1.5).requires_grad_(True)
        tg = _rand_targets(counts, nc, g, same)
        layer.train()
        boxes, b_o, b_s = layer._format_outputs(x)
        losses = layer(x, tg)
        wts = {"obj_loss": 1.0, "noobj_loss": 0.7, "bbox_loss": 1.3, "clf_loss": 0.9}
        (dx,) = torch.autograd.grad(sum(wts[k] * v.sum() for k, v in losses.items()), x)
        layer.eval()
        with torch.no_grad():
            dets = layer(x.detach().clone())
        layer_cases.append({"x": x.detach(), "target": tg, "nc": nc, "scale_xy": scale_xy, "anchors": anchors.clone(),
                            "boxes": boxes.detach(), "losses": {k: v.detach() for k, v in losses.items()}, "weights": wts,
                            "dx": dx, "detections": dets})
    # reduced YOLOv4: one block per stage, 16 stem channels, 5 classes, 2 x 3 x 128 x 128
    layout = [(64, 1), (128, 1), (256, 1), (512, 1), (1024, 1)]
    torch.manual_seed(41)
    m = ry.YOLOv4(layout, num_classes=5, stem_channels=16)
    gh = torch.Generator().manual_seed(43)
    for seq in (m.head.head1, m.head.head2_2, m.head.head3):       # the zero-initialised output convs would hide the backbone
        seq[-1].weight.data = torch.randn(seq[-1].weight.shape, generator=gh) * 0.05
        seq[-1].bias.data = torch.randn(seq[-1].bias.shape, generator=gh) * 0.5
    for mod in m.modules():
        if isinstance(mod, ref.nn.DropBlock2d):
            mod.p = 0.1 * 49 * 2                                   # gamma = 0.2 / 49: blocks really get dropped
    x = torch.rand((2, 3, 256, 256), generator=g)
    tg = _rand_targets([3, 2], 5, g)
    m.train()
    gn = torch.Generator().manual_seed(47)                         # the draws are replayed from this seed, not stored
    with _RecordRand(gn) as rec:
        losses = m(x, tg)
    total = sum(v.sum() for v in losses.values())
    total.backward()
    names = ["backbone.stem.0.weight", "backbone.stem.1.weight", "backbone.stages.0.base_layer.0.weight",
             "backbone.stages.2.main.0.conv.0.weight", "backbone.stages.2.main.0.conv.5.weight",
             "backbone.stages.4.transition.1.bias", "neck.fpn.1.weight", "neck.fpn.14.bias", "neck.pan2.conv1.0.weight",
             "neck.pan1.conv2.1.weight", "head.head1.3.weight", "head.head1.3.bias", "head.head2_2.3.weight",
             "head.head3.24.weight", "head.head3.24.bias", "head.pre_head3.1.weight"]
    params = dict(m.named_parameters())
    grads = {n: params[n].grad.clone() for n in names}
    gnorm = {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}
    rstats = {k: v.clone() for k, v in m.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    m.eval()
    with torch.no_grad():
        dets = m(x)
    save("yolo.pt", {"layers": layer_cases,
                     "model": {"layout": layout, "seed": 41, "head_seed": 43, "num_classes": 5, "stem_channels": 16,
                               "drop_p": 0.1 * 49 * 2, "x": x, "target": tg, "noise_seed": 47, "n_draws": len(rec.draws),
                               "dropped_fraction": [float((d <= 0.2 / 49).float().mean()) for d in rec.draws[:3]],
                               "losses": {k: v.detach() for k, v in losses.items()}, "grads": grads, "grad_norms": gnorm,
                               "running": rstats, "n_detections": [int(d["boxes"].shape[0]) for d in dets]}})


def gen_rexnet():
    """Reference ReXBlocks (with / without expansion, squeeze-excite, stride 2, partial-width shortcut), FReLU, and one
    training step of rexnet1_0x whose weights are reproducible from the seed."""
    import importlib
    rx = importlib.import_module("ref_holocron.models.classification.rexnet")
    g = torch.Generator().manual_seed(53)
    blocks = []
    for (cin, c, t, stride, se, hw) in [(32, 16, 1, 1, False, 12), (16, 27, 6, 2, False, 12), (38, 50, 6, 2, True, 10),
                                        (50, 61, 6, 1, True, 7), (61, 61, 6, 1, True, 5)]:
        torch.manual_seed(cin + c)
        blk = rx.ReXBlock(cin, c, t, stride, use_se=se)
        ref.nn.init.init_module(blk, "relu")
        _randomize_bn(blk, g)
        for p in blk.parameters():
            if p.dim() == 4:
                p.data = bf16r(p.data)
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = bf16r(torch.randn((3, cin, hw, hw), generator=g)).requires_grad_(True)
        blk.train()
        out = blk(x)
        r = bf16r(torch.randn(out.shape, generator=g))
        names = [n for n, _ in blk.named_parameters()]
        grads = torch.autograd.grad((out * r).sum(), [x] + list(blk.parameters()))
        blocks.append({"cfg": (cin, c, t, stride, se), "state": sd0, "x": x.detach(), "r": r, "out": out.detach(), "dx": grads[0],
                       "dparams": dict(zip(names, grads[1:])),
                       "state_after": {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "tracked" in k}})
    torch.manual_seed(7)
    fr = ref.nn.FReLU(24)
    _randomize_bn(fr, g)
    fr.conv.bias.data = torch.randn((24,), generator=g) * 0.3
    sd0 = {k: v.clone() for k, v in fr.state_dict().items()}
    x = bf16r(torch.randn((2, 24, 9, 9), generator=g)).requires_grad_(True)
    fr.train()
    out = fr(x)
    r = bf16r(torch.randn(out.shape, generator=g))
    names = [n for n, _ in fr.named_parameters()]
    grads = torch.autograd.grad((out * r).sum(), [x] + list(fr.parameters()))
    frelu = {"state": sd0, "x": x.detach(), "r": r, "out": out.detach(), "dx": grads[0], "dparams": dict(zip(names, grads[1:])),
             "state_after": {k: v.clone() for k, v in fr.state_dict().items()}}
    torch.manual_seed(51)
    m = rx.rexnet1_0x(num_classes=10, dropout_ratio=0.0)
    x = bf16r(torch.rand((4, 3, 96, 96), generator=g))
    t = torch.randint(0, 10, (4,), generator=g)
    m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, t)
    loss.backward()
    params = dict(m.named_parameters())
    keep = ["features.0.weight", "features.1.weight", "features.3.conv.0.weight", "features.4.conv.0.weight", "features.6.conv.3.weight",
            "features.6.conv.5.conv.0.weight", "features.6.conv.5.conv.3.bias", "features.10.conv.7.weight", "features.18.conv.4.weight",
            "features.20.weight", "head.1.weight", "head.1.bias"]
    save("rexnet.pt", {"blocks": blocks, "frelu": frelu,
                       "model": {"seed": 51, "num_classes": 10, "x": x, "target": t, "logits": logits.detach(), "loss": loss.detach(),
                                 "grads": {n: params[n].grad.clone() for n in keep},
                                 "grad_norms": {n: float(p.grad.norm()) for n, p in params.items()},
                                 "running": {k: v.clone() for k, v in m.state_dict().items()
                                             if k.endswith("running_mean") or k.endswith("running_var")}}})


def gen_mobileone():
    """Reference MobileOneBlocks (stem-like 3 channels, both identities, stride 2 with channel change, single branch), in
    training mode (outputs, gradients, running statistics), eval mode and re-parametrised; one training step and the
    re-parametrised inference of mobileone_s0 (weights reproducible from the seed)."""
    import importlib
    mo = importlib.import_module("ref_holocron.models.classification.mobileone")
    g = torch.Generator().manual_seed(71)
    blocks = []
    for (cin, cout, K, stride, hw) in [(3, 48, 4, 2, 16), (48, 48, 4, 1, 9), (48, 128, 2, 2, 10), (32, 32, 1, 1, 7)]:
        torch.manual_seed(cin + cout + K)
        blk = mo.MobileOneBlock(cin, cout, K, stride)
        ref.nn.init.init_module(blk, "relu")
        _randomize_bn(blk, g)
        for p in blk.parameters():
            if p.dim() == 4 and p.shape[1] != 1:
                p.data = bf16r(p.data)
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = bf16r(torch.randn((3, cin, hw, hw), generator=g)).requires_grad_(True)
        blk.train()
        out = blk(x)
        r = bf16r(torch.randn(out.shape, generator=g))
        names = [n for n, _ in blk.named_parameters()]
        grads = torch.autograd.grad((out * r).sum(), [x] + list(blk.parameters()))
        after = {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "tracked" in k}
        blk.load_state_dict(sd0)
        blk.eval()
        with torch.no_grad():
            out_eval = blk(x.detach())
            blk.reparametrize()
            out_rep = blk(x.detach())
        blocks.append({"cfg": (cin, cout, K, stride), "state": sd0, "x": x.detach(), "r": r, "out": out.detach(), "dx": grads[0],
                       "dparams": dict(zip(names, grads[1:])), "state_after": after, "out_eval": out_eval,
                       "rep_state": {k: v.clone() for k, v in blk.state_dict().items()}, "out_rep": out_rep})
    torch.manual_seed(61)
    m = mo.mobileone_s0(num_classes=10)
    x = bf16r(torch.rand((4, 3, 64, 64), generator=g))
    t = torch.randint(0, 10, (4,), generator=g)
    m.train()
    stage_means = {}
    hooks = [m.features[i].register_forward_hook(lambda mod, inp, out, i=i: stage_means.__setitem__(i, out.detach().mean((2, 3))))
             for i in (0, 1, 2)]
    logits = m(x)
    for hk in hooks:
        hk.remove()
    loss = torch.nn.functional.cross_entropy(logits, t)
    loss.backward()
    params = dict(m.named_parameters())
    keep = ["features.0.0.0.0.weight", "features.0.0.1.0.weight", "features.0.2.0.0.weight", "features.1.1.0.0.weight",
            "features.1.1.0.3.0.weight", "features.2.3.2.0.weight", "features.2.3.2.2.0.weight", "features.3.9.0.1.0.weight",
            "features.4.0.0.2.0.weight", "head.weight", "head.bias"]
    running = {k: v.clone() for k, v in m.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    m.eval()
    with torch.no_grad():
        logits_eval = m(x)
        m.reparametrize()
        logits_rep = m(x)
    save("mobileone.pt", {"blocks": blocks,
                          "model": {"seed": 61, "num_classes": 10, "x": x, "target": t, "logits": logits.detach(), "loss": loss.detach(),
                                    "grads": {n: params[n].grad.clone() for n in keep},
                                    "grad_norms": {n: float(p.grad.norm()) for n, p in params.items()},
                                    "running_sample": {k: running[k] for k in list(running)[:24]}, "stage_means": stage_means,
                                    "logits_eval": logits_eval, "logits_rep": logits_rep}})


def gen_convs():
    """SlimConv2d and NormConv2d of the reference (holocron/nn/modules/conv.py:55-147,262-370)."""
    g = torch.Generator().manual_seed(61)
    slim = []
    for (cin, k, stride, pad, r, hw, n) in [(32, 3, 1, 1, 8, 10, 3), (8, 3, 1, 1, 32, 7, 4), (64, 1, 1, 0, 16, 6, 2), (16, 3, 2, 1, 4, 9, 3)]:
        torch.manual_seed(cin + k)
        m = ref.nn.SlimConv2d(cin, k, stride=stride, padding=pad, r=r)
        _randomize_bn(m, g)
        for p in m.parameters():
            if p.dim() == 4:
                p.data = bf16r(p.data)
        sd0 = {kk: v.clone() for kk, v in m.state_dict().items()}
        x = bf16r(torch.randn((n, cin, hw, hw), generator=g)).requires_grad_(True)
        m.train()
        out = m(x)
        rr = bf16r(torch.randn(out.shape, generator=g))
        names = [nn_ for nn_, _ in m.named_parameters()]
        grads = torch.autograd.grad((out * rr).sum(), [x] + list(m.parameters()))
        slim.append({"cfg": (cin, k, stride, pad, r), "state": sd0, "x": x.detach(), "r": rr, "out": out.detach(), "dx": grads[0],
                     "dparams": dict(zip(names, grads[1:])),
                     "state_after": {kk: v.clone() for kk, v in m.state_dict().items() if "running" in kk}})
    norm = []
    for (cin, cout, k, stride, pad, mode, hw, n) in [(16, 24, 3, 1, 1, "zeros", 8, 2), (32, 16, 3, 2, 1, "zeros", 9, 2),
                                                    (8, 8, 3, 1, 1, "reflect", 6, 2), (16, 32, 1, 1, 0, "zeros", 5, 3)]:
        torch.manual_seed(cin + cout + k)
        m = ref.nn.NormConv2d(cin, cout, k, stride=stride, padding=pad, padding_mode=mode)
        m.weight.data = bf16r(m.weight.data)
        sd0 = {kk: v.clone() for kk, v in m.state_dict().items()}
        # the reference normalises the unfolded patches in place (functional.py:347-349): autograd refuses to
        # back-propagate into an input that requires grad, so (like tests/test_nn_conv.py:7-13) only dW / db exist
        x = bf16r(torch.randn((n, cin, hw, hw), generator=g) + 0.5)
        out = m(x)
        rr = bf16r(torch.randn(out.shape, generator=g))
        grads = torch.autograd.grad((out * rr).sum(), [m.weight, m.bias])
        norm.append({"cfg": (cin, cout, k, stride, pad, mode), "state": sd0, "x": x, "r": rr, "out": out.detach(),
                     "dw": grads[0], "db": grads[1]})
    save("convs.pt", {"slim": slim, "norm": norm})


def gen_optim2():
    """AdamP (with tensors on both sides of the projection test) and AdEMAMix trajectories of the reference; parameters and
    gradients come from `optim2_inputs` seeds, only the results are stored."""
    out = {"adamp": [], "ademamix": []}
    adamp_cfgs = [dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, delta=0.1),
                  dict(lr=5e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2, amsgrad=True, delta=0.3)]
    for case, kw in enumerate(adamp_cfgs):
        shapes = [(16, 8, 3, 3), (40,), (70000,) if case == 0 else (300,), (24, 24)]
        params = [torch.nn.Parameter(optim2_inputs(case, -1, k, sh)) for k, sh in enumerate(shapes)]
        opt = ref.optim.AdamP(params, **kw)
        proj, mid = [], None
        for it in range(3):
            gs = [optim2_inputs(case, it, k, p.shape, p.data, adamp=True) for k, p in enumerate(params)]
            for p, gr in zip(params, gs):
                p.grad = gr.clone()
            proj.append([bool(torch.nn.functional.cosine_similarity(p.data.view(1, -1), (gr + kw["weight_decay"] * p.data).view(1, -1)).max()
                              < kw["delta"] / p.numel() ** 0.5) for p, gr in zip(params, gs)])
            opt.step()
            if it == 0:
                mid = [p.data.clone() for p in params if p.numel() < 5000]
        out["adamp"].append({"kw": kw, "shapes": shapes, "projected": proj, "after_first_small": mid,
                             "final": [p.data.clone() for p in params],
                             "exp_avg_sq": [opt.state[p]["exp_avg_sq"].clone() for p in params if p.numel() < 5000]})
    mix_cfgs = [dict(lr=1e-2, betas=(0.9, 0.999, 0.9999), alpha=5.0, eps=1e-8, weight_decay=0.0),
                dict(lr=3e-3, betas=(0.8, 0.95, 0.99), alpha=2.0, eps=1e-6, weight_decay=5e-2)]
    for case, kw in enumerate(mix_cfgs):
        shapes = [(8, 4, 3, 3), (33,), (66000,) if case == 0 else (500,)]
        params = [torch.nn.Parameter(optim2_inputs(10 + case, -1, k, sh)) for k, sh in enumerate(shapes)]
        opt = ref.optim.AdEMAMix(params, **kw)
        for it in range(3):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(10 + case, it, k, p.shape)
            opt.step()
        out["ademamix"].append({"kw": kw, "shapes": shapes, "final": [p.data.clone() for p in params],
                                "exp_avg_slow": [opt.state[p]["exp_avg_slow"].clone() for p in params if p.numel() < 5000]})
    save("optim2.pt", out)


def gen_optim3():
    """LAMB, RaLars (through the step where the rectification switches on, and with force_adaptive_momentum), TAdam, Adan and
    the Lookahead / Scout wrappers of the reference; parameters and gradients come from `optim2_inputs` seeds."""
    out = {}

    def run(cls, case, kw, shapes, iters, extra_state=()):
        params = [torch.nn.Parameter(optim2_inputs(case, -1, k, sh) * (0.0 if (k == 1 and case % 2 == 1) else 1.0)) for k, sh in enumerate(shapes)]
        opt = cls(params, **kw)
        traj = []
        for it in range(iters):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(case, it, k, p.shape)
            opt.step()
            traj.append([p.data[..., :4].flatten()[:4].clone() for p in params])
        st = {name: [opt.state[p][name].clone() if torch.is_tensor(opt.state[p][name]) else torch.tensor(float(opt.state[p][name]))
                     for p in params if p.numel() < 5000 or name in ("local_lr", "W_t")] for name in extra_state}
        return {"kw": kw, "shapes": shapes, "iters": iters, "traj": traj, "final": [p.data.clone() for p in params if p.numel() < 5000],
                "final_sum": [float(p.data.double().sum()) for p in params], "state": st}

    out["lamb"] = [run(ref.optim.LAMB, 20, dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0), [(16, 8, 3, 3), (40,), (70000,)], 3, ("local_lr", "exp_avg_sq")),
                   run(ref.optim.LAMB, 21, dict(lr=5e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2, scale_clip=(0.5, 2.0)), [(8, 8), (24,), (300,)], 3, ("local_lr", "exp_avg"))]
    out["ralars"] = [run(ref.optim.RaLars, 22, dict(lr=1e-2, betas=(0.9, 0.9), eps=1e-8, weight_decay=0.0), [(16, 8, 3, 3), (40,), (66000,)], 8, ("local_lr", "exp_avg_sq")),
                     run(ref.optim.RaLars, 23, dict(lr=5e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2, force_adaptive_momentum=True, scale_clip=(0.1, 5.0)), [(8, 8), (24,), (300,)], 3, ("local_lr", "exp_avg"))]
    out["tadam"] = [run(ref.optim.TAdam, 24, dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0), [(16, 8, 3, 3), (40,), (70000,)], 3, ("W_t", "exp_avg")),
                    run(ref.optim.TAdam, 25, dict(lr=5e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2, amsgrad=True, dof=3.0), [(8, 8), (24,), (300,)], 3, ("W_t", "max_exp_avg_sq"))]
    out["adan"] = [run(ref.optim.Adan, 26, dict(lr=1e-2), [(16, 8, 3, 3), (40,), (70000,)], 3, ("exp_avg_sq", "exp_avg_delta", "prev_grad")),
                   run(ref.optim.Adan, 27, dict(lr=5e-3, betas=(0.9, 0.8, 0.95), eps=1e-6, weight_decay=1e-2, amsgrad=True), [(8, 8), (24,), (300,)], 3, ("max_exp_avg_delta",))]
    wrap = []
    for case, (wcls, kw) in enumerate([(ref.optim.wrapper.Lookahead, dict(sync_rate=0.5, sync_period=3)), (ref.optim.wrapper.Scout, dict(sync_rate=0.3, sync_period=2))]):
        shapes = [(8, 4, 3, 3), (33,), (500,)]
        params = [torch.nn.Parameter(optim2_inputs(30 + case, -1, k, sh)) for k, sh in enumerate(shapes)]
        opt = wcls(torch.optim.SGD(params, lr=0.1), **kw)
        traj = []
        for it in range(7):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(30 + case, it, k, p.shape)
            opt.step()
            traj.append([p.data.clone() for p in params])
        wrap.append({"cls": wcls.__name__, "kw": kw, "shapes": shapes, "traj": traj,
                     "slow": [p.data.clone() for g_ in opt.param_groups for p in g_["params"]]})
    out["wrapper"] = wrap
    save("optim3.pt", out)


def gen_yolo_v1():
    """The reference's own known-answer cases for YOLOv1 / YOLOv2 (tests/test_models_detection.py:95-233) and random
    predictions / targets through the reference `_compute_losses` (values + gradients), `post_process` and
    `_format_outputs`; one small training step and eval pass of each detector (weights reproducible from the seed)."""
    import importlib
    y1 = importlib.import_module("ref_holocron.models.detection.yolo")
    y2 = importlib.import_module("ref_holocron.models.detection.yolov2")
    g = torch.Generator().manual_seed(91)
    torch.manual_seed(5)
    m1 = y1.yolov1(num_classes=10, pretrained_backbone=False)
    torch.manual_seed(6)
    m2 = y2.yolov2(num_classes=10, pretrained_backbone=False)
    out = {"kat": [], "rand": [], "post": [], "fmt": {}}
    # ---- known-answer cases (values asserted by the reference tests)
    for tag, m, (h, w), A in (("v1", m1, (7, 7), 2), ("v2", m2, (13, 13), 5)):
        nc = 10
        if tag == "v1":
            target = [{"boxes": torch.tensor([[0, 0, 1 / 7, 1 / 7]], dtype=torch.float32), "labels": torch.zeros((1,), dtype=torch.long)}]
            pb = torch.zeros((1, h, w, A, 4)); pb[..., :2] = 0.5; pb[..., 2:] = 1 / 7; pb[0, 0, 0, 1, 0] = 0.8
            po = torch.zeros((1, h, w, A)); po[0, 0, 0, 0] = 0.5; po[0, -1, -1, 0] = 0.5
            ps = torch.zeros((1, h, w, 1, nc)); ps[0, 0, 0, 0, 0] = 0.5; ps[0, 0, 0, 0, 1:] = 0.5 / (nc - 1)
        else:
            target = [{"boxes": torch.tensor([[0, 0, 1, 1]], dtype=torch.float32), "labels": torch.zeros((1,), dtype=torch.long)}]
            pb = torch.zeros((1, h, w, A, 4)); pb[..., :2] = 0.5; pb[..., 2:] = 1
            pb[0, -1, -1, 0, 0] = (w - 1) / w; pb[0, -1, -1, 0, 1] = (h - 1) / h; pb[0, -1, -1, 0, 2] = 1 / w; pb[0, -1, -1, 0, 3] = 1 / h
            po = torch.zeros((1, h, w, A)); po[0, h // 2, w // 2, 0] = 0.5; po[0, -1, -1, 0] = 0.5
            ps = torch.zeros((1, h, w, 1, nc)); ps[0, h // 2, w // 2, 0, 0] = 0.5; ps[0, h // 2, w // 2, 0, 1:] = 0.5 / (nc - 1)
        with torch.no_grad():
            ld = m._compute_losses(pb, po, ps, target, ignore_high_iou=True)
        out["kat"].append({"tag": tag, "pb": pb, "po": po, "ps": ps, "target": target, "losses": {k: v.clone() for k, v in ld.items()},
                           "lambdas": (m.lambda_obj, m.lambda_noobj, m.lambda_coords, m.lambda_class)})
        n = 2
        bc = torch.zeros((n, h * w * A, 4)); bc[..., :2] = 0.5; bc[..., 2:] = (1 / h if tag == "v1" else 1)
        bo = torch.zeros((n, h * w * A)); bo[:, ::2] = 0.5
        bs = torch.zeros((n, h * w * A, nc)); bs[..., 0] = 0.5; bs[..., 1:] = 0.5 / (nc - 1)
        with torch.no_grad():
            dets = m.post_process(bc, bo, bs, (h, w))
        out["post"].append({"tag": tag, "bc": bc, "bo": bo, "bs": bs, "grid": (h, w), "A": A, "dets": dets, "kat": True})
    # ---- random predictions and targets: values and gradients
    for tag, m, (h, w), A, As in (("v1", m1, (7, 7), 2, 1), ("v2", m2, (13, 13), 5, 5), ("v2", m2, (5, 6), 3, 3)):
        nc = 10 if (h, w) != (5, 6) else 4
        N = 3
        for ignore in (False, True):
            pb = torch.rand((N, h, w, A, 4), generator=g)
            if tag == "v2":
                pb[..., 0] = (pb[..., 0] + torch.arange(w).view(1, 1, -1, 1)) / w
                pb[..., 1] = (pb[..., 1] + torch.arange(h).view(1, -1, 1, 1)) / h
            pb[..., 2:] = pb[..., 2:] * 0.5 + 0.05
            po = torch.rand((N, h, w, A), generator=g)
            ps = torch.softmax(torch.randn((N, h, w, As, nc), generator=g), -1)
            target = []
            for i, k in enumerate((3, 1, 5) if ignore else (3, 0, 5)):      # the reference cannot take an empty image with ignore_high_iou
                xy = torch.rand((k, 2), generator=g) * 0.55
                wh = torch.rand((k, 2), generator=g) * 0.4 + 0.04
                bx = torch.cat([xy, xy + wh], 1)
                if k == 5:
                    bx[4] = bx[3] + 0.004          # two boxes in one cell (and, with two anchors, often one anchor)
                target.append({"boxes": bx, "labels": torch.randint(0, nc, (k,), generator=g)})
            pb.requires_grad_(True); po.requires_grad_(True); ps.requires_grad_(True)
            ld = m._compute_losses(pb, po, ps, target, ignore_high_iou=ignore)
            wts = {"obj_loss": 1.0, "noobj_loss": 0.7, "bbox_loss": 1.3, "clf_loss": 0.9}
            total = sum(wts[k] * v.sum() for k, v in ld.items())
            grads = torch.autograd.grad(total, [pb, po, ps])
            out["rand"].append({"tag": tag, "ignore": ignore, "pb": pb.detach(), "po": po.detach(), "ps": ps.detach(), "target": target,
                                "weights": wts, "losses": {k: v.detach().clone() for k, v in ld.items()},
                                "grads": [t.clone() for t in grads],
                                "lambdas": (m.lambda_obj, m.lambda_noobj, m.lambda_coords, m.lambda_class)})
        bc = pb.detach().reshape(N, -1, 4)
        bo = po.detach().reshape(N, -1)
        bs = (ps.detach().repeat_interleave(A, dim=3) if As == 1 else ps.detach()).contiguous().reshape(N, -1, nc)
        if (h, w) == (5, 6):
            m2nc = y2.yolov2(num_classes=4, pretrained_backbone=False, anchors=torch.rand((3, 2), generator=g))
            dets = m2nc.post_process(bc, bo, bs, (h, w))
        else:
            with torch.no_grad():
                dets = m.post_process(bc, bo, bs, (h, w))
        out["post"].append({"tag": tag, "bc": bc, "bo": bo, "bs": bs, "grid": (h, w), "A": A, "dets": dets, "kat": False})
    # ---- _format_outputs
    x1 = torch.randn((2, 7 * 7 * (2 * 5 + 10)), generator=g)
    x2 = torch.randn((2, 5 * 15, 13, 13), generator=g)
    out["fmt"] = {"x1": x1, "v1": [t.clone() for t in m1._format_outputs(x1)], "x2": x2, "v2": [t.clone() for t in m2._format_outputs(x2)],
                  "anchors": m2.anchors.clone()}
    # ---- one training-mode forward / backward and one eval pass of each detector (weights reproducible from the seeds 5 / 6)
    out["model"] = {}
    for tag, m in (("v1", m1), ("v2", m2)):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0                       # the draw cannot be replayed on another device
        x = yolo12_image(tag)
        target = [{"boxes": torch.tensor([[0.1, 0.2, 0.5, 0.7], [0.55, 0.5, 0.95, 0.9]]), "labels": torch.tensor([1, 3])},
                  {"boxes": torch.tensor([[0.3, 0.3, 0.8, 0.6]]), "labels": torch.tensor([7])}]
        m.train()
        raw = m._forward(x)
        ld = m._compute_losses(*m._format_outputs(raw), target)
        sum(v.sum() for v in ld.values()).backward()
        params = dict(m.named_parameters())
        out["model"][tag] = {"target": target, "raw": raw.detach().clone(), "losses": {k: v.detach().clone() for k, v in ld.items()},
                             "grad_norms": {n: float(p.grad.norm()) for n, p in params.items() if p.grad is not None}}
    save("yolo_v1.pt", out)


def gen_mixup():
    """Reference Mixup (holocron/utils/data/collate.py) under fixed host seeds: index targets, dense targets, the binary case and
    alpha = 0; and the top-1 / top-5 accuracy arithmetic of ClassificationTrainer.evaluate (trainer/classification.py:60-66)."""
    import importlib
    col = importlib.import_module("ref_holocron.utils.data.collate")
    out = {"cases": []}
    g = torch.Generator().manual_seed(77)
    for seed, (nc, alpha, shape, tkind) in enumerate([(10, 0.4, (6, 3, 8, 8), "index"), (5, 1.0, (4, 2, 5, 5), "dense"), (1, 0.3, (5, 3, 4, 4), "binary"),
                                                      (7, 0.0, (3, 1, 4, 4), "index")]):
        x = torch.rand(shape, generator=g)
        if tkind == "index":
            t = torch.randint(0, nc, (shape[0],), generator=g)
        elif tkind == "dense":
            t = torch.rand((shape[0], nc), generator=g)
        else:
            t = torch.randint(0, 2, (shape[0],), generator=g).float()
        torch.manual_seed(1000 + seed)
        mx, mt = col.Mixup(nc, alpha)(x.clone(), t.clone())
        out["cases"].append({"nc": nc, "alpha": alpha, "x": x, "t": t, "seed": 1000 + seed, "mx": mx, "mt": mt})
    logits = torch.randn((64, 12), generator=g)
    logits[5, 3] = logits[5, 7]
    target = torch.randint(0, 12, (64,), generator=g)
    pred = logits.topk(5, dim=1)[1]
    correct = pred.eq(target.view(-1, 1).expand_as(pred))
    out["topk"] = {"logits": logits, "target": target, "top1": int(correct[:, 0].sum()), "top5": int(correct.any(dim=1).sum())}
    save("mixup.pt", out)


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


def gen_whole_models():
    """One training step of rexnet1_0x and mobileone_s0 on a batch whose LAST stage still holds 16 x 4 x 4 = 256 positions per channel
    (16 images of 128 x 128), so that batch-statistics BatchNorm is well conditioned down to the head and a whole-model comparison means
    something (with the 4-image / 2 x 2-map fixtures above it amplifies rounding to the 10-100 % level).  The weights
    are reproducible from the seed (the mirror draws the same RNG stream as the reference); the inputs are stored as bytes."""
    import importlib
    g = torch.Generator().manual_seed(97)
    out = {}
    for name, seed, mod, ctor, kw in (("rexnet1_0x", 151, "rexnet", "rexnet1_0x", {"dropout_ratio": 0.0}),
                                      ("mobileone_s0", 161, "mobileone", "mobileone_s0", {})):
        rm = importlib.import_module("ref_holocron.models.classification." + mod)
        torch.manual_seed(seed)
        m = getattr(rm, ctor)(num_classes=10, **kw)
        x8 = torch.randint(0, 256, (16, 3, 128, 128), generator=g, dtype=torch.uint8)
        x = bf16r(x8.float() / 255.0)
        t = torch.randint(0, 10, (16,), generator=g)
        m.train()
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits, t)
        loss.backward()
        params = dict(m.named_parameters())
        # full gradients of every tensor up to 4096 elements (BatchNorm affine, biases, depthwise taps, narrow convs) + the head
        keep = {n: p.grad.clone() for n, p in params.items() if p.numel() <= 4096 or n.startswith("head")}
        running = {k: v.clone() for k, v in m.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
        out[name] = {"seed": seed, "num_classes": 10, "kwargs": kw, "x8": x8, "target": t, "logits": logits.detach(), "loss": loss.detach(),
                     "grads": keep, "grad_norms": {n: float(p.grad.norm()) for n, p in params.items()},
                     "grad_abs_max": {n: float(p.grad.abs().max()) for n, p in params.items()},
                     "running_sample": {k: running[k] for k in list(running)[:8] + list(running)[-8:]}}
    save("whole_models.pt", out)


if __name__ == "__main__":
    gens = {"whole_models": gen_whole_models, "boxes": gen_boxes, "functional": gen_functional, "optim": gen_optim, "repblock": gen_repblock,
            "repvgg_small": gen_repvgg_small, "darknet": gen_darknet, "losses": gen_losses, "yolo": gen_yolo, "rexnet": gen_rexnet, "convs": gen_convs, "optim2": gen_optim2, "nms": gen_nms, "mobileone": gen_mobileone, "optim3": gen_optim3, "yolo_v1": gen_yolo_v1, "mixup": gen_mixup}
    for name in (sys.argv[1:] or list(gens)):
        gens[name]()
