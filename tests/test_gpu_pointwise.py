"""GPU: pointwise / box / NMS / loss / optimizer kernels against golden vectors and the oracle.
Box indices and NMS results are integer work: bit-exact.  Box arithmetic (fp32, no contraction) is
bit-exact too except the atan-based aspect term."""
import os

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_boxes_bit_exact_vs_reference(golden):
    from holocron_amd.ops import boxes as hb
    g = golden("boxes.pt")
    for tag, (x, y) in {"rand": (g["b1"], g["b2"]), "kat": (g["kat_boxes"], g["kat_boxes"])}.items():
        ref = g[tag]
        xc, yc = x.cuda(), y.cuda()
        assert torch.equal(hb.box_iou(xc, yc).cpu(), ref["iou"])
        assert torch.equal(hb.box_giou(xc, yc).cpu(), ref["giou"])
        assert torch.equal(hb.iou_penalty(xc, yc).cpu(), ref["penalty"])
        assert torch.equal(hb.diou_loss(xc, yc).cpu(), ref["diou"])
        assert torch.equal(hb.ciou_loss(xc, yc).cpu(), ref["ciou"])
        assert torch.allclose(hb.aspect_ratio_consistency(xc, yc).cpu(), ref["arc"], rtol=1e-5, atol=1e-7)
    with pytest.raises(AssertionError):
        hb.box_giou(torch.tensor([[1.0, 1.0, 0.0, 2.0]]).cuda(), g["b2"].cuda())


def test_boxes_large_vs_oracle():
    from holocron_amd.ops import boxes as hb
    from oracle import boxes as ob, tv_ops
    g = torch.Generator().manual_seed(1)
    xy = torch.rand((3000, 2), generator=g)
    b1 = torch.cat([xy, xy + torch.rand((3000, 2), generator=g) * 0.3 + 1e-3], 1)
    xy = torch.rand((257, 2), generator=g)
    b2 = torch.cat([xy, xy + torch.rand((257, 2), generator=g) * 0.3 + 1e-3], 1)
    assert torch.equal(hb.box_iou(b1.cuda(), b2.cuda()).cpu(), tv_ops.box_iou(b1, b2))
    assert torch.equal(hb.diou_loss(b1.cuda(), b2.cuda()).cpu(), ob.diou_loss(b1, b2))
    assert torch.equal(hb.box_giou(b1.cuda(), b2.cuda()).cpu(), ob.box_giou(b1, b2))
    assert hb.box_iou(b1[:0].cuda(), b2.cuda()).shape == (0, 257)


def test_nms_bit_exact(golden):
    from holocron_amd.ops import boxes as hb
    for c in golden("nms.pt"):
        keep = hb.nms(c["boxes"].cuda(), c["scores"].cuda(), c["thr"])
        assert keep.dtype == torch.int64
        assert torch.equal(keep.cpu(), c["keep"]), c["pinned_by"]
    assert hb.nms(torch.zeros((0, 4)).cuda(), torch.zeros((0,)).cuda(), 0.5).numel() == 0


def test_nms_yolov4_scale_all_equal_scores():
    """a zero-initialised YOLOv4 head hands NMS thousands of boxes with identical scores;
    the stable sort order decides the result.  4332 boxes (one 38x38x3 scale) vs the oracle."""
    from holocron_amd.ops import boxes as hb
    from oracle import tv_ops
    g = torch.Generator().manual_seed(2)
    n = 4332
    c = torch.rand((n, 2), generator=g)
    wh = torch.rand((n, 2), generator=g) * 0.2 + 0.01
    b = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    s = torch.full((n,), 0.25)
    assert torch.equal(hb.nms(b.cuda(), s.cuda(), 0.7).cpu(), tv_ops.nms(b, s, 0.7))


def test_hard_mish_and_focal(golden):
    import holocron_amd as h
    g = golden("functional.pt")
    hm = g["hard_mish"]
    x = hm["x"].cuda().requires_grad_(True)
    y = h.nn.functional.hard_mish(x)
    assert torch.equal(y.detach().cpu(), hm["y"])
    (y * hm["r"].cuda()).sum().backward()
    assert torch.allclose(x.grad.cpu(), hm["dx"], rtol=1e-6, atol=1e-7)
    z = hm["x"].cuda().clone()
    z2 = h.nn.functional.hard_mish(z, inplace=True)
    assert z2.data_ptr() == z.data_ptr() and torch.equal(z.cpu(), hm["y"])      # tests/test_nn_activation.py:24-27
    for c in g["focal"]:
        x = c["x"].cuda().requires_grad_(True)
        w = None if c["weight"] is None else c["weight"].cuda()
        loss = h.nn.functional.focal_loss(x, c["target"].cuda(), w, c["ignore_index"], c["reduction"], c["gamma"])
        assert torch.allclose(loss.detach().cpu(), c["loss"], rtol=2e-5, atol=1e-6)
        (loss * c["r"].cuda()).sum().backward()
        assert torch.allclose(x.grad.cpu(), c["dx"], rtol=2e-4, atol=1e-6)
    with pytest.raises(NotImplementedError):
        h.nn.FocalLoss(reduction="bogus")


def test_poly_dice_dropblock(golden):
    import holocron_amd as h
    Fh = h.nn.functional
    g = golden("losses.pt")
    for c in g["poly"]:
        x = c["x"].cuda().requires_grad_(True)
        w = None if c["weight"] is None else c["weight"].cuda()
        loss = Fh.poly_loss(x, c["target"].cuda(), c["eps"], w, c["ignore_index"], c["reduction"])
        assert loss.shape == c["loss"].shape
        assert torch.allclose(loss.detach().cpu(), c["loss"], rtol=2e-5, atol=2e-6)
        (loss * c["r"].cuda()).sum().backward()
        assert torch.allclose(x.grad.cpu(), c["dx"], rtol=2e-4, atol=2e-6)
    with pytest.raises(TypeError):
        Fh.poly_loss(torch.zeros(2, 3).cuda(), torch.zeros(2).cuda())            # functional.py:568-569
    with pytest.raises(ValueError):
        Fh.poly_loss(torch.zeros(2, 3).cuda(), torch.zeros(2, 4).cuda())         # functional.py:574-575
    assert isinstance(h.nn.PolyLoss(eps=1.0)(torch.randn(4, 3).cuda(), torch.randint(0, 3, (4,)).cuda()).item(), float)
    for c in g["dice"]:
        x = c["x"].cuda().requires_grad_(True)
        w = None if c["weight"] is None else c["weight"].cuda()
        loss = Fh.dice_loss(x, c["target"].cuda(), w, c["gamma"], c["eps"])
        assert torch.allclose(loss.detach().cpu(), c["loss"], rtol=1e-5, atol=1e-6)
        loss.backward()
        assert torch.allclose(x.grad.cpu(), c["dx"], rtol=1e-4, atol=1e-7)
    # identities the reference tests pin (tests/test_nn_loss.py): perfect prediction -> ~0
    t = torch.zeros(2, 3, 8, 8).scatter_(1, torch.randint(0, 3, (2, 1, 8, 8)), 1.0).cuda()
    assert abs(float(h.nn.DiceLoss()(t, t))) < 1e-6
    for c in g["dropblock"]:
        x = c["x"].cuda().requires_grad_(True)
        xin = x * 1.0
        y = Fh.dropblock2d(xin, c["drop_prob"], c["block_size"], c["inplace"], True, noise=c["noise"].cuda())
        if c["inplace"]:
            assert y.data_ptr() == xin.data_ptr()
        assert torch.allclose(y.detach().cpu(), c["y"], rtol=1e-6, atol=0)
        (y * c["r"].cuda()).sum().backward()
        assert torch.allclose(x.grad.cpu(), c["dx"], rtol=1e-6, atol=0)
    # bf16 channels_last activations take the same kernel (the layout the conv path hands over)
    c = g["dropblock"][1]
    xb = c["x"].cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    yb = Fh.dropblock2d(xb, c["drop_prob"], c["block_size"], False, True, noise=c["noise"].cuda())
    from oracle import functional as of
    ref = of.dropblock2d(xb.float().cpu(), c["drop_prob"], c["block_size"], c["noise"])
    assert torch.allclose(yb.float().cpu(), ref, rtol=1.6e-2, atol=1e-6)
    # eval / p == 0: identity, same object (functional.py:476-477)
    m = h.nn.DropBlock2d(0.1, 3).eval()
    z = torch.rand(1, 2, 4, 4).cuda()
    assert m(z) is z
    m.train()
    out = m(z)
    assert out.shape == z.shape


def test_dropblock_batched_masks_equal_per_call_masks():
    """hc_dropblock_mask_batched (one launch per step) against hc_dropblock_mask per layer on the same noise."""
    import ctypes as C
    import numpy as np
    from holocron_amd import _lib
    from holocron_amd._lib import check, ptr, stream
    lib = _lib.load()
    # single-tile maps, maps of several 32 x 32 tiles (ragged edges, tiles straddling images) and more layers than one launch of the
    # flat tile list takes (256): the tile -> (layer, image, ty, tx) decode of the persistent grid
    shapes = [(2, 9, 11, 3, 0.05), (3, 19, 19, 7, 0.004), (1, 5, 4, 5, 0.2), (2, 16, 16, 7, 1e-9), (2, 76, 76, 7, 0.004),
              (1, 100, 37, 5, 0.01), (3, 33, 64, 3, 0.02), (2, 152, 152, 13, 0.0005)]
    shapes += [(1 + i % 3, 7 + i % 11, 5 + i % 40, 3 + 2 * (i % 3), 0.01 + 0.001 * (i % 7)) for i in range(290)]
    total = sum(n * h * w for n, h, w, _, _ in shapes)
    noise = torch.rand((total,), device="cuda")
    arr = (_lib.DropItem * len(shapes))()
    off = 0
    for a, (n, h, w, bs, gamma) in zip(arr, shapes):
        a.off, a.N, a.H, a.W, a.block_size, a.gamma = off, n, h, w, bs, gamma
        off += n * h * w
    tab = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).cuda()
    keep = torch.empty((total,), device="cuda")
    counts = torch.empty((len(shapes),), device="cuda")
    check(lib.hc_dropblock_mask_batched(tab.data_ptr(), len(shapes), max(n * h * w for n, h, w, _, _ in shapes), ptr(noise), ptr(keep),
                                        ptr(counts), stream()), "hc_dropblock_mask_batched")
    off = 0
    for i, (n, h, w, bs, gamma) in enumerate(shapes):
        k1 = torch.empty((n * h * w,), device="cuda")
        c1 = torch.empty((1,), device="cuda")
        nz = noise[off:off + n * h * w].contiguous()
        check(lib.hc_dropblock_mask(ptr(nz), ptr(k1), ptr(c1), n, h, w, bs, gamma, stream()), "hc_dropblock_mask")
        assert torch.equal(keep[off:off + n * h * w], k1) and float(counts[i]) == float(c1)
        off += n * h * w


def test_cross_entropy_matches_torch_cpu_fp32():
    """F.cross_entropy / nn.CrossEntropyLoss (the training loop's criterion, references/classification/train.py:194): value and
    gradient against torch's own composition on the CPU in fp32, with label smoothing, ignored rows, a non-unit upstream gradient,
    bf16 logits and more rows than one workgroup pass (1024)."""
    import holocron_amd as h
    g = torch.Generator().manual_seed(3)
    for N, K, ls, ign, dt in [(256, 10, 0.1, -100, torch.float32), (37, 1000, 0.0, -100, torch.float32),
                              (3000, 17, 0.2, 5, torch.float32), (64, 10, 0.1, -100, torch.bfloat16),
                              # wide heads (K > 64): the wave-per-row kernels - ignored rows, K not a multiple of 4, more rows than
                              # one pass of the single-workgroup sum, bf16 logits
                              (256, 1000, 0.1, 7, torch.float32), (33, 1001, 0.1, -100, torch.float32),
                              (5000, 100, 0.1, 3, torch.float32), (64, 1000, 0.1, -100, torch.bfloat16)]:
        x = (torch.randn((N, K), generator=g) * 3).to(dt)
        t = torch.randint(0, K, (N,), generator=g)
        up = torch.rand((), generator=g) + 0.5
        xr = x.float().clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(xr, t, label_smoothing=ls, ignore_index=ign)
        (ref * up).backward()
        xg = x.cuda().requires_grad_(True)
        out = h.nn.functional.cross_entropy(xg, t.cuda(), label_smoothing=ls, ignore_index=ign)
        (out * up.cuda()).backward()
        assert out.dtype == torch.float32 and out.shape == ()
        assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (N, K, float(out), float(ref))
        tol = 1e-6 if dt == torch.float32 else 4e-3          # bf16 logits: the returned gradient is rounded to bf16
        assert xg.grad.dtype == dt
        assert torch.allclose(xg.grad.float().cpu(), xr.grad, rtol=tol, atol=tol / N), (N, K, (xg.grad.float().cpu() - xr.grad).abs().max())
        if ign >= 0:
            assert bool((xg.grad[t.cuda() == ign] == 0).all())
    # logits that start 4 bytes into an allocation (no 16-byte loads): same numbers as the aligned copy
    flat = (torch.randn((1 + 48 * 1000,), generator=g) * 3).cuda()
    xm, t = flat[1:].view(48, 1000).requires_grad_(True), torch.randint(0, 1000, (48,), generator=g).cuda()
    xa = flat[1:].clone().view(48, 1000).requires_grad_(True)
    lm, la = h.nn.functional.cross_entropy(xm, t, 0.1), h.nn.functional.cross_entropy(xa, t, 0.1)
    gm, ga = torch.autograd.grad(lm, xm)[0], torch.autograd.grad(la, xa)[0]
    assert abs(float(lm) - float(la)) < 2e-6 and torch.allclose(gm, ga, rtol=1e-6, atol=1e-9)
    # the module routes the common case to the HIP launches and anything else to torch
    crit = h.nn.CrossEntropyLoss(label_smoothing=0.1)
    x = torch.randn((32, 10), generator=g).cuda()
    t = torch.randint(0, 10, (32,), generator=g).cuda()
    assert abs(float(crit(x, t)) - float(torch.nn.functional.cross_entropy(x.cpu(), t.cpu(), label_smoothing=0.1))) < 2e-6
    w = torch.rand(10).cuda()
    assert torch.allclose(h.nn.CrossEntropyLoss(weight=w)(x, t), torch.nn.functional.cross_entropy(x, t, weight=w))
    # two passes agree bit for bit (fixed-order sums)
    a, b = h.nn.functional.cross_entropy(x, t, 0.1), h.nn.functional.cross_entropy(x, t, 0.1)
    assert torch.equal(a, b)
    # a class index outside [0, K) that is not ignore_index (torch asserts on the device): NaN loss here, never an out-of-bounds read
    #  - narrow and wide heads
    for K in (10, 1000):
        xb = torch.randn((16, K), generator=g).cuda()
        tb = torch.randint(0, K, (16,), generator=g).cuda()
        for bad in (K, K + 12345, -1):
            tb2 = tb.clone()
            tb2[3] = bad
            assert torch.isnan(h.nn.functional.cross_entropy(xb, tb2, 0.1)), (K, bad)
        assert torch.isfinite(h.nn.functional.cross_entropy(xb, tb, 0.1))


def test_global_avg_pool_matches_adaptive_avg_pool():
    """tests/test_nn_downsample.py:37-50 of the reference: GlobalAvgPool2d == AdaptiveAvgPool2d(1), plus gradients."""
    import holocron_amd as h
    g = torch.Generator().manual_seed(9)
    for (n, c, hh, ww) in [(2, 16, 7, 7), (3, 240, 28, 28), (2, 2056, 5, 4), (5, 1280, 1, 1), (4, 48, 56, 56)]:
        x = torch.randn((n, c, hh, ww), generator=g).to(torch.bfloat16).float()
        xg = x.cuda().requires_grad_(True)
        y = h.nn.GlobalAvgPool2d(flatten=True)(xg)
        ref = x.mean(dim=(2, 3))
        assert y.shape == (n, c) and torch.allclose(y.float().cpu(), ref, rtol=1e-5, atol=1e-6)
        r = torch.randn((n, c), generator=g)
        (y.float() * r.cuda()).sum().backward()
        assert torch.allclose(xg.grad.float().cpu(), (r / (hh * ww)).view(n, c, 1, 1).expand(n, c, hh, ww), rtol=1e-2, atol=1e-6)
    assert h.nn.GlobalAvgPool2d(flatten=False)(torch.rand(2, 8, 4, 4).cuda()).shape == (2, 8, 1, 1)


def test_adamp_and_ademamix_match_reference(golden):
    import holocron_amd as h
    from _inputs import optim2_inputs
    g = golden("optim2.pt")
    for case, c in enumerate(g["adamp"]):
        params = [torch.nn.Parameter(optim2_inputs(case, -1, k, sh).cuda()) for k, sh in enumerate(c["shapes"])]
        opt = h.optim.AdamP(params, **c["kw"])
        for it in range(3):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(case, it, k, p.shape, p.detach().cpu(), adamp=True).cuda()
            opt.step()
            if it == 0:
                small = [p for p in params if p.numel() < 5000]
                for p, r in zip(small, c["after_first_small"]):
                    assert torch.allclose(p.detach().cpu(), r, rtol=2e-5, atol=1e-6)
        for p, f in zip(params, c["final"]):
            assert torch.allclose(p.detach().cpu(), f, rtol=1e-4, atol=2e-6), (case, p.shape)
        st = opt.state[params[0]]
        assert st["step"] == 3 and set(st) >= {"exp_avg", "exp_avg_sq"}
        assert torch.allclose(st["exp_avg_sq"].cpu(), c["exp_avg_sq"][0], rtol=1e-5, atol=1e-9)
    for case, c in enumerate(g["ademamix"]):
        params = [torch.nn.Parameter(optim2_inputs(10 + case, -1, k, sh).cuda()) for k, sh in enumerate(c["shapes"])]
        opt = h.optim.AdEMAMix(params, **c["kw"])
        for it in range(3):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(10 + case, it, k, p.shape).cuda()
            opt.step()
        for p, f in zip(params, c["final"]):
            assert torch.allclose(p.detach().cpu(), f, rtol=2e-5, atol=1e-6), (case, p.shape)
        assert torch.allclose(opt.state[params[0]]["exp_avg_slow"].cpu(), c["exp_avg_slow"][0], rtol=1e-5, atol=1e-8)
    with pytest.raises(ValueError):
        h.optim.AdEMAMix([torch.nn.Parameter(torch.zeros(1))], betas=(0.9, 0.999, 1.0))      # ademamix.py:68-70


def test_lamb_ralars_tadam_adan_match_reference(golden):
    import holocron_amd as h
    from _inputs import optim2_inputs
    g = golden("optim3.pt")
    specs = [("lamb", h.optim.LAMB, (20, 21), ("local_lr",)), ("ralars", h.optim.RaLars, (22, 23), ("local_lr",)),
             ("tadam", h.optim.TAdam, (24, 25), ("W_t",)), ("adan", h.optim.Adan, (26, 27), ())]
    for name, cls, cases, snames in specs:
        for case, c in zip(cases, g[name]):
            params = [torch.nn.Parameter((optim2_inputs(case, -1, k, sh) * (0.0 if (k == 1 and case % 2 == 1) else 1.0)).cuda())
                      for k, sh in enumerate(c["shapes"])]
            opt = cls(params, **c["kw"])
            for it in range(c["iters"]):
                for k, p in enumerate(params):
                    p.grad = optim2_inputs(case, it, k, p.shape).cuda()
                opt.step()
                for p, f in zip(params, c["traj"][it]):
                    assert torch.allclose(p.data[..., :4].flatten()[:4].cpu(), f, rtol=2e-5, atol=1e-6), (name, case, it)
            small = [p for p in params if p.numel() < 5000]
            for p, f in zip(small, c["final"]):
                assert torch.allclose(p.data.cpu(), f, rtol=2e-5, atol=1e-6), (name, case, float((p.data.cpu() - f).abs().max()))
            for p, fs in zip(params, c["final_sum"]):
                assert abs(float(p.data.double().sum()) - fs) < 2e-3 * max(1.0, abs(fs)) + 1e-2, (name, case)
            for sn in snames:
                for p, f in zip(params, c["state"][sn]):
                    v = opt.state[p][sn]
                    assert torch.allclose(torch.as_tensor(v).float().cpu().flatten(), f.flatten().float(), rtol=1e-4, atol=1e-6), (name, case, sn)
            for sn, vals in c["state"].items():
                if sn in snames:
                    continue
                for p, f in zip(small, vals):
                    assert torch.allclose(opt.state[p][sn].cpu(), f, rtol=1e-4, atol=1e-7), (name, case, sn)
            assert set(opt.state[params[0]].keys()) >= set(c["state"].keys()) | {"step"}


def test_lookahead_and_scout_match_reference(golden):
    import holocron_amd as h
    from _inputs import optim2_inputs
    for case, c in enumerate(golden("optim3.pt")["wrapper"]):
        params = [torch.nn.Parameter(optim2_inputs(30 + case, -1, k, sh).cuda()) for k, sh in enumerate(c["shapes"])]
        wcls = getattr(h.optim.wrapper, c["cls"])
        opt = wcls(torch.optim.SGD(params, lr=0.1), **c["kw"])
        for it in range(len(c["traj"])):
            for k, p in enumerate(params):
                p.grad = optim2_inputs(30 + case, it, k, p.shape).cuda()
            opt.step()
            for p, f in zip(params, c["traj"][it]):
                assert torch.allclose(p.data.cpu(), f, rtol=1e-5, atol=1e-6), (c["cls"], it)
        slow = [p for g_ in opt.param_groups for p in g_["params"]]
        for p, f in zip(slow, c["slow"]):
            assert torch.allclose(p.data.cpu(), f, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        h.optim.Lookahead(torch.optim.SGD(params, lr=0.1), sync_rate=1.5)
    with pytest.raises(ValueError):
        h.optim.Scout(torch.optim.SGD(params, lr=0.1), sync_period=0)


def test_adabelief_matches_reference(golden):
    import holocron_amd as h
    for c in golden("optim.pt")["adabelief"]:
        p = torch.nn.Parameter(c["p0"].clone().cuda())
        opt = h.optim.AdaBelief([p], **c["kw"])
        for i, gr in enumerate(c["grads"]):
            p.grad = gr.clone().cuda()
            opt.step()
            assert torch.allclose(p.detach().cpu(), c["traj"][i], rtol=1e-6, atol=1e-7), (c["kw"], i)
        st = opt.state[p]
        assert st["step"] == len(c["grads"])
        assert torch.allclose(st["exp_avg"].cpu(), c["exp_avg"], rtol=1e-6, atol=1e-8)
        assert torch.allclose(st["exp_avg_sq"].cpu(), c["exp_avg_sq"], rtol=1e-5, atol=1e-10)


def test_adabelief_multi_tensor_large_vs_oracle():
    import holocron_amd as h
    from oracle import optim as oo
    torch.manual_seed(0)
    shapes = [(1280, 640, 3, 3), (48, 3, 3, 3), (1280,), (10, 1280), (7,)]
    ps = [torch.nn.Parameter(torch.randn(s).cuda()) for s in shapes]
    ref = [p.detach().cpu().clone() for p in ps]
    ms, ss = [torch.zeros_like(r) for r in ref], [torch.zeros_like(r) for r in ref]
    opt = h.optim.AdaBelief([{"params": ps[:2], "lr": 1e-3}, {"params": ps[2:], "lr": 5e-3, "weight_decay": 1e-2}],
                            betas=(0.95, 0.99), eps=1e-6)
    for step in range(1, 3):
        gs = [torch.randn(s) for s in shapes]
        for p, g_ in zip(ps, gs):
            p.grad = g_.cuda()
        opt.step()
        for i in range(len(ps)):
            lr, wd = (1e-3, 0.0) if i < 2 else (5e-3, 1e-2)
            oo.adabelief_step(ref[i], gs[i], ms[i], ss[i], step, lr, 0.95, 0.99, 1e-6, wd)
            assert torch.allclose(ps[i].detach().cpu(), ref[i], rtol=1e-5, atol=1e-6), (step, i)


def test_lars_matches_reference(golden):
    import holocron_amd as h
    for c in golden("optim.pt")["lars"]:
        p = torch.nn.Parameter(c["p0"].clone().cuda())
        opt = h.optim.LARS([p], **c["kw"])
        for i, gr in enumerate(c["grads"]):
            p.grad = gr.clone().cuda()
            opt.step()
            assert torch.allclose(p.detach().cpu(), c["traj"][i], rtol=2e-5, atol=1e-6), (c["kw"], i)
            assert torch.allclose(p.grad.cpu(), c["grad_after"][i], rtol=1e-6, atol=1e-7)   # weight decay lands in .grad (Q4)
    with pytest.raises(ValueError):
        h.optim.LARS([torch.nn.Parameter(torch.zeros(1))], lr=-1.0)
    with pytest.raises(ValueError):
        h.optim.LARS([torch.nn.Parameter(torch.zeros(1))], lr=1e-3, nesterov=True)


def test_mixup_and_topk_match_reference(golden):
    import holocron_amd as h
    g = golden("mixup.pt")
    for c in g["cases"]:
        torch.manual_seed(c["seed"])
        mx, mt = h.utils.data.Mixup(c["nc"], c["alpha"])(c["x"].cuda(), c["t"].cuda())
        assert mx.shape == c["mx"].shape and mt.shape == c["mt"].shape and mt.dtype == c["mt"].dtype
        assert torch.allclose(mx.cpu(), c["mx"], rtol=1e-6, atol=1e-7)
        assert torch.allclose(mt.cpu(), c["mt"], rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        h.utils.data.Mixup(10, -0.1)
    xb = c["x"].cuda().to(torch.bfloat16)
    torch.manual_seed(3)
    mb, _ = h.utils.data.Mixup(7, 0.5)(xb, c["t"].cuda())
    assert mb.dtype == torch.bfloat16 and mb.shape == xb.shape
    t = g["topk"]
    acc = h.utils.metrics.TopKAccuracy(5)
    acc.update(t["logits"][:40].cuda(), t["target"][:40].cuda())
    acc.update(t["logits"][40:].cuda(), t["target"][40:].cuda())
    a1, a5, n = acc.compute()
    assert n == 64 and abs(a1 * 64 - t["top1"]) < 1e-6 and abs(a5 * 64 - t["top5"]) <= 1      # a tie at rank 5 may go either way in topk
    small = h.utils.metrics.TopKAccuracy(5)
    small.update(torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1]]).cuda(), torch.tensor([1, 2]).cuda())
    assert small.compute() == (0.5, 0.0, 2)                                                 # fewer than 5 classes: top-1 only

    class _M(torch.nn.Module):
        def forward(self, x):
            return x
    lg, tg = t["logits"].cuda(), t["target"].cuda()
    loader = [(lg[:32], tg[:32]), (lg[32:], tg[32:])]
    res = h.utils.metrics.evaluate_classification(_M(), loader, torch.nn.functional.cross_entropy, lg.device)
    ref_loss = (torch.nn.functional.cross_entropy(t["logits"][:32], t["target"][:32]) + torch.nn.functional.cross_entropy(t["logits"][32:], t["target"][32:])) / 2
    assert abs(res["val_loss"] - float(ref_loss)) < 1e-5 and abs(res["acc1"] - t["top1"] / 64) < 1e-6


def test_lars_step_is_graph_capturable_and_uploads_tables_once():
    """device-resident chunk / group tables (optim/_multi_tensor.py DeviceTables): after the first step nothing is uploaded, and a
    captured step replays to the same parameters as eager steps."""
    import holocron_amd as h
    from holocron_amd.optim import _multi_tensor as mt
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (10, 64)]
    grads = [[torch.randn(s, device="cuda") for s in shapes] for _ in range(3)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).cuda()) for i, s in enumerate(shapes)]
        for p in ps:
            p.grad = torch.zeros_like(p)
        return ps, h.optim.LARS(ps, lr=0.1, momentum=0.9, weight_decay=1e-4)
    ps_e, opt_e = make()
    uploads = []
    orig = mt.Staging.upload

    def counting(self, raw):
        uploads.append(raw.size)
        return orig(self, raw)
    mt.Staging.upload = counting
    try:
        per_step = []
        for gs in grads:
            for p, g_ in zip(ps_e, gs):
                p.grad.copy_(g_)
            before = len(uploads)
            opt_e.step()
            per_step.append(len(uploads) - before)
        torch.cuda.synchronize()
        # step 1: chunk table + group block; step 2: the chunk table once more (the "first momentum step" flag of lars.py:126-130
        # clears); from then on nothing crosses PCIe
        assert per_step[0] == 2 and per_step[1] <= 1 and per_step[2] == 0, per_step
    finally:
        mt.Staging.upload = orig
    ps_g, opt_g = make()
    for p, g_ in zip(ps_g, grads[0]):
        p.grad.copy_(g_)
    opt_g.step()                                   # warm-up steps: momentum buffers exist, first-step flags cleared, tables resident
    for p, g_ in zip(ps_g, grads[1]):
        p.grad.copy_(g_)
    opt_g.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt_g.step()
    for gs in grads[2:]:
        for p, g_ in zip(ps_g, gs):
            p.grad.copy_(g_)
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(ps_e, ps_g):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["AdaBelief", "LAMB", "RaLars", "TAdam", "AdamP", "AdEMAMix", "Adan"])
def test_optimizers_with_a_parameter_that_skips_steps(name):
    """A parameter whose ``grad is None`` on some iterations keeps its own step count in the reference (per-parameter
    ``state['step']``, e.g. holocron/optim/adabelief.py:120-128): here it moves into a launch group of its own.  Checked against
    two single-parameter optimizers of the same class stepped only when their parameter has a gradient."""
    import holocron_amd as h
    cls = getattr(h.optim, name)
    torch.manual_seed(0)
    a0, b0 = torch.randn(300, device="cuda"), torch.randn(17, 5, device="cuda")
    pa, pb = a0.clone().requires_grad_(), b0.clone().requires_grad_()
    qa, qb = a0.clone().requires_grad_(), b0.clone().requires_grad_()
    joint = cls([pa, pb], lr=1e-2)
    sa, sb = cls([qa], lr=1e-2), cls([qb], lr=1e-2)
    for it in range(6):
        ga, gb = torch.randn_like(a0), torch.randn_like(b0)
        skip_b = it in (1, 2, 4)
        pa.grad, qa.grad = ga.clone(), ga.clone()
        pb.grad, qb.grad = (None, None) if skip_b else (gb.clone(), gb.clone())
        joint.step()
        sa.step()
        if not skip_b:
            sb.step()
    torch.cuda.synchronize()
    assert joint.state[pa]["step"] == 6 and joint.state[pb]["step"] == 3
    assert torch.allclose(pa, qa, rtol=1e-6, atol=1e-7) and torch.allclose(pb, qb, rtol=1e-6, atol=1e-7)


def test_scout_nan_coherence_skips_outer_update():
    import holocron_amd as h
    w = torch.randn(64, device="cuda").requires_grad_()
    frozen = torch.randn(8, device="cuda").requires_grad_()      # zero gradient: its updates are all 0 -> std / max_dev = 0 / 0
    base = torch.optim.SGD([w, frozen], lr=0.1)
    opt = h.optim.Scout(base, sync_rate=0.5, sync_period=3)
    for _ in range(3):
        w.grad = torch.randn_like(w)
        frozen.grad = torch.zeros_like(frozen)
        opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(w).all() and torch.isfinite(frozen).all()


def test_batched_nms_equals_nms_problem_by_problem():
    """hc_nms_sorted_batched: the kept rows of every problem are those of hc_nms_sorted on that problem alone (bit-exact index work),
    for ragged sizes incl. empty and single-box problems and one larger than a 64-box block row."""
    from holocron_amd.ops.boxes import batched_nms_sorted, nms
    g = torch.Generator().manual_seed(11)
    sizes = [0, 1, 63, 64, 65, 700, 0, 129, 2500]
    boxes, offs = [], [0]
    for n in sizes:
        c = torch.rand((n, 2), generator=g)
        wh = torch.rand((n, 2), generator=g) * 0.3 + 0.02
        boxes.append(torch.cat([c, c + wh], 1))
        offs.append(offs[-1] + n)
    allb = torch.cat(boxes, 0).cuda()
    off = torch.tensor(offs, dtype=torch.int32).cuda()
    keep, nkeep = batched_nms_sorted(allb, off, sizes, 0.5)
    nk = nkeep.cpu().tolist()
    for p, n in enumerate(sizes):
        # already "sorted": equal scores -> nms keeps the given order (stable sort)
        want = nms(allb[offs[p]:offs[p + 1]], torch.ones((n,), device="cuda"), 0.5)
        assert nk[p] == want.numel(), (p, n, nk[p], want.numel())
        assert torch.equal(keep[offs[p]:offs[p] + nk[p]].long(), want)


def test_batched_nms_bit_exact_vs_oracle_at_eval_bench_sizes():
    """hc_nms_sorted_batched - the kernel `bench_yolov4.py --eval` times - against `oracle/tv_ops.nms` (the
    restated torchvision algorithm) DIRECTLY, not through hc_nms_sorted: the 3 x 16 problems of a 608^2 batch of 16 at their maximum
    sizes (76^2 x 3 = 17 328, 38^2 x 3 = 4 332, 19^2 x 3 = 1 083 candidate boxes), clustered so that suppression chains are long;
    keep lists are compared index for index."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import tv_ops
    from holocron_amd.ops.boxes import batched_nms_sorted
    g = torch.Generator().manual_seed(23)
    sizes = [n for _ in range(16) for n in (17328, 4332, 1083)]
    boxes, offs = [], [0]
    for n in sizes:
        centres = torch.rand((40, 2), generator=g)                       # 40 objects per image: boxes cluster around them
        c = centres[torch.randint(0, 40, (n,), generator=g)] + torch.randn((n, 2), generator=g) * 0.02
        wh = torch.rand((n, 2), generator=g) * 0.15 + 0.03
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        offs.append(offs[-1] + n)
    allb = torch.cat(boxes, 0)
    keep, nkeep = batched_nms_sorted(allb.cuda(), torch.tensor(offs, dtype=torch.int32).cuda(), sizes, 0.5)
    keep, nk = keep.cpu().long(), nkeep.cpu().tolist()
    total = 0
    for p, n in enumerate(sizes):
        want = tv_ops.nms(allb[offs[p]:offs[p + 1]], torch.ones((n,)), 0.5)      # equal scores: stable order = the given order
        assert nk[p] == want.numel(), (p, n, nk[p], want.numel())
        assert torch.equal(keep[offs[p]:offs[p] + nk[p]], want), p
        total += want.numel()
    assert 0 < total < sum(sizes) // 4                                   # suppression really happened
