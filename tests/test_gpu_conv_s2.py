"""MI355X: the stride-2 RepBlock forward kernel (csrc/conv_s2.hip, hc_conv_s2_fwd) against torch-CPU fp32 ``F.conv2d`` - the two
convs of a stride-2 RepBlock (holocron/models/classification/repvgg.py:57-60) and their BatchNorm batch statistics - on small
batches (the full-size launches are in test_gpu_fullsize_layers.py), both rows-per-workgroup variants, ragged statistics slots."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("rsel", [0, 1])
@pytest.mark.parametrize("cfg", [(3, 48, 224), (48, 48, 112), (48, 96, 56)], ids=["stem", "48-48@112", "48-96@56"])
def test_conv_s2_forward_and_statistics(cfg, rsel):
    from holocron_amd import _lib
    from holocron_amd.nn import repblock_op as rb
    from holocron_amd.ops import conv as cv
    cin, cout, H = cfg
    N = 3
    g = torch.Generator().manual_seed(100 + cin + cout)
    x = bf16r(torch.rand((N, cin, H, H), generator=g) - 0.3)
    w3 = bf16r(torch.randn((cout, cin, 3, 3), generator=g) * 0.2)
    w1 = bf16r(torch.randn((cout, cin, 1, 1), generator=g) * 0.5)
    dev = torch.device("cuda:0")
    st = rb.RepState(2, False)
    geom = (N, cin, H, H, cout)
    assert st.s2_desc(*geom) is not None, "the stride-2 row kernel must take these shapes"
    xg = x.to(dev)
    src = xg if cin == 3 else cv.to_cl_bf16(xg)
    stats = torch.zeros((2, _lib.stat_replicas(), 2, cout), device=dev)
    os.environ["HC_CONV_S2_R"] = str(rsel)
    try:
        y3, y1 = rb.block_convs_forward(st, src, w3.to(dev), w1.to(dev), geom, stats, cin if cin == 3 else None)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("HC_CONV_S2_R", None)
    c3, c1 = F.conv2d(x, w3, None, 2, 1), F.conv2d(x, w1, None, 2, 0)
    e3, e1 = rel_l2(y3.float().cpu(), c3), rel_l2(y1.float().cpu(), c1)
    assert e3 < 2e-3 and e1 < 2e-3, (cfg, e3, e1)          # one bf16 store of an fp32 result (1.65e-3)
    for st_, ref in ((stats[0], c3), (stats[1], c1)):
        s = st_.double().sum(0).cpu()
        r = ref.double()
        cnt = r.numel() / r.shape[1]
        s1, s2 = r.sum((0, 2, 3)), (r * r).sum((0, 2, 3))
        assert float(((s[0] - s1).abs() / torch.sqrt(s2 * cnt)).max()) < 2e-4
        assert rel_l2(s[1], s2) < 2e-4


@pytest.mark.parametrize("cfg", [(48, 48, 112), (48, 96, 56)], ids=["48-48@112", "48-96@56"])
def test_conv_s2_data_gradient(cfg):
    """hc_conv_s2_dgrad against torch.nn.grad.conv2d_input of both convs (the sum autograd forms in the reference block)."""
    from holocron_amd.nn import repblock_op as rb
    from holocron_amd.ops import conv as cv
    cin, cout, H = cfg
    N = 3
    g = torch.Generator().manual_seed(200 + cin + cout)
    w3 = bf16r(torch.randn((cout, cin, 3, 3), generator=g) * 0.2)
    w1 = bf16r(torch.randn((cout, cin, 1, 1), generator=g) * 0.5)
    dy3 = bf16r(torch.randn((N, cout, H // 2, H // 2), generator=g))
    dy1 = bf16r(torch.randn((N, cout, H // 2, H // 2), generator=g))
    dev = torch.device("cuda:0")
    st = rb.RepState(2, False)
    geom = (N, cin, H, H, cout)
    st.descs(*geom)
    assert st.s2_dgrad
    dx = rb.block_dgrad(st, cv.to_cl_bf16(dy3.to(dev)), cv.to_cl_bf16(dy1.to(dev)), None, w3.to(dev), w1.to(dev), geom)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_input((N, cin, H, H), w3, dy3, 2, 1) + torch.nn.grad.conv2d_input((N, cin, H, H), w1, dy1, 2, 0)
    e = rel_l2(dx.float().cpu(), ref)
    assert e < 2e-3, (cfg, e)


def test_conv_s2_stem_weight_gradient():
    """hc_conv_s2_stem_wgrad (both weight gradients of the stem from the fp32 image batch) against torch.nn.grad.conv2d_weight."""
    from holocron_amd.nn import repblock_op as rb
    from holocron_amd.ops import conv as cv
    N, H, cout = 5, 224, 48
    g = torch.Generator().manual_seed(300)
    x = bf16r(torch.rand((N, 3, H, H), generator=g) - 0.4)
    w3 = torch.randn((cout, 3, 3, 3), generator=g)
    w1 = torch.randn((cout, 3, 1, 1), generator=g)
    dy3 = bf16r(torch.randn((N, cout, H // 2, H // 2), generator=g))
    dy1 = bf16r(torch.randn((N, cout, H // 2, H // 2), generator=g))
    dev = torch.device("cuda:0")
    st = rb.RepState(2, False)
    geom = (N, 3, H, H, cout)
    dw3, dw1 = rb.block_wgrad(st, x.to(dev), cv.to_cl_bf16(dy3.to(dev)), cv.to_cl_bf16(dy1.to(dev)), w3.to(dev), w1.to(dev), geom, 3)
    torch.cuda.synchronize()
    r3 = torch.nn.grad.conv2d_weight(x, w3.shape, dy3, 2, 1)
    r1 = torch.nn.grad.conv2d_weight(x, w1.shape, dy1, 2, 0)
    e3, e1 = rel_l2(dw3.cpu(), r3), rel_l2(dw1.cpu(), r1)
    assert e3 < 2e-4 and e1 < 2e-4, (e3, e1)
    # bit-reproducible (fixed-order slab reduction)
    dw3b, dw1b = rb.block_wgrad(st, x.to(dev), cv.to_cl_bf16(dy3.to(dev)), cv.to_cl_bf16(dy1.to(dev)), w3.to(dev), w1.to(dev), geom, 3)
    assert torch.equal(dw3, dw3b) and torch.equal(dw1, dw1b)


def test_conv_s2_is_what_the_model_runs():
    """repvgg_a0's three front stride-2 blocks go through hc_conv_s2_fwd in a training step (and the step still matches the
    reference-generated goldens: tests/test_gpu_repvgg.py runs the same blocks at fixture size on the gather-conv path)."""
    import holocron_amd as h
    torch.manual_seed(0)
    m = h.models.repvgg_a0(num_classes=10).cuda().train()
    x = torch.rand((2, 3, 224, 224), device="cuda")
    out = m(x)
    out.float().sum().backward()
    torch.cuda.synchronize()
    front = [m.features[0][0], m.features[1][0], m.features[2][0]]
    assert all(b._hc.s2 for b in front) and all(b._hc.s2_dgrad for b in front[1:])
    assert not m.features[3][0]._hc.s2                      # 96 -> 192 @ 28 stays on the gather-conv
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def _stem_fused_setup(N, seed):
    import ctypes as C
    from holocron_amd import _lib
    from holocron_amd.nn import repblock_op as rb
    g = torch.Generator().manual_seed(seed)
    x = bf16r(torch.rand((N, 3, 224, 224), generator=g) - 0.3)
    w3 = bf16r(torch.randn((48, 3, 3, 3), generator=g) * 0.3)
    w1 = bf16r(torch.randn((48, 3, 1, 1), generator=g) * 0.6)
    dev = torch.device("cuda:0")
    st = rb.RepState(2, False)
    xg = x.to(dev)
    d = rb.stem_fused_desc(st, xg, w3.to(dev), w1.to(dev), (N, 3, 224, 224, 48))
    assert d is not None, "the fused stem kernels must take 3 -> 48 @ 224"
    keep = (st, xg)                                           # the descriptor holds raw pointers
    return g, x, w3, w1, d, keep, _lib, C


@pytest.mark.parametrize("N", [2, 9])
def test_stem_fused_passes_vs_fp32_cpu(N):
    """The stem block fused with its BatchNorm passes (hc_stem_stats / _apply / _bwd: y3 and y1 are recomputed from the image in every
    pass, never stored; the backward is ONE pass through G = dz^T X and the Gram matrix of the conv windows) against torch-CPU fp32 of the
    reference expressions (repvgg.py:57-60,71-73: two convs, two training-mode BatchNorm2d, sum, ReLU, and their autograd) - each launch
    in isolation.
    N = 9: more tiles than one pass of the persistent grids' XCD runs is not needed, but ragged runs over the 8 XCDs are."""
    g, x, w3, w1, d, keep, _lib, C = _stem_fused_setup(N, 400 + N)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    c3, c1 = F.conv2d(x, w3, None, 2, 1), F.conv2d(x, w1, None, 2, 0)
    R = _lib.stat_replicas()
    # ---- statistics
    stats = torch.zeros((2, R, 2, 48), device=dev)
    _lib.check(lib.hc_stem_stats(C.byref(d), stats[0].data_ptr(), stats[1].data_ptr(), stream), "hc_stem_stats")
    for st_, ref in ((stats[0], c3), (stats[1], c1)):
        s = st_.double().sum(0).cpu()
        r = ref.double()
        cnt = r.numel() / r.shape[1]
        s1, s2 = r.sum((0, 2, 3)), (r * r).sum((0, 2, 3))
        assert float(((s[0] - s1).abs() / torch.sqrt(s2 * cnt)).max()) < 2e-4
        assert rel_l2(s[1], s2) < 2e-4
    # ---- apply: out = relu(a3 c3 + a1 c1 + shift), one bf16 store
    coef = torch.zeros((4, 48))
    coef[0], coef[1], coef[3] = torch.rand(48, generator=g) + 0.5, torch.rand(48, generator=g) - 0.5, torch.randn(48, generator=g) * 0.3
    V = lambda t: t.view(1, -1, 1, 1)
    z = V(coef[0]) * c3 + V(coef[1]) * c1 + V(coef[3])
    cg = coef.to(dev)
    for act in (1, 0):
        out = torch.empty((N, 112, 112, 48), dtype=torch.bfloat16, device=dev)
        ost = torch.zeros((R, 2, 48), device=dev) if act else None
        _lib.check(lib.hc_stem_apply(C.byref(d), cg.data_ptr(), act, out.data_ptr(), None if ost is None else ost.data_ptr(), stream),
                   "hc_stem_apply")
        ref = F.relu(z) if act else z
        e = rel_l2(out.float().cpu().permute(0, 3, 1, 2), ref)
        assert e < 2e-3, (act, e)                             # the floor of one bf16 store (1.65e-3)
        if ost is not None:                                   # statistics of the ROUNDED output, as the next block's identity BN reads it
            o = out.double().cpu()
            s = ost.double().sum(0).cpu()
            assert rel_l2(s[0], o.sum((0, 1, 2))) < 1e-5 and rel_l2(s[1], (o * o).sum((0, 1, 2))) < 1e-5
    # ---- backward, one pass: BatchNorm parameter gradients and both conv weight gradients against fp32 autograd of the block
    # expression on the CPU (BatchNorm backward through batch statistics; `save` = the true batch means / inverse deviations, `coef`
    # the matching forward affine, so that the whole chain is the reference's)
    gam3, gam1 = torch.rand(48, generator=g) + 0.5, torch.rand(48, generator=g) + 0.5
    bet3, bet1 = torch.randn(48, generator=g) * 0.2, torch.randn(48, generator=g) * 0.2
    gr = bf16r(torch.rand((N, 48, 112, 112), generator=g) + 0.5)
    gg = gr.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    w3r, w1r = w3.clone().requires_grad_(True), w1.clone().requires_grad_(True)
    g3r, g1r, b3r, b1r = (t.clone().requires_grad_(True) for t in (gam3, gam1, bet3, bet1))
    y3r, y1r = F.conv2d(x, w3r, None, 2, 1), F.conv2d(x, w1r, None, 2, 0)
    zr = F.batch_norm(y3r, None, None, g3r, b3r, True, 0.1, 1e-5) + F.batch_norm(y1r, None, None, g1r, b1r, True, 0.1, 1e-5)
    (F.relu(zr) * gr).sum().backward()
    m3, v3 = c3.mean((0, 2, 3)), c3.var((0, 2, 3), unbiased=False)
    m1, v1 = c1.mean((0, 2, 3)), c1.var((0, 2, 3), unbiased=False)
    i3, i1 = torch.rsqrt(v3 + 1e-5), torch.rsqrt(v1 + 1e-5)
    coef = torch.zeros((4, 48))
    coef[0], coef[1] = gam3 * i3, gam1 * i1
    coef[3] = (bet3 - coef[0] * m3) + (bet1 - coef[1] * m1)
    save = torch.zeros((6, 48))
    save[0], save[1], save[2], save[3] = m3, i3, m1, i1
    cg, sg = coef.to(dev), save.to(dev)
    ws = torch.empty((int(lib.hc_stem_bwd_ws_bytes()),), dtype=torch.uint8, device=dev)
    dw3, dw1 = torch.empty((48, 3, 3, 3), device=dev), torch.empty((48, 3, 1, 1), device=dev)
    dgb = torch.empty((4, 48), device=dev)
    w3g, w1g, g3g, g1g = w3.to(dev), w1.to(dev), gam3.to(dev), gam1.to(dev)
    b = _lib.StemBwdDesc()
    b.coef, b.g, b.save, b.gamma3, b.gamma1, b.w3, b.w1 = (t.data_ptr() for t in (cg, gg, sg, g3g, g1g, w3g, w1g))
    b.dgamma3, b.dbeta3, b.dgamma1, b.dbeta1 = (dgb[i].data_ptr() for i in range(4))
    b.dw3, b.dw1, b.ws, b.act, b.frozen = dw3.data_ptr(), dw1.data_ptr(), ws.data_ptr(), 1, 0

    def run(acc):
        b.accumulate = acc
        _lib.check(lib.hc_stem_bwd(C.byref(d), C.byref(b), stream), "hc_stem_bwd")
        torch.cuda.synchronize()
    run(0)
    errs = {"dw3": rel_l2(dw3.cpu(), w3r.grad), "dw1": rel_l2(dw1.cpu(), w1r.grad), "dgamma3": rel_l2(dgb[0].cpu(), g3r.grad),
            "dbeta3": rel_l2(dgb[1].cpu(), b3r.grad), "dgamma1": rel_l2(dgb[2].cpu(), g1r.grad), "dbeta1": rel_l2(dgb[3].cpu(), b1r.grad)}
    print(N, errs)
    # fp32 sums of exact bf16 products against fp32 autograd; pre-activations within fp32 rounding of zero flip a mask bit here and there
    assert all(v < 1e-3 for v in errs.values()), errs
    a3, a1, ag = dw3.clone(), dw1.clone(), dgb.clone()
    run(0)
    assert torch.equal(dw3, a3) and torch.equal(dw1, a1) and torch.equal(dgb, ag)      # fixed-order slab sums: bit-reproducible
    run(1)
    assert torch.allclose(dw3, 2 * a3, rtol=1e-6, atol=0) and torch.allclose(dgb, 2 * ag, rtol=1e-6, atol=0)
    # eval-mode BatchNorm (frozen statistics): dy = a dz, no centring terms
    b.frozen = 1
    run(0)
    dz = gr * (zr.detach() > 0)
    V = lambda t: t.view(1, -1, 1, 1)
    r3 = torch.nn.grad.conv2d_weight(x, w3.shape, V(coef[0]) * dz, 2, 1)
    r1 = torch.nn.grad.conv2d_weight(x, w1.shape, V(coef[1]) * dz, 2, 0)
    assert rel_l2(dw3.cpu(), r3) < 1e-3 and rel_l2(dw1.cpu(), r1) < 1e-3
    assert rel_l2(dgb[1].cpu(), dz.sum((0, 2, 3))) < 1e-4


def test_stem_fused_block_matches_the_unfused_block():
    """A RepBlock 3 -> 48 at 224 x 224 through RepBlockFn with the fused stem against the same block with HC_STEM_FUSED's unfused
    sequence (conv -> y3 / y1 -> BatchNorm passes): outputs within two bf16 roundings of each other, parameter gradients and running
    statistics within fp32 / flip noise - and the fused path must really be the one that ran."""
    import holocron_amd as h
    from holocron_amd.nn import repblock_op as rb
    torch.manual_seed(5)
    x = torch.rand((4, 3, 224, 224), device="cuda")
    r = (torch.rand((4, 48, 112, 112), device="cuda") + 0.5).to(torch.bfloat16).float()
    res = []
    for fused in (True, False):
        torch.manual_seed(6)
        blk = h.models.RepBlock(3, 48, 2, False).cuda().train()
        real = rb.stem_fused_desc
        taken = []
        if not fused:
            rb.stem_fused_desc = lambda *a, **k: None
        else:
            def counting(*a, **k):
                d = real(*a, **k)
                taken.append(d is not None)
                return d
            rb.stem_fused_desc = counting
        try:
            out = blk(x)
            (out.float() * r).sum().backward()
            torch.cuda.synchronize()
        finally:
            rb.stem_fused_desc = real
        assert not fused or (len(taken) == 2 and all(taken)), taken       # forward and backward both ran on the fused launches
        res.append((out.detach().float(), {n: p.grad.clone() for n, p in blk.named_parameters()},
                    {k: v.clone() for k, v in blk.state_dict().items() if "running" in k}))
    (of, gf, rf), (ou, gu, ru) = res
    assert rel_l2(of, ou) < 6e-3          # measured 4.2e-3: the unfused path rounds y3 and y1 to bf16 before it normalises them
    for n in gf:
        assert rel_l2(gf[n], gu[n]) < (3e-2 if n.endswith("0.weight") else 3e-3), (n, rel_l2(gf[n], gu[n]))
    for k in rf:
        assert rel_l2(rf[k], ru[k]) < 1e-6, k


def test_stem_fused_eval_mode_and_frozen_backward_match_the_unfused_block():
    """Eval-mode BatchNorm (running statistics: no statistics launch, `frozen` backward: dy = a dz) through the fused stem against the
    unfused sequence on the same block - forward within two bf16 roundings, weight gradients within flip noise."""
    import holocron_amd as h
    from holocron_amd.nn import repblock_op as rb
    torch.manual_seed(11)
    x = torch.rand((3, 3, 224, 224), device="cuda")
    r = (torch.rand((3, 48, 112, 112), device="cuda") + 0.5).to(torch.bfloat16).float()
    res = []
    for fused in (True, False):
        torch.manual_seed(12)
        blk = h.models.RepBlock(3, 48, 2, False).cuda()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(torch.randn(48, device="cuda") * 0.05)
                    m.running_var.copy_(torch.rand(48, device="cuda") * 0.05 + 0.02)
        blk.eval()
        real = rb.stem_fused_desc
        if not fused:
            rb.stem_fused_desc = lambda *a, **k: None
        try:
            out = blk(x)
            (out.float() * r).sum().backward()
            torch.cuda.synchronize()
        finally:
            rb.stem_fused_desc = real
        res.append((out.detach().float(), {n: p.grad.clone() for n, p in blk.named_parameters()}))
    (of, gf), (ou, gu) = res
    assert rel_l2(of, ou) < 6e-3
    for n in gf:
        assert rel_l2(gf[n], gu[n]) < 3e-2, (n, rel_l2(gf[n], gu[n]))
