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
