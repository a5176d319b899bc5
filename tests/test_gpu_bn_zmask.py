"""hc_rep_bwd_reduce_z / hc_rep_bwd_apply_z recompute the ReLU mask from the pre-activation instead of reading `out`
(reference semantics: the ReLU of RepBlock.forward, holocron/models/classification/repvgg.py:71-73, under autograd).
They must give the SAME BITS as the `out`-reading kernels on what hc_rep_apply produced: same fma chain, same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 48, 56, True), (8, 96, 28, False), (256, 192, 14, True), (4, 1280, 7, True)])
def test_zmask_bit_identical(shape):
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    N, Cc, H, ident = shape
    lib = _lib.load()
    S = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(3)
    t = [cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda", generator=g)) for _ in range(4)]   # g, y3, y1, x
    coef = torch.randn(4, Cc, device="cuda", generator=g)
    coef[3] *= 0.1
    bc = torch.randn(9, Cc, device="cuda", generator=g)
    p = lambda x: None if x is None else x.data_ptr()
    npix = N * H * H
    x = t[3] if ident else None
    out = torch.empty_like(t[0])
    assert lib.hc_rep_apply(p(t[1]), p(t[2]), p(x), p(coef), p(out), None, npix, Cc, 1, S) == 0
    R = _lib.stat_replicas()
    red_o, red_z = torch.zeros(R, 4, Cc, device="cuda"), torch.zeros(R, 4, Cc, device="cuda")
    _lib.set_deterministic(True)     # one writer per replica slot: the two reductions are comparable bit for bit
    try:
        R = _lib.stat_replicas()
        red_o, red_z = torch.zeros(R, 4, Cc, device="cuda"), torch.zeros(R, 4, Cc, device="cuda")
        assert lib.hc_rep_bwd_reduce(p(t[0]), p(out), p(t[1]), p(t[2]), p(x), p(red_o), npix, Cc, S) == 0
        assert lib.hc_rep_bwd_reduce_z(p(t[0]), p(coef), 1, p(t[1]), p(t[2]), p(x), p(red_z), npix, Cc, S) == 0
    finally:
        _lib.set_deterministic(False)
    outs = []
    for z in (False, True):
        dy3, dy1 = torch.empty_like(t[0]), torch.empty_like(t[0])
        dxid = torch.empty_like(t[0]) if ident else None
        if z:
            rc = lib.hc_rep_bwd_apply_z(p(t[0]), p(coef), 1, p(t[1]), p(t[2]), p(x), p(bc), p(dy3), p(dy1), p(dxid), npix, Cc, S)
        else:
            rc = lib.hc_rep_bwd_apply(p(t[0]), p(out), p(t[1]), p(t[2]), p(x), p(bc), p(dy3), p(dy1), p(dxid), npix, Cc, S)
        assert rc == 0
        outs.append((dy3, dy1, dxid))
    torch.cuda.synchronize()
    assert (out.float() > 0).float().mean().item() > 0.2 and (out.float() == 0).float().mean().item() > 0.2   # both mask values occur
    assert torch.equal(red_o, red_z)
    for a, b in zip(outs[0], outs[1]):
        if a is not None:
            assert torch.equal(a, b)
    # act = 0: no mask at all
    dy3, dy1 = torch.empty_like(t[0]), torch.empty_like(t[0])
    dxid = torch.empty_like(t[0]) if ident else None
    assert lib.hc_rep_bwd_apply_z(p(t[0]), p(coef), 0, p(t[1]), p(t[2]), p(x), p(bc), p(dy3), p(dy1), p(dxid), npix, Cc, S) == 0
    ones = torch.ones_like(out)
    r3, r1 = torch.empty_like(t[0]), torch.empty_like(t[0])
    rid = torch.empty_like(t[0]) if ident else None
    assert lib.hc_rep_bwd_apply(p(t[0]), p(ones), p(t[1]), p(t[2]), p(x), p(bc), p(r3), p(r1), p(rid), npix, Cc, S) == 0
    torch.cuda.synchronize()
    assert torch.equal(dy3, r3) and torch.equal(dy1, r1)
