"""GPU: the fused RepBlock weight-gradient kernel (csrc/conv_wgrad_rep.hip: dW3 and dW1 of up to 16 same-shaped blocks from one
launch) against torch-CPU fp32 (`torch.nn.grad.conv2d_weight`, i.e. aten::convolution_backward) on bf16-representable operands.
fp32 outputs: rel-L2 <= 2e-4.  The full-size shapes run in test_gpu_fullsize_layers.py (block_wgrad routes through this kernel)."""
import ctypes as C

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


CASES = [
    # N, Cin, H, W, Cout, stride, jobs
    (2, 48, 14, 14, 48, 1, 1),
    (3, 48, 10, 14, 48, 1, 2),       # non-square, two blocks in one launch
    (2, 48, 16, 16, 96, 2, 1),
    (2, 48, 15, 13, 48, 2, 1),       # odd sizes with stride 2
    (2, 96, 12, 12, 96, 1, 3),       # (6, 6) tile: 96 ci x 96 co x 10 taps, eight waves (round 5)
    (2, 96, 28, 28, 192, 2, 1),      # (6, 6), stride 2
    (5, 192, 14, 14, 192, 1, 2),     # (6, 6), 2 x 2 tiles
    (3, 192, 9, 11, 96, 1, 1),       # (6, 6), odd map, 2 x 1 tiles
    (2, 96, 12, 12, 144, 1, 2),      # (6, 3) tile of rounds 2-4: 96 | Cin, 48 | Cout only
    (2, 192, 14, 14, 48, 1, 1),      # (6, 3), 2 x 1 tiles
    (2, 64, 9, 9, 128, 1, 1),        # (4, 4) tile
    (1, 128, 56, 56, 64, 1, 1),
    (9, 48, 7, 7, 48, 1, 1),         # steps that span several images
    (2, 48, 112, 112, 48, 1, 1),     # one row per step
    (1, 48, 112, 112, 48, 2, 1),
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,stride,jobs", CASES)
def test_rep_wgrad_matches_cpu(N, Cin, H, W, Cout, stride, jobs):
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    lib = _lib.load()
    key = (N, Cin, H, W, Cout, stride)
    assert cv._WREP.supported(key), key
    plan = (C.c_int32 * 8)()
    assert lib.hc_rep_wgrad_plan(C.byref(cv._WREP._desc(key, jobs)), plan) == 0
    g = torch.Generator().manual_seed(N * 100 + Cin + H + Cout)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    work, refs = [], []
    for _ in range(jobs):
        x = bf16r(torch.randn((N, Cin, H, W), generator=g))
        dy3 = bf16r(torch.randn((N, Cout, OH, OW), generator=g))
        dy1 = bf16r(torch.randn((N, Cout, OH, OW), generator=g))
        dw3 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
        dw1 = torch.full((Cout, Cin, 1, 1), float("nan"), device="cuda")
        work.append((cv.to_cl_bf16(x.cuda()), cv.to_cl_bf16(dy3.cuda()), cv.to_cl_bf16(dy1.cuda()), dw3, dw1))
        refs.append((torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy3, stride, 1),
                     torch.nn.grad.conv2d_weight(x, (Cout, Cin, 1, 1), dy1, stride, 0)))
    ptrs = [(w[0], w[1], w[2], w[3].data_ptr(), w[4].data_ptr()) for w in work]
    cv._WREP.launch(key, ptrs)
    torch.cuda.synchronize()
    for (_, _, _, dw3, dw1), (r3, r1) in zip(work, refs):
        e3, e1 = rel_l2(dw3.cpu(), r3), rel_l2(dw1.cpu(), r1)
        assert e3 < 2e-4 and e1 < 2e-4, (key, list(plan), e3, e1)
    # a second launch into the same buffers gives the same bits (fixed-order split reduction, no atomics)
    first = [(w[3].clone(), w[4].clone()) for w in work]
    cv._WREP.launch(key, ptrs)
    torch.cuda.synchronize()
    for (a3, a1), w in zip(first, work):
        assert torch.equal(a3, w[3]) and torch.equal(a1, w[4])


def test_rep_wgrad_deferred_in_backward_matches_immediate():
    """inside autograd the launches are queued and grouped (ops/conv.py _RepWgradQueue); same numbers as launching in place"""
    import holocron_amd as h
    from holocron_amd.ops import conv as cv
    torch.manual_seed(0)
    cfg = dict(num_blocks=[1, 2, 1, 1, 1], planes=[48, 48, 96, 96, 64], width_multiplier=1, final_width_multiplier=1)
    m = h.models.RepVGG(**cfg).cuda().train()
    x = torch.rand(4, 3, 64, 64, device="cuda")

    def grads(defer):
        cv._WREP.enabled = defer
        for p in m.parameters():
            p.grad = None
        m(x).float().square().mean().backward()
        torch.cuda.synchronize()
        assert not cv._WREP.jobs
        return [p.grad.detach().clone() for p in m.parameters()]

    try:
        a = grads(True)
        b = grads(False)
    finally:
        cv._WREP.enabled = True
    for (n, _), u, v in zip(m.named_parameters(), a, b):
        if u.dim() == 4:       # run to run the statistics atomics move activations by 1 bf16 ulp: not bit-equal, but close
            assert rel_l2(u, v) < 2e-2, n
