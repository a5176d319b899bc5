"""MI355X tests of the fp8 (OCP e4m3) inference path of the re-parametrised RepVGG (BASELINE config C5)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _fp8r(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def test_fp8_conv_kernel_matches_emulation():
    """One conv on fp8 bytes through hc_conv_gather (ch_mult mode) against fp32 math on the same fp8 values; the only
    differences allowed are requantisation flips at fp8 rounding boundaries (accumulation order)."""
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    from holocron_amd.models.classification.repvgg_fp8 import quantize_weight_fp8
    g = torch.Generator().manual_seed(3)
    for (N, Cin, Cout, H, k, stride) in [(2, 64, 64, 12, 3, 1), (2, 128, 192, 9, 3, 2), (1, 192, 100, 7, 3, 1), (3, 64, 32, 8, 1, 1)]:
        pad = k // 2
        x = _fp8r(torch.randn((N, Cin, H, H), generator=g) * 2)
        w = torch.randn((Cout, Cin, k, k), generator=g) * 0.1
        bias = torch.randn((Cout,), generator=g)
        sx_in, sx_out = 0.5, 0.25
        sw = w.abs().amax(dim=(1, 2, 3)) / 448
        wq = _fp8r(w / sw.view(-1, 1, 1, 1))
        acc = F.conv2d(x, wq, None, stride, pad)
        ref = _fp8r(torch.relu(acc * (sw * sx_in / sx_out).view(1, -1, 1, 1) + (bias / sx_out).view(1, -1, 1, 1)))
        cout_p = (Cout + 63) // 64 * 64
        wpk, swp = quantize_weight_fp8(w.cuda(), Cin, cout_p)
        assert torch.allclose(swp[:Cout].cpu(), sw)
        xq = x.permute(0, 2, 3, 1).contiguous().to(torch.float8_e4m3fn).view(torch.uint8).cuda()
        d = cv.fwd_desc(N, Cin, H, H, cout_p, k, k, stride, pad)
        out = torch.empty((N, d.OH, d.OW, cout_p), dtype=torch.uint8, device="cuda")
        mult = (swp * (sx_in / sx_out)).contiguous()
        badd = torch.zeros((cout_p,), device="cuda")
        badd[:Cout] = (bias / sx_out).cuda()
        d.ch_mult = _lib.ptr(mult)
        cv.launch_conv(d, xq, wpk, out, bias=badd, act=1)
        got = out.cpu().view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2)
        assert float(got[:, Cout:].abs().max()) == 0.0 if cout_p > Cout else True
        got = got[:, :Cout]
        same = float((got == ref).float().mean())
        assert same > 0.99, (N, Cin, Cout, H, k, stride, same)
        # a flip moves a value by one fp8 step (2^-3 relative at most)
        assert float(((got - ref).abs() <= 0.126 * ref.abs().clamp(min=2 ** -6)).float().mean()) == 1.0


def test_repvgg_fp8_inference_matches_emulation_and_reference():
    import holocron_amd as h
    from holocron_amd.models.classification.repvgg_fp8 import Fp8RepVGG
    from oracle import repvgg as orv
    torch.manual_seed(0)
    m = h.models.repvgg_a2(num_classes=10)
    g = torch.Generator().manual_seed(1)
    for mod in m.modules():      # non-trivial BatchNorm statistics so that the re-parametrisation matters
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.data = torch.randn(mod.running_mean.shape, generator=g) * 0.1
            mod.running_var.data = torch.rand(mod.running_var.shape, generator=g) + 0.5
            mod.weight.data = torch.rand(mod.weight.shape, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.1
    m.eval()
    m.reparametrize()
    x = torch.rand((4, 3, 64, 64), generator=g)
    with torch.no_grad():                                   # fp32 inference graph of the reference, on CPU
        hcpu = x
        convs = []
        for stage in m.features:
            for b in stage:
                c = b.branches
                hcpu = torch.relu(F.conv2d(hcpu, c.weight, c.bias, c.stride, 1))
                convs.append((c.weight.detach().clone(), c.bias.detach().clone(), c.stride[0]))
        ref = F.linear(hcpu.flatten(2).mean(2), m.head.weight, m.head.bias)
    m = m.cuda()
    q = Fp8RepVGG(m, x.cuda())
    out = q(x.cuda()).float().cpu()
    emu = orv.forward_fp8_emulated(convs, m.head.weight.detach().cpu(), m.head.bias.detach().cpu(), x, q.input_scale, q.act_scales)
    assert out.shape == (4, 10)
    assert rel_l2(out, emu) < 0.08, rel_l2(out, emu)         # same quantisation points; rounding-boundary flips only
    assert rel_l2(out, ref) < 0.25, rel_l2(out, ref)         # fp8 storage against fp32
    assert rel_l2(emu, ref) < 0.25
    with pytest.raises(ValueError):
        Fp8RepVGG(h.models.repvgg_a0().cuda().eval(), x.cuda())    # not re-parametrised
