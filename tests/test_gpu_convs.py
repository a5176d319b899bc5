"""MI355X parity tests of SlimConv2d and NormConv2d (reference: holocron/nn/modules/conv.py:55-147,262-370)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_slimconv2d_matches_reference(golden):
    import holocron_amd as h
    for c in golden("convs.pt")["slim"]:
        cin, k, stride, pad, r = c["cfg"]
        m = h.nn.SlimConv2d(cin, k, stride=stride, padding=pad, r=r)
        m.load_state_dict(c["state"])
        m = m.cuda().train()
        x = c["x"].cuda().requires_grad_(True)
        out = m(x)
        assert out.shape == c["out"].shape                                  # tests/test_nn_conv.py:29: 3C/4 channels
        assert rel_l2(out.float().cpu(), c["out"]) < 6e-3, (c["cfg"], rel_l2(out.float().cpu(), c["out"]))
        (out.float() * c["r"].cuda()).sum().backward()
        assert rel_l2(x.grad.float().cpu(), c["dx"]) < 2e-2, (c["cfg"], rel_l2(x.grad.float().cpu(), c["dx"]))
        params = dict(m.named_parameters())
        for n, gg in c["dparams"].items():
            if n == "fc1.bias":   # a conv bias in front of a training-mode BatchNorm: no gradient (round-off in the reference)
                assert float(gg.abs().max()) < 1e-3 * float(c["dparams"]["fc1.weight"].abs().max()) + 1e-6
                assert float(params[n].grad.abs().max()) < 1e-3 * float(c["dparams"]["fc1.weight"].abs().max()) + 1e-6
                continue
            if float(gg.abs().max()) < 1e-5:
                continue
            if c["x"].shape[0] == 2 and n in ("fc1.weight", "bn.weight"):
                continue   # BatchNorm over 2 pooled samples outputs +-gamma + beta whatever fc1 does: these gradients are round-off
            e = rel_l2(params[n].grad.float().cpu(), gg)
            # the gate network sees batch statistics over N samples of a pooled vector: its gradients are the
            # least well conditioned part
            assert e < (0.12 if n.startswith(("fc", "bn")) else 2e-2), (c["cfg"], n, e)
        for kk, v in c["state_after"].items():
            assert rel_l2(m.state_dict()[kk].cpu(), v) < 2e-2, kk


def test_normconv2d_matches_reference(golden):
    import holocron_amd as h
    for c in golden("convs.pt")["norm"]:
        cin, cout, k, stride, pad, mode = c["cfg"]
        m = h.nn.NormConv2d(cin, cout, k, stride=stride, padding=pad, padding_mode=mode)
        m.load_state_dict(c["state"])
        m = m.cuda()
        out = m(c["x"].cuda())
        assert out.shape == c["out"].shape
        assert rel_l2(out.float().cpu(), c["out"]) < 6e-3, (c["cfg"], rel_l2(out.float().cpu(), c["out"]))
        (out.float() * c["r"].cuda()).sum().backward()
        assert rel_l2(m.weight.grad.cpu(), c["dw"]) < 1.5e-2, (c["cfg"], rel_l2(m.weight.grad.cpu(), c["dw"]))
        assert rel_l2(m.bias.grad.cpu(), c["db"]) < 5e-3
    # like the reference (in-place normalisation of the unfolded input) there is no gradient w.r.t. the input
    x = c["x"].cuda().requires_grad_(True)
    with pytest.raises(RuntimeError):
        m(x).float().sum().backward()
    # tests/test_nn_conv.py:16-18
    assert h.nn.NormConv2d(8, 16, 3, padding=1).cuda()(torch.rand(2, 8, 16, 16).cuda()).shape == (2, 16, 16, 16)
    assert h.nn.NormConv2d(8, 16, 3, padding=1, padding_mode="reflect").cuda()(torch.rand(2, 8, 16, 16).cuda()).shape == (2, 16, 16, 16)
