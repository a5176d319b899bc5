"""MI355X parity tests of the MobileOne blocks and model (reference: holocron/models/classification/mobileone.py,
tests/test_models_classification.py:48-63)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _scale_only(t):
    return t.dim() == 4 and tuple(t.shape[1:]) == (1, 1, 1)


def _run_block_case(c):
    import holocron_amd as h
    from oracle import mobileone as omo
    cin, cout, K, stride = c["cfg"]
    blk = h.models.MobileOneBlock(cin, cout, K, stride)
    blk.load_state_dict(c["state"])
    blk = blk.cuda().train()
    x = c["x"].cuda().requires_grad_(True)
    out = blk(x)
    assert out.shape == c["out"].shape
    # sharp check: the bf16-emulating oracle on the same inputs
    sd = {"b." + k: v.clone() for k, v in c["state"].items()}
    names = list(c["dparams"])
    leaves = [sd["b." + n].requires_grad_(True) for n in names]
    xe = c["x"].clone().requires_grad_(True)
    oe = omo.block(xe, sd, "b", stride, True, emu=True)
    ge = torch.autograd.grad((oe * c["r"]).sum(), [xe] + leaves)
    assert rel_l2(out.float().cpu(), oe.detach()) < 1e-2, (c["cfg"], rel_l2(out.float().cpu(), oe.detach()))
    assert rel_l2(out.float().cpu(), c["out"]) < 3e-2            # and the fp32 reference itself
    (out.float() * c["r"].cuda()).sum().backward()
    assert rel_l2(x.grad.float().cpu(), ge[0]) < 5e-2, (c["cfg"], "dx", rel_l2(x.grad.float().cpu(), ge[0]))
    params = dict(blk.named_parameters())
    gscale = max(float(gg.abs().max()) for gg in ge[1:])
    for n, gg in zip(names, ge[1:]):
        got = params[n].grad.float().cpu()
        if _scale_only(gg):
            # a per-channel scale in front of BatchNorm: zero gradient except where |w| is so small that eps matters
            # (there it is large, and matches); elsewhere rounding noise - compared on the block's gradient scale
            assert float((got - gg).abs().max()) < 0.02 * max(gscale, float(gg.abs().max())) + 1e-3, (c["cfg"], n)
            continue
        e = rel_l2(got, gg)
        assert e < 6e-2, (c["cfg"], n, e)
        # vs the fp32 reference: bf16 storage of the branch planes; the 3-channel stem's BatchNorm gradients are
        # cancelling sums of three numbers
        assert rel_l2(got, c["dparams"][n]) < (0.3 if got.numel() < 16 else 0.15), (c["cfg"], n)
    for k, v in c["state_after"].items():
        if "running" in k:
            assert rel_l2(blk.state_dict()[k].cpu(), v) < 1e-2, k
        else:
            assert int(blk.state_dict()[k]) == int(v), k
    # eval mode (running statistics) and the re-parametrised block
    blk.load_state_dict(c["state"])
    blk.eval()
    with torch.no_grad():
        oe = blk(c["x"].cuda())
        assert rel_l2(oe.float().cpu(), c["out_eval"]) < 2e-2, (c["cfg"], "eval")
        blk.reparametrize()
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in blk.modules())
        for k, v in c["rep_state"].items():
            assert torch.allclose(blk.state_dict()[k].cpu(), v, rtol=1e-4, atol=1e-5), k
        orp = blk(c["x"].cuda())
        assert rel_l2(orp.float().cpu(), c["out_rep"]) < 2e-2, (c["cfg"], "rep")


def test_mobileone_blocks_match_reference_and_bf16_oracle(golden):
    for c in golden("mobileone.pt")["blocks"]:
        _run_block_case(c)


def test_eval_mode_backward_uses_running_statistics(golden):
    """In eval mode BatchNorm is a fixed affine: the block gradient is the folded convolution's."""
    import holocron_amd as h
    from oracle import mobileone as omo
    c = golden("mobileone.pt")["blocks"][1]
    cin, cout, K, stride = c["cfg"]
    blk = h.models.MobileOneBlock(cin, cout, K, stride)
    blk.load_state_dict(c["state"])
    blk = blk.cuda().eval()
    x = c["x"].cuda().requires_grad_(True)
    out = blk(x)
    (out.float() * c["r"].cuda()).sum().backward()
    sd = {"b." + k: v.clone() for k, v in c["state"].items()}
    xe = c["x"].clone().requires_grad_(True)
    oe = omo.block(xe, sd, "b", stride, False, emu=True)
    (ge,) = torch.autograd.grad((oe * c["r"]).sum(), xe)
    assert rel_l2(out.float().cpu(), oe.detach()) < 1e-2
    assert rel_l2(x.grad.float().cpu(), ge) < 5e-2


def test_mobileone_s0_train_step_and_reparametrize(golden):
    """tests/test_models_classification.py:48-63 (reparametrize keeps the eval output) + one training step against the
    reference's (loose: see the chaos note in DESIGN.md) and the state_dict contract."""
    import holocron_amd as h
    gm = golden("mobileone.pt")["model"]
    torch.manual_seed(gm["seed"])
    m = h.models.mobileone_s0(num_classes=gm["num_classes"]).cuda().train()
    x, t = gm["x"].cuda(), gm["target"].cuda()
    import copy
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        hcur = m2.features[0].forward_padded(x)
        errs = [rel_l2(hcur.float().mean((2, 3)).cpu(), gm["stage_means"][0])]
        for i in (1, 2):
            for blk in m2.features[i]:
                hcur = blk.forward_padded(hcur)
            errs.append(rel_l2(hcur.float().mean((2, 3)).cpu(), gm["stage_means"][i]))
    assert errs[0] < 1e-2 and errs[1] < 3e-2 and errs[2] < 0.15, errs
    logits = m(x)
    assert logits.shape == gm["logits"].shape and logits.dtype == torch.float32
    # SMOKE only beyond the stage means above (4 images, 2 x 2 maps at the end): finiteness and the loss in the right place.  The
    # whole-model PARITY check - every block in situ against the oracle, free-running logits against the emulating oracle with a
    # yardstick - is tests/test_gpu_whole_models.py on the 16 x 128 x 128 fixture
    assert bool(torch.isfinite(logits).all())
    loss = F.cross_entropy(logits, t)
    assert abs(float(loss.detach()) - float(gm["loss"])) < 0.25 * max(1.0, float(gm["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params.values())
    for k, v in gm["running_sample"].items():
        assert rel_l2(m.state_dict()[k].cpu(), v) < 0.2, k
    # reparametrize: no BatchNorm left, same eval output
    m.eval()
    with torch.no_grad():
        out = m(x)
        m.reparametrize()
        for mod in m.modules():
            assert not isinstance(mod, torch.nn.BatchNorm2d)
        rep = m(x)
    assert rel_l2(rep.cpu(), out.cpu()) < 6e-2, rel_l2(rep.cpu(), out.cpu())
    with pytest.raises(NotImplementedError):
        m.train()
        m(x.requires_grad_(True))


@pytest.mark.parametrize("arch", ["mobileone_s0", "mobileone_s1", "mobileone_s2", "mobileone_s3"])
def test_mobileone_full_size_shapes(arch):
    """tests/test_models_classification.py:9-26 (_test_classification_model) at the reference's 224^2 input: output shape,
    gradients on every parameter, and the folded model reproducing the multi-branch eval output."""
    import holocron_amd as h
    torch.manual_seed(0)
    m = h.models.__dict__[arch](num_classes=10).cuda().train()
    x = torch.rand((4, 3, 224, 224), device="cuda")
    out = m(x)
    assert out.shape == (4, 10) and bool(torch.isfinite(out).all())
    out.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    m.eval()
    with torch.no_grad():
        a = m(x)
        m.reparametrize()
        b = m(x)
    assert rel_l2(b.cpu(), a.cpu()) < 3e-2
