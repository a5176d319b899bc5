"""GPU: the fused conv -> BN -> LeakyReLU (+residual) path (DarkNet-53 rows of SURVEY §8a: A1, A6)."""
import pytest
import torch

from conftest import close_frac, rel_l2

pytestmark = pytest.mark.gpu


def test_resblock_matches_reference_and_bf16_oracle(golden):
    import holocron_amd as h
    from oracle import darknet as od
    from oracle import repvgg as orv
    for c in golden("darknet.pt")["resblocks"]:
        planes = c["planes"]
        blk = h.models.ResBlock(planes, planes // 2, torch.nn.LeakyReLU(0.1, inplace=True), torch.nn.BatchNorm2d)
        blk.load_state_dict(c["state"])
        blk = blk.cuda().train()
        x = c["x"].cuda().requires_grad_(True)
        out = blk(x)
        assert rel_l2(out.float().cpu(), c["out"]) < 4e-3                      # vs the fp32 reference
        (out.float() * c["r"].cuda()).sum().backward()
        # vs the oracle with bf16 rounding injected at the tensors the HIP path stores as bf16
        sd = {"b." + k: v.clone() for k, v in c["state"].items()}
        keys = orv.trainable_keys(sd)
        for k in keys:
            sd[k].requires_grad_(True)
        xe = orv.bf16r(c["x"]).requires_grad_(True)
        eout = od.res_block(xe, sd, "b", True, emulate_bf16=True)
        eg = torch.autograd.grad((eout * c["r"]).sum(), [xe] + [sd[k] for k in keys])
        o = out.float().cpu()
        assert float((o == eout.detach()).double().mean()) > 0.995
        assert close_frac(x.grad.float().cpu(), eg[0], 1e-2, 1e-2 * float(eg[0].abs().mean())) > 0.99
        for k, ge in zip(keys, eg[1:]):
            gp = dict(blk.named_parameters())[k[2:]].grad.cpu()
            assert rel_l2(gp, ge) < 1.5e-2, (k, rel_l2(gp, ge))
        st = blk.state_dict()
        for k, v in c["state_after"].items():
            if "running" in k:
                assert torch.allclose(st[k].cpu(), v, rtol=2e-3, atol=2e-3), k
            if k.endswith("num_batches_tracked"):
                assert int(st[k]) == int(v)


def test_darknet_small_train_step(golden):
    import holocron_amd as h
    from oracle import darknet as od
    g = golden("darknet.pt")
    m = h.models.DarknetV3(g["layout"], num_classes=10, stem_channels=g["stem"])
    m.load_state_dict(g["state"])
    m = m.cuda().train()
    logits = m(g["x"].cuda())
    loss = torch.nn.functional.cross_entropy(logits, g["target"].cuda())
    loss.backward()
    assert rel_l2(logits.float().cpu(), g["logits"]) < 3e-2
    assert abs(float(loss) - float(g["loss"])) < 3e-2 * max(1.0, abs(float(g["loss"])))
    # larger batch against the bf16-emulating oracle (cosine of the full gradient)
    torch.manual_seed(3)
    xb = torch.rand(32, 3, 32, 32).to(torch.bfloat16).float()
    tb = torch.randint(0, 10, (32,))
    m2 = h.models.DarknetV3(g["layout"], num_classes=10, stem_channels=g["stem"])
    m2.load_state_dict(g["state"])
    m2 = m2.cuda().train()
    lg = m2(xb.cuda())
    torch.nn.functional.cross_entropy(lg, tb.cuda()).backward()
    sd = {k: v.clone() for k, v in g["state"].items()}
    keys = [k for k in sd if not ("running" in k or "num_batches" in k)]
    for k in keys:
        sd[k].requires_grad_(True)
    el = od.forward(sd, xb, g["layout"], training=True, emulate_bf16=True)
    eg = torch.autograd.grad(torch.nn.functional.cross_entropy(el, tb), [sd[k] for k in keys])
    assert rel_l2(lg.float().cpu(), el.detach()) < 1.5e-2
    fh = torch.cat([dict(m2.named_parameters())[k].grad.flatten().cpu() for k in keys])
    fe = torch.cat([x.flatten() for x in eg])
    assert float(torch.nn.functional.cosine_similarity(fh.double(), fe.double(), dim=0)) > 0.98
    for k in ("features.stem.1.running_var", "features.layers.1.4.conv.4.running_mean"):
        assert torch.allclose(m2.state_dict()[k].cpu(), sd[k].detach(), rtol=5e-3, atol=5e-3), k


@pytest.mark.parametrize("act", ["relu", "leaky", "mish", "silu", "relu6", "hard_mish", "none"])
def test_conv_bn_act_activations_vs_torch(act):
    """every fused activation, forward and backward, against torch autograd on CPU (bf16 tolerances)"""
    import holocron_amd as h
    from holocron_amd.nn.convbn_op import conv_bn_act
    torch.manual_seed(5)
    acts = {"relu": torch.nn.ReLU(), "leaky": torch.nn.LeakyReLU(0.1), "mish": torch.nn.Mish(), "silu": torch.nn.SiLU(),
            "relu6": torch.nn.ReLU6(), "hard_mish": h.nn.HardMish(), "none": None}
    ref_acts = dict(acts, hard_mish=lambda v: 0.5 * v * (v + 2).clamp(0, 2))
    conv = torch.nn.Conv2d(32, 48, 3, stride=2, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(48)
    with torch.no_grad():
        conv.weight.copy_(conv.weight.to(torch.bfloat16).float())
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.5)
    x = torch.randn(4, 32, 13, 13).to(torch.bfloat16).float().requires_grad_(True)
    r = torch.randn(4, 48, 7, 7)
    y = torch.nn.functional.batch_norm(conv(x), None, None, bn.weight, bn.bias, True, 0.1, 1e-5)
    ref = y if ref_acts[act] is None else ref_acts[act](y)
    gr = torch.autograd.grad((ref * r).sum(), [x, conv.weight, bn.weight, bn.bias])
    import copy
    cg, bg = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda().train()
    xg = x.detach().cuda().requires_grad_(True)
    out = conv_bn_act(xg, cg, bg, None if acts[act] is None else acts[act])
    assert rel_l2(out.float().cpu(), ref.detach()) < 5e-3
    (out.float() * r.cuda()).sum().backward()
    assert close_frac(xg.grad.float().cpu(), gr[0], 3e-2, 3e-2 * float(gr[0].abs().mean())) > 0.95
    assert rel_l2(cg.weight.grad.cpu(), gr[1]) < 3e-2
    # d(gamma) is a cancelling sum over only 4*7*7 bf16-stored values per channel
    assert rel_l2(bg.weight.grad.cpu(), gr[2]) < 8e-2 and rel_l2(bg.bias.grad.cpu(), gr[3]) < 3e-2


@pytest.mark.parametrize("act", ["mish", "leaky0.01", "none"])
@pytest.mark.parametrize("cin,cout,k,stride,res", [(64, 64, 3, 1, True), (32, 96, 1, 1, False), (128, 256, 3, 2, False), (3, 32, 3, 1, False)])
def test_inference_unit_is_one_launch_and_matches_fp32_eval(act, cin, cout, k, stride, res):
    """Eval mode under no_grad: conv -> BatchNorm(running statistics) -> activation [+ residual] runs as ONE gather-conv launch with
    the normalisation in its epilogue (round 6).  Checked against torch-CPU fp32 eval of the same modules (models/utils.py:73-84) and
    against the three-launch path the same unit takes with autograd on."""
    import torch.nn.functional as F
    from torch import nn
    from holocron_amd.nn import convbn_op as cb
    from holocron_amd.ops import conv as cv
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
    conv.weight.data = conv.weight.data.to(torch.bfloat16).float()
    bn = nn.BatchNorm2d(cout)
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.5, 0.5)
    bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
    a = {"mish": nn.Mish(), "leaky0.01": nn.LeakyReLU(0.01), "none": None}[act]
    x = torch.randn(3, cin, 20, 20).to(torch.bfloat16).float()
    r = torch.randn(3, cout, 20, 20).to(torch.bfloat16).float() if res else None
    conv.eval(); bn.eval()
    with torch.no_grad():
        ref = bn(conv(x))
        if a is not None:
            ref = a(ref)
        if r is not None:
            ref = ref + r
    conv, bn = conv.cuda(), bn.cuda()
    cv.PROFILE, cv.PROFILE_TAGS = [], []
    try:
        with torch.no_grad():
            got = cb.conv_bn_act(x.cuda(), conv, bn, a, residual=None if r is None else r.cuda())
        torch.cuda.synchronize()
        fams = [p[0] for p in cv.PROFILE]
    finally:
        cv.PROFILE = cv.PROFILE_TAGS = None
    assert [f for f in fams if f != "stem_im2col"] == ["conv_gather"], fams          # no BatchNorm launch
    assert rel_l2(got.float().cpu(), ref) < 3e-3, rel_l2(got.float().cpu(), ref)   # one bf16 store of an fp32 result
    three = cb.conv_bn_act(x.cuda().requires_grad_(True), conv, bn, a, residual=None if r is None else r.cuda())
    assert rel_l2(three.float().cpu(), ref) < 6e-3
    assert rel_l2(got.float().cpu(), three.float().cpu()) < 6e-3


def test_cspstage_split_in_the_concat_buffer_equals_the_three_copy_path(monkeypatch):
    """CSPStage (models/classification/darknetv4.py; reference darknetv4.py:112-115): base layer written into the transition's concat
    buffer + `split_keep_cl` (one activation copy forward, one backward - the split's gradient is the concat's gradient buffer) against
    chunk + cat with three / two copies (HC_CSP_SPLIT=0): output, input gradient and every parameter gradient bit for bit in
    deterministic mode (same kernels, same order; only the copies differ)."""
    import copy

    import holocron_amd as h
    from holocron_amd.models.classification.darknetv4 import CSPStage
    from holocron_amd.ops import conv as cv
    from holocron_amd.ops import nhwc
    h.set_deterministic(True)
    try:
        torch.manual_seed(3)
        ref = CSPStage(32, 64, num_blocks=2, act_layer=torch.nn.Mish(), norm_layer=torch.nn.BatchNorm2d).cuda().train()
        x0 = torch.randn(2, 32, 24, 20, device="cuda")
        r = None
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("HC_CSP_SPLIT", mode)
            m = copy.deepcopy(ref)
            x = x0.clone().requires_grad_(True)
            y = m(x)
            if r is None:
                r = torch.randn(y.shape, device="cuda")
            (y.float() * r).sum().backward()
            cv.flush_deferred_wgrads()
            torch.cuda.synchronize()
            res[mode] = [y.detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
            if mode == "0":
                reused0 = nhwc.SPLIT_STATS["reused"]
        assert nhwc.SPLIT_STATS["reused"] == reused0 + 1          # the one-copy backward really ran in the second arm
        assert len(res["0"]) == len(res["1"])
        for k, (a, b) in enumerate(zip(res["0"], res["1"])):
            assert torch.equal(a, b), k
    finally:
        h.set_deterministic(False)
