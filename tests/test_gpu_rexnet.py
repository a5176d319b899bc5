"""MI355X parity tests of the depthwise / squeeze-excite / channel-padded units, FReLU and ReXNet
(reference: holocron/models/classification/rexnet.py, holocron/nn/modules/activation.py:58-82)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import close_frac, rel_l2

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.to(torch.bfloat16).float()


DW_UNIT_CASES = [(24, 9, 7, 1), (40, 12, 10, 2), (162, 6, 5, 1), (16, 5, 5, 2),
                 # the LDS-tiled kernels (W >= 12, H >= 8, C >= 32 padded): 4-group 8 x 64 tiles, 8 x 32 tiles with a partial channel
                 # slice, 16 x 16 tiles - all with ragged edges
                 (24, 13, 37, 1), (72, 19, 45, 1), (40, 15, 14, 1), (90, 9, 33, 1),
                 # stride 2: the 2 x 2-block data gradient (odd sizes) and the tiled forward / weight gradient (output maps of at
                 # least 12 columns), partial 8-group slices
                 (24, 17, 29, 2), (72, 21, 47, 2), (90, 16, 24, 2)]


@pytest.mark.parametrize("case", DW_UNIT_CASES, ids=["dw%dx%dx%d_s%d" % c for c in DW_UNIT_CASES])
def test_depthwise_conv_kernels_vs_torch(case):
    """hc_dw3x3_{fwd,dgrad,wgrad} through the padded conv unit with an identity BatchNorm (eval mode is not enough:
    training statistics are part of the kernel), stride 1 and 2, channel counts that need padding."""
    import copy

    import holocron_amd as h  # noqa: F401
    from holocron_amd.nn.mbconv_op import ceil16, padded_conv_bn_act
    Cc, H, W, stride = case
    g = torch.Generator().manual_seed(3 + Cc + 7 * H + 31 * W + stride)
    conv = torch.nn.Conv2d(Cc, Cc, 3, stride, 1, groups=Cc, bias=False)
    bn = torch.nn.BatchNorm2d(Cc)
    conv.weight.data = torch.randn(conv.weight.shape, generator=g) * 0.3
    bn.weight.data = torch.rand((Cc,), generator=g) + 0.5
    bn.bias.data = torch.randn((Cc,), generator=g) * 0.2
    x = _bf16(torch.randn((3, Cc, H, W), generator=g))
    xr = x.clone().requires_grad_(True)
    yr = F.relu6(bn(conv(xr)))
    r = _bf16(torch.randn(yr.shape, generator=g))
    gr = torch.autograd.grad((yr * r).sum(), [xr, conv.weight, bn.weight, bn.bias])
    rm_ref = bn.running_mean.clone()
    cg, bg = copy.deepcopy(conv).cuda(), torch.nn.BatchNorm2d(Cc).cuda()
    bg.weight.data, bg.bias.data = bn.weight.data.cuda(), bn.bias.data.cuda()
    xg = x.cuda().requires_grad_(True)
    y = padded_conv_bn_act(xg, cg, bg, torch.nn.ReLU6())
    Cp = ceil16(Cc)          # the package's padding rule: multiples of 16, of 64 for wide layers
    assert y.shape[1] == Cp and (Cp == Cc or float(y[:, Cc:].detach().float().abs().max()) == 0.0)
    ey = rel_l2(y[:, :Cc].float().cpu(), yr.detach())
    (y[:, :Cc].float() * r.cuda()).sum().backward()
    ex, ew = rel_l2(xg.grad.float().cpu(), gr[0]), rel_l2(cg.weight.grad.cpu(), gr[1])
    eg, eb = rel_l2(bg.weight.grad.cpu(), gr[2]), rel_l2(bg.bias.grad.cpu(), gr[3])
    em = rel_l2(bg.running_mean.cpu(), rm_ref)
    # bounds: bf16 storage of the conv output and of the BatchNorm gradient (relative 2^-9 each) through a 9-tap sum and the
    # batch statistics of only 3 H W samples, with cancellation in the tap and bias sums.  Measured over these cases (the round-3
    # strip kernels and the tiled kernels give the same values to 3 digits): y <= 0.0024, dx <= 0.0205, dw <= 0.0289, dgamma <= 0.0074,
    # dbeta <= 0.0229; the values are printed on failure
    assert ey < 6e-3 and ex < 3e-2 and ew < 4e-2 and eg < 3e-2 and eb < 3e-2 and em < 2e-3, (ey, ex, ew, eg, eb, em)


SE_MLP_CASES = [(5, 20, 32, 3, 6, True), (70, 228, 256, 19, 6, True), (256, 1044, 1088, 87, 6, True), (64, 96, 96, 128, 1, False),
                (130, 40, 48, 33, 0, True)]


@pytest.mark.parametrize("case", SE_MLP_CASES, ids=["N%d_C%d_Cp%d_R%d_act%d_b%d" % c for c in SE_MLP_CASES])
def test_se_mlp_kernels_vs_torch(case):
    """hc_se_mlp_fwd / hc_se_mlp_bwd (csrc/se_mlp.hip; reference: the nn.Sequential of SEBlock, rexnet.py:49-61, on the pooled
    vectors with a training-mode BatchNorm2d) against the same MLP in float64 on the CPU: logits, saved statistics, running
    statistics, and every gradient.  Ragged row tiles, channel padding, R up to the supported 128, ReLU6 / ReLU / no activation."""
    import ctypes as C

    from holocron_amd import _lib
    from holocron_amd._lib import check, stream
    N, Cc, Cp, R, act, has_b2 = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(11 + N + Cc + R)
    # the kernels round their MFMA operands to bf16 (like every convolution of the framework): pooled vectors and weights are given
    # bf16-representable values here so that the float64 reference sees the same numbers; what is left is the rounding of the hidden
    # activations and of their gradients
    pooled = torch.zeros((N, Cp))
    pooled[:, :Cc] = _bf16(torch.rand((N, Cc), generator=g))
    w1 = _bf16(torch.randn((R, Cc), generator=g) * (2.0 / Cc ** 0.5))
    gamma, beta = torch.rand((R,), generator=g) + 0.5, torch.randn((R,), generator=g) * 0.5 + (1.0 if act == 6 else 0.0)
    w2 = _bf16(torch.randn((Cc, R), generator=g) * (1.0 / R ** 0.5))
    b2 = torch.randn((Cc,), generator=g) * 0.3 if has_b2 else None
    rm, rv = torch.randn((R,), generator=g) * 0.1, torch.rand((R,), generator=g) + 0.5
    dl = torch.zeros((N, Cp))
    dl[:, :Cc] = torch.randn((N, Cc), generator=g)
    dl = _bf16(dl)
    eps, mom = 1e-5, 0.1
    # float64 reference
    P, W1, G, B, W2 = (t.double().requires_grad_(True) for t in (pooled[:, :Cc], w1, gamma, beta, w2))
    B2 = b2.double().requires_grad_(True) if has_b2 else None
    h1 = P @ W1.t()
    mean, var = h1.mean(0), h1.var(0, unbiased=False)
    y = (h1 - mean) / torch.sqrt(var + eps) * G + B
    h = {0: lambda t: t, 1: torch.relu, 6: lambda t: t.clamp(0, 6)}[act](y)
    logits = h @ W2.t() + (B2 if has_b2 else 0.0)
    grads = torch.autograd.grad((logits * dl[:, :Cc].double()).sum(), [P, W1, G, B, W2] + ([B2] if has_b2 else []))
    rm_ref = (1 - mom) * rm.double() + mom * mean.detach()
    rv_ref = (1 - mom) * rv.double() + mom * h1.detach().var(0, unbiased=N > 1)
    # the kernels
    dev = torch.device("cuda:0")
    nf = int(lib.hc_se_mlp_part_floats(N, R))
    t = {k: v.to(dev).contiguous() for k, v in dict(pooled=pooled, w1=w1, gamma=gamma, beta=beta, w2=w2, rm=rm, rv=rv).items()}
    b2g = b2.to(dev) if has_b2 else None
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    h1g, part, stat = (torch.empty(n, device=dev) for n in (N * R, nf, 2 * R))
    lg = torch.full((N, Cp), 7.0, dtype=torch.bfloat16, device=dev)
    d = _lib.SeMlpDesc()
    d.pooled, d.w1, d.gamma, d.beta, d.w2 = (t[k].data_ptr() for k in ("pooled", "w1", "gamma", "beta", "w2"))
    d.b2 = b2g.data_ptr() if has_b2 else None
    d.running_mean, d.running_var, d.num_batches_tracked = t["rm"].data_ptr(), t["rv"].data_ptr(), nbt.data_ptr()
    d.h1, d.part, d.stat, d.logits = h1g.data_ptr(), part.data_ptr(), stat.data_ptr(), lg.data_ptr()
    d.N, d.C, d.Cp, d.R, d.act, d.eps, d.momentum = N, Cc, Cp, R, act, eps, mom
    check(lib.hc_se_mlp_fwd(C.byref(d), stream()), "hc_se_mlp_fwd")
    assert rel_l2(lg[:, :Cc].float().cpu(), logits.detach().float()) < 8e-3          # bf16 hidden activations, bf16 storage of the logits
    assert Cp == Cc or float(lg[:, Cc:].float().abs().max()) == 0.0
    assert rel_l2(h1g.view(N, R).cpu(), h1.detach().float()) < 1e-5
    assert rel_l2(stat[:R].cpu(), mean.detach().float()) < 1e-5
    assert rel_l2(stat[R:].cpu(), (1.0 / torch.sqrt(var + eps)).detach().float()) < 1e-4
    assert rel_l2(t["rm"].cpu(), rm_ref.float()) < 1e-5 and rel_l2(t["rv"].cpu(), rv_ref.float()) < 1e-4 and int(nbt) == 1
    dlg = dl.to(dev).to(torch.bfloat16)
    gbuf, part2 = torch.empty(N * R, device=dev), torch.empty(nf, device=dev)
    dpool = torch.full((N, Cp), 7.0, device=dev)
    dw1, dw2, dgb = torch.empty((R, Cc), device=dev), torch.empty((Cc, R), device=dev), torch.empty((2, R), device=dev)
    db2 = torch.empty((Cc,), device=dev) if has_b2 else None
    d.dl, d.g, d.part2, d.dpool, d.dw1, d.dw2 = (x.data_ptr() for x in (dlg, gbuf, part2, dpool, dw1, dw2))
    d.dgamma, d.dbeta = dgb.data_ptr(), dgb.data_ptr() + 4 * R
    d.db2 = db2.data_ptr() if has_b2 else None
    check(lib.hc_se_mlp_bwd(C.byref(d), stream()), "hc_se_mlp_bwd")
    got = [dpool[:, :Cc], dw1, dgb[0], dgb[1], dw2] + ([db2] if has_b2 else [])
    for name, a, b in zip(("dpool", "dw1", "dgamma", "dbeta", "dw2", "db2"), got, grads):
        assert rel_l2(a.cpu(), b.float()) < 1e-2, (name, rel_l2(a.cpu(), b.float()))
    assert Cp == Cc or float(dpool[:, Cc:].abs().max()) == 0.0


def _run_block_case(c):
    import holocron_amd as h
    from oracle import rexnet as orx
    cin, cout, t, stride, se = c["cfg"]
    blk = h.models.ReXBlock(cin, cout, t, stride, use_se=se)
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
    oe = orx.rex_block(xe, sd, "b", stride, stride == 1 and cin <= cout, True, emu=True)
    ge = torch.autograd.grad((oe * c["r"]).sum(), [xe] + leaves)
    assert rel_l2(out.float().cpu(), oe.detach()) < 1e-2, (c["cfg"], rel_l2(out.float().cpu(), oe.detach()))
    assert rel_l2(out.float().cpu(), c["out"]) < 3e-2            # and the fp32 reference itself
    (out.float() * c["r"].cuda()).sum().backward()
    assert rel_l2(x.grad.float().cpu(), ge[0]) < 5e-2, (c["cfg"], "dx", rel_l2(x.grad.float().cpu(), ge[0]))
    params = dict(blk.named_parameters())
    for n, gg in zip(names, ge[1:]):
        if float(gg.abs().max()) < 1e-6:
            continue
        e = rel_l2(params[n].grad.float().cpu(), gg)
        assert e < 6e-2, (c["cfg"], n, e)
        assert rel_l2(params[n].grad.float().cpu(), c["dparams"][n]) < 0.15, (c["cfg"], n)
    for k, v in c["state_after"].items():
        if "running" in k:
            assert rel_l2(blk.state_dict()[k].cpu(), v) < 1e-2, k


def test_rexblocks_match_reference_and_bf16_oracle(golden):
    for c in golden("rexnet.pt")["blocks"]:
        _run_block_case(c)


def test_frelu_matches_reference(golden):
    import holocron_amd as h
    f = golden("rexnet.pt")["frelu"]
    m = h.nn.FReLU(24)
    m.load_state_dict(f["state"])
    m = m.cuda().train()
    x = f["x"].cuda().requires_grad_(True)
    out = m(x)
    assert out.shape == f["out"].shape
    assert rel_l2(out.float().cpu(), f["out"]) < 5e-3
    (out.float() * f["r"].cuda()).sum().backward()
    # max(x, t) has a kink at x == t: where bf16 rounding of t flips the selection the gradient moves by O(1)
    assert close_frac(x.grad.float().cpu(), f["dx"], 2e-2, 2e-2) > 0.97, rel_l2(x.grad.float().cpu(), f["dx"])
    params = dict(m.named_parameters())
    for n, gg in f["dparams"].items():
        if float(gg.abs().max()) < 1e-4:     # the conv bias in front of a training-mode BatchNorm has no gradient
            assert float(params[n].grad.abs().max()) < 1e-4
            continue
        assert rel_l2(params[n].grad.float().cpu(), gg) < 8e-2, (n, rel_l2(params[n].grad.float().cpu(), gg))
    assert rel_l2(m.bn.running_mean.cpu(), f["state_after"]["bn.running_mean"]) < 2e-3
    # tests/test_nn_activation.py: output shape equals input shape, for channel counts that need padding too
    assert h.nn.FReLU(3).cuda()(torch.rand(2, 3, 8, 8).cuda()).shape == (2, 3, 8, 8)


def test_rexnet1_0x_train_step_matches_reference(golden):
    import holocron_amd as h
    gm = golden("rexnet.pt")["model"]
    torch.manual_seed(gm["seed"])
    m = h.models.rexnet1_0x(num_classes=gm["num_classes"], dropout_ratio=0.0).cuda().train()
    logits = m(gm["x"].cuda())
    assert logits.shape == gm["logits"].shape
    # SMOKE only (4 images, 3 x 3 maps at the end): shapes, finiteness, the loss in the right place and the first layers' running
    # statistics.  The whole-model PARITY check - every block in situ against the oracle, and the free-running logits against the
    # emulating oracle with a yardstick - is tests/test_gpu_whole_models.py on the 16 x 128 x 128 fixture
    assert bool(torch.isfinite(logits).all())
    loss = F.cross_entropy(logits.float(), gm["target"].cuda())
    assert abs(float(loss) - float(gm["loss"])) < 0.1 * float(gm["loss"])
    loss.backward()
    params = dict(m.named_parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params.values())
    for n in ("features.1.running_mean", "features.3.conv.1.running_var"):
        assert rel_l2(m.state_dict()[n].cpu(), gm["running"][n]) < 1e-2, n
    m.eval()
    with torch.no_grad():
        assert m(gm["x"].cuda()).shape == (4, 10)


def test_rexnet_capture_without_two_eager_steps_is_refused():
    """a ReXNet's SECOND forward uploads the item tables of its padded convs (host-to-device copy).  Under stream capture
    that copy would be recorded with a host buffer that is gone at replay, so it is refused; after two eager steps capture works."""
    import holocron_amd as h
    from holocron_amd import parallel
    torch.manual_seed(0)
    m = h.models.rexnet1_0x(num_classes=10, dropout_ratio=0.0).cuda().train()
    opt = h.optim.AdaBelief(m.parameters(), lr=1e-3)
    x, t = torch.rand(4, 3, 64, 64).cuda(), torch.randint(0, 10, (4,)).cuda()

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        F.cross_entropy(m(x).float(), t).backward()
    fwd_bwd()                                   # forward 1 registers the units
    opt.step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with pytest.raises(RuntimeError, match="outside stream capture"):
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                m(x)                            # forward 2 would upload the tables inside the capture
    torch.cuda.synchronize()
    fwd_bwd()                                   # eager forward 2: tables uploaded
    opt.step()
    gs = parallel.GraphedStep(fwd_bwd, opt, None)
    gs.capture()
    before = m.head[1].weight.detach().clone()
    gs.run()
    torch.cuda.synchronize()
    assert not torch.equal(before, m.head[1].weight.detach())
    gs.release()
