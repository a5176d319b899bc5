"""MI355X parity tests of the CSP / PAN / SPP data movement, conv+bias, DropBlock fusion and the YOLOv4 layer and
model (reference: holocron/models/detection/yolov4.py, holocron/models/classification/darknetv4.py)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.to(torch.bfloat16).float()


def test_nhwc_cat_chunk_upsample_spp_vs_torch():
    from holocron_amd.ops import nhwc
    from holocron_amd.ops.conv import to_cl_bf16
    g = torch.Generator().manual_seed(2)
    # chunk / cat
    x = _bf16(torch.randn((2, 32, 5, 7), generator=g))
    xg = x.cuda().requires_grad_(True)
    a, b = nhwc.chunk2_cl(xg)
    assert torch.equal(a.float().cpu(), x[:, :16]) and torch.equal(b.float().cpu(), x[:, 16:])
    buf, (p0, p1) = nhwc.cat_buffer(2, [16, 16], 5, 7, "cuda")
    up_src = _bf16(torch.randn((2, 16, 5, 7), generator=g)).cuda().requires_grad_(True)
    y = nhwc.cat_cl([b, up_src], buf)                       # both parts get copied in
    assert torch.equal(y.float().cpu(), torch.cat([x[:, 16:], up_src.detach().float().cpu()], 1))
    r = _bf16(torch.randn(y.shape, generator=g))
    (y.float() * r.cuda()).sum().backward()
    assert torch.equal(xg.grad.float().cpu(), torch.cat([torch.zeros_like(r[:, :16]), r[:, :16]], 1))
    assert torch.equal(up_src.grad.float().cpu(), r[:, 16:])
    # upsample straight into a concat slice
    s = _bf16(torch.randn((2, 8, 3, 4), generator=g))
    sg = s.cuda().requires_grad_(True)
    buf, (p0, p1) = nhwc.cat_buffer(2, [8, 8], 6, 8, "cuda")
    o = nhwc.upsample2x_cl(sg, out=p1)
    other = to_cl_bf16(_bf16(torch.randn((2, 8, 6, 8), generator=g)).cuda())
    yc = nhwc.cat_cl([other, o], buf)
    ref_up = F.interpolate(s, scale_factor=2, mode="nearest")
    assert torch.equal(yc.float().cpu(), torch.cat([other.float().cpu(), ref_up], 1))
    r = _bf16(torch.randn(yc.shape, generator=g))
    (yc.float() * r.cuda()).sum().backward()
    sref = s.clone().requires_grad_(True)
    (F.interpolate(sref, scale_factor=2, mode="nearest") * r[:, 8:]).sum().backward()
    assert torch.allclose(sg.grad.float().cpu(), sref.grad, rtol=8e-3, atol=1e-6)
    # SPP (5, 9, 13)
    for (N, Cc, H, W) in [(2, 16, 9, 11), (1, 8, 19, 19), (2, 24, 4, 3)]:
        x = _bf16(torch.randn((N, Cc, H, W), generator=g))
        x[0, :, : H // 2] = x[0, :, : H // 2].round()        # plenty of exact ties
        xg = x.cuda().requires_grad_(True)
        y = nhwc.spp_cl(xg)
        xr = x.clone().requires_grad_(True)
        yr = torch.cat([xr] + [F.max_pool2d(xr, k, 1, k // 2) for k in (5, 9, 13)], 1)
        assert torch.equal(y.float().cpu(), yr.detach())
        r = _bf16(torch.randn(yr.shape, generator=g))
        (y.float() * r.cuda()).sum().backward()
        (yr * r).sum().backward()
        assert torch.allclose(xg.grad.float().cpu(), xr.grad, rtol=1.6e-2, atol=1e-2), (N, Cc, H, W)
        assert rel_l2(xg.grad.float().cpu(), xr.grad) < 4e-3


def test_spp_lds_tiled_kernels_are_bit_identical_to_the_window_walk(monkeypatch):
    """hc_spp_fwd (csrc/nhwc_ops.hip): the LDS-tiled integer-key kernel (one image x two channel groups per workgroup, smallest-key halo)
    against the global-memory window walk (HC_SPP_TILE=0): values (== : the keyed kernel stores +0 where the walk may store -0) and argmax
    bytes (ties!) bit for bit; hc_spp_bwd's LDS scatter against the walk's gather: the same sums up to fp32 addition order (and the walk
    itself in deterministic mode); incl. a partial channel-group slice (40 channels = 5 groups) and non-square maps."""
    from holocron_amd import _lib
    from holocron_amd._lib import check, ptr, stream
    from holocron_amd.ops.conv import empty_cl, to_cl_bf16
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    for (N, Cc, H, W) in [(3, 40, 19, 19), (2, 64, 13, 16), (1, 8, 5, 23), (16, 512, 19, 19)]:
        x = _bf16(torch.randn((N, Cc, H, W), generator=g))
        x[0, :, : H // 2] = x[0, :, : H // 2].round()
        xg = to_cl_bf16(x.cuda())
        gr = to_cl_bf16(_bf16(torch.randn((N, 4 * Cc, H, W), generator=g)).cuda())
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("HC_SPP_TILE", mode)
            out = empty_cl(N, 4 * Cc, H, W, xg.device)
            idx = torch.zeros((3, N, H, W, Cc), dtype=torch.uint8, device=xg.device)
            dx = empty_cl(N, Cc, H, W, xg.device)
            check(lib.hc_spp_fwd(ptr(xg), ptr(out), ptr(idx), N, H, W, Cc, stream()), "hc_spp_fwd")
            check(lib.hc_spp_bwd(ptr(gr), ptr(idx), ptr(dx), N, H, W, Cc, stream()), "hc_spp_bwd")
            torch.cuda.synchronize()
            res[mode] = (out.clone(), idx.clone(), dx.clone())
        for a, b, what in zip(res["0"][:2], res["1"][:2], ("out", "idx")):
            assert torch.equal(a, b), (what, N, Cc, H, W)
        # the backward of mode 1 is the LDS scatter: same terms, fp32 additions in arrival order, rounded to bf16 once
        a, b = res["0"][2].float(), res["1"][2].float()
        assert torch.allclose(a, b, rtol=8e-3, atol=1e-3 * float(a.abs().max())), ("dx", N, Cc, H, W, float((a - b).abs().max()))
        assert float((a != b).float().mean()) < 0.05, ("dx: more than rounding-order differences", N, Cc, H, W)
        import holocron_amd as h
        h.set_deterministic(True)                      # deterministic mode keeps the fixed-order walk: bit-equal to mode 0
        try:
            dxd = empty_cl(N, Cc, H, W, xg.device)
            check(lib.hc_spp_bwd(ptr(gr), ptr(res["1"][1]), ptr(dxd), N, H, W, Cc, stream()), "hc_spp_bwd")
            torch.cuda.synchronize()
            assert torch.equal(dxd, res["0"][2]), ("deterministic dx", N, Cc, H, W)
        finally:
            h.set_deterministic(False)


def test_conv_bias_matches_torch():
    import holocron_amd as h
    from holocron_amd.nn.convbn_op import conv_bias
    g = torch.Generator().manual_seed(4)
    for (cin, cout, k) in [(32, 27, 1), (64, 255, 1), (16, 24, 3)]:
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)
        conv.weight.data = _bf16(torch.randn(conv.weight.shape, generator=g) * 0.1)
        conv.bias.data = torch.randn((cout,), generator=g)
        x = _bf16(torch.randn((2, cin, 6, 5), generator=g))
        xr = x.clone().requires_grad_(True)
        yr = conv(xr)
        r = _bf16(torch.randn(yr.shape, generator=g))
        gr = torch.autograd.grad((yr * r).sum(), [xr, conv.weight, conv.bias])
        cg = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=True).cuda()
        cg.load_state_dict(conv.state_dict())
        xg = x.cuda().requires_grad_(True)
        y = conv_bias(xg, cg)
        assert y.shape[1] == (cout + 15) // 16 * 16
        assert y.shape[1] == cout or float(y[:, cout:].detach().float().abs().max()) == 0.0
        assert rel_l2(y[:, :cout].float().cpu(), yr.detach()) < 4e-3
        (y[:, :cout].float() * r.cuda()).sum().backward()
        assert rel_l2(xg.grad.float().cpu(), gr[0]) < 6e-3
        assert rel_l2(cg.weight.grad.cpu(), gr[1]) < 2e-3
        assert rel_l2(cg.bias.grad.cpu(), gr[2]) < 2e-3
    assert h.models.detection.YoloLayer is not None


def test_conv_bn_act_dropblock_into_concat_slice(monkeypatch):
    """conv -> BN -> Mish -> DropBlock written into a concat slice, against torch ops with the same noise."""
    import holocron_amd as h
    from holocron_amd.nn import functional as Fh
    from holocron_amd.nn.convbn_op import run_conv_sequence
    from holocron_amd.ops import nhwc
    from oracle import functional as of
    g = torch.Generator().manual_seed(6)
    cin, cout, N, H, W = 32, 48, 3, 9, 8
    seq_cpu = torch.nn.Sequential(*h.models.utils.conv_sequence(cin, cout, torch.nn.Mish(), torch.nn.BatchNorm2d,
                                                                h.nn.DropBlock2d, kernel_size=3, padding=1))
    seq_cpu[0].weight.data = _bf16(torch.randn(seq_cpu[0].weight.shape, generator=g) * 0.1)
    seq_cpu[1].weight.data = torch.rand((cout,), generator=g) + 0.5
    seq_cpu[1].bias.data = torch.randn((cout,), generator=g) * 0.2
    seq_cpu[3].p, seq_cpu[3].block_size = 0.1 * 9 * 3, 3
    noise = torch.rand((N, H, W), generator=g)
    x = _bf16(torch.randn((N, cin, H, W), generator=g))
    xr = x.clone().requires_grad_(True)
    z = F.mish(F.batch_norm(F.conv2d(xr, seq_cpu[0].weight, None, 1, 1), None, None, seq_cpu[1].weight, seq_cpu[1].bias, True))
    yr = of.dropblock2d(z, seq_cpu[3].drop_prob, 3, noise)
    r = _bf16(torch.randn(yr.shape, generator=g))
    gr = torch.autograd.grad((yr * r).sum(), [xr, seq_cpu[0].weight, seq_cpu[1].weight, seq_cpu[1].bias])
    assert float((yr == 0).float().mean()) > 0.05          # blocks really dropped

    import copy
    seq = copy.deepcopy(seq_cpu).cuda().train()
    monkeypatch.setattr(Fh, "_noise", lambda shape, device: noise.to(device))
    buf, (p0, p1) = nhwc.cat_buffer(N, [16, cout], H, W, "cuda")
    xg = x.cuda().requires_grad_(True)
    y = run_conv_sequence(seq, xg, out=p1)
    assert y.data_ptr() == p1.data_ptr()
    full = nhwc.cat_cl([torch.zeros((N, 16, H, W), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last), y], buf)
    assert rel_l2(full[:, 16:].float().cpu(), yr.detach()) < 5e-3
    rr = torch.cat([torch.zeros((N, 16, H, W)), r], 1)
    (full.float() * rr.cuda()).sum().backward()
    assert rel_l2(xg.grad.float().cpu(), gr[0]) < 2.5e-2
    assert rel_l2(seq[0].weight.grad.cpu(), gr[1]) < 1.5e-2
    assert rel_l2(seq[1].weight.grad.cpu(), gr[2]) < 3e-2 and rel_l2(seq[1].bias.grad.cpu(), gr[3]) < 2e-2


def _targets_to(tg, dev):
    return [{k: v.to(dev) for k, v in t.items()} for t in tg]


def test_yolo_layer_fp32_logits_match_reference(golden):
    import holocron_amd as h
    for c in golden("yolo.pt")["layers"]:
        layer = h.models.detection.YoloLayer(c["anchors"].clone(), num_classes=c["nc"], scale_xy=c["scale_xy"]).cuda()
        x = c["x"].cuda().requires_grad_(True)
        layer.train()
        boxes, b_o, b_s = layer._format_outputs(x)
        assert torch.allclose(boxes.cpu(), c["boxes"], rtol=1e-5, atol=1e-6)
        assert b_o.shape == c["boxes"].shape[:-1] and b_s.shape[-1] == c["nc"]
        losses = layer(x, _targets_to(c["target"], "cuda"))
        for k, v in c["losses"].items():
            assert losses[k].shape == v.shape, k
            assert torch.allclose(losses[k].cpu(), v, rtol=2e-5, atol=1e-6), (k, losses[k], v)
        sum(c["weights"][k] * v.sum() for k, v in losses.items()).backward()
        assert torch.allclose(x.grad.cpu(), c["dx"], rtol=2e-4, atol=2e-7)
        layer.eval()
        dets = layer(c["x"].cuda())
        for d, r in zip(dets, c["detections"]):
            assert torch.equal(d["labels"].cpu(), r["labels"])
            assert torch.allclose(d["boxes"].cpu(), r["boxes"], rtol=1e-5, atol=1e-6)
            assert torch.allclose(d["scores"].cpu(), r["scores"], rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        layer.train()(c["x"].cuda())                              # yolov4.py:433-434


def test_yolo_layer_full_size_against_oracle():
    """608-input scale (76 x 76 x 3 anchors x 85) in the padded NHWC bf16 layout the head produces."""
    import holocron_amd as h
    from holocron_amd.ops.conv import to_cl_bf16
    from oracle import yolo as oy
    g = torch.Generator().manual_seed(8)
    N, H, W, nc = 2, 76, 76, 80
    anchors = torch.tensor([[12, 16], [19, 36], [40, 28]], dtype=torch.float32) / 608
    x = _bf16(torch.randn((N, 255, H, W), generator=g))
    tg = []
    for k in (5, 8):
        b = torch.rand((k, 4), generator=g)
        b[:, :2] *= b[:, 2:]
        b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.02).clamp(max=0.999)
        tg.append({"boxes": b, "labels": torch.randint(0, nc, (k,), generator=g)})
    xr = x.clone().requires_grad_(True)
    ref = oy.compute_losses(xr, tg, anchors, nc, 1.2)
    (dref,) = torch.autograd.grad(sum(v.sum() for v in ref.values()), xr)
    layer = h.models.detection.YoloLayer(anchors.clone(), num_classes=nc, scale_xy=1.2).cuda().train()
    xp = to_cl_bf16(torch.cat([x, torch.zeros((N, 1, H, W))], 1).cuda()).requires_grad_(True)     # 256 channels per pixel
    losses = layer(xp, _targets_to(tg, "cuda"))
    for k, v in ref.items():
        assert torch.allclose(losses[k].cpu().reshape(v.shape), v.detach(), rtol=1e-4, atol=1e-6), k
    sum(v.sum() for v in losses.values()).backward()
    got = xp.grad.float().cpu()
    assert float(got[:, 255:].abs().max()) == 0.0
    assert torch.allclose(got[:, :255], dref, rtol=1.6e-2, atol=1e-7)
    layer.eval()
    dets = layer(xp.detach())
    dref = oy.post_process(x, anchors, nc, 1.2)
    for d, r in zip(dets, dref):
        # expf on the GPU and libm's exp differ in the last bit, which can swap two almost-equal scores in the NMS
        # order (the NMS kernel itself is bit-exact on identical inputs: tests/test_gpu_pointwise.py): compare as sets
        assert abs(d["boxes"].shape[0] - r["boxes"].shape[0]) <= 0.002 * r["boxes"].shape[0]
        key = lambda b, l: set(zip((b * 1e5).round().long().view(-1, 4).sum(1).tolist(), l.tolist()))   # noqa: E731
        a, b = key(d["boxes"].cpu(), d["labels"].cpu()), key(r["boxes"], r["labels"])
        assert len(a & b) >= 0.995 * len(b)


class _Replay:
    """Replays DropBlock's uniform draws from a seed on the CPU generator (oracle side: ``draw(shape)``; HIP side: the
    ``functional._noise(shape, device)`` hook)."""

    def __init__(self, seed):
        self.g, self.n = torch.Generator().manual_seed(seed), 0

    def draw(self, shape):
        self.n += 1
        return torch.rand(tuple(shape), generator=self.g)

    def hook(self, shape, device):
        return self.draw(shape).to(device)


def _block_vs_oracle(monkeypatch, module, run_hip, run_oracle, inputs, drop_p, seed, tol_out, tol_grad):
    """Run a sub-block on the HIP path and through the bf16-emulating oracle on identical inputs / weights / noise.
    Sub-blocks are a handful of conv units deep, so (unlike the 70-layer random net, which amplifies a 1e-3 input
    perturbation to 5-15 % at the logits) storage rounding stays at the 1e-2 level and wiring errors show."""
    from holocron_amd.nn import functional as Fh
    import holocron_amd as h
    from oracle import yolov4 as ov
    g = torch.Generator().manual_seed(seed)
    for mod in module.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = torch.rand(mod.weight.shape, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.2
        if isinstance(mod, h.nn.DropBlock2d):
            mod.p = drop_p
    sd = {"m." + k: v.clone() for k, v in module.state_dict().items()}
    names = [n for n, _ in module.named_parameters()]
    leaves = [sd["m." + n].requires_grad_(True) for n in names]
    xs = [t.clone().requires_grad_(True) for t in inputs]
    cfg = ov.Cfg(act="mish", drop=(drop_p, 7), noise=_Replay(seed + 1), training=True, emulate_bf16=True)
    outs = run_oracle(sd, xs, cfg)
    rs = [_bf16(torch.randn(o.shape, generator=g)) for o in outs]
    gref = torch.autograd.grad(sum((o * r).sum() for o, r in zip(outs, rs)), xs + leaves)
    module = module.cuda().train()
    rp = _Replay(seed + 1)
    monkeypatch.setattr(Fh, "_noise", rp.hook)
    xg = [t.cuda().requires_grad_(True) for t in inputs]
    got = run_hip(module, xg)
    assert rp.n == cfg.noise.n and rp.n > 0
    for o, oref in zip(got, outs):
        assert o.shape == oref.shape
        assert rel_l2(o.float().cpu(), oref.detach()) < tol_out, rel_l2(o.float().cpu(), oref.detach())
    sum((o.float() * r.cuda()).sum() for o, r in zip(got, rs)).backward()
    for t, gr in zip(xg, gref[:len(xg)]):
        assert rel_l2(t.grad.float().cpu(), gr) < tol_grad, ("input", rel_l2(t.grad.float().cpu(), gr))
    params = dict(module.named_parameters())
    for n, gr in zip(names, gref[len(xg):]):
        e = rel_l2(params[n].grad.float().cpu(), gr)
        assert e < tol_grad, (n, e)
    for k, v in module.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_l2(v.cpu(), sd["m." + k].detach()) < 5e-3, k


def test_csp_stage_matches_bf16_oracle(monkeypatch):
    import holocron_amd as h
    from holocron_amd.models.classification.darknetv4 import CSPStage
    from oracle import yolov4 as ov
    for (cin, cout, nb, hw, seed) in [(32, 64, 2, 20, 3), (16, 32, 1, 16, 5)]:
        torch.manual_seed(seed)
        st = CSPStage(cin, cout, nb, torch.nn.Mish(inplace=True), torch.nn.BatchNorm2d, h.nn.DropBlock2d)
        h.nn.init.init_module(st, "leaky_relu")
        x = _bf16(torch.randn((2, cin, hw, hw), generator=torch.Generator().manual_seed(seed)))
        _block_vs_oracle(monkeypatch, st, lambda m, xs: [m(xs[0])], lambda sd, xs, cfg: [ov.csp_stage(xs[0], sd, "m", nb, cfg)],
                         [x], 0.1 * 49 * 2, seed, 2e-2, 6e-2)


def test_pan_and_neck_match_bf16_oracle(monkeypatch):
    import holocron_amd as h
    from holocron_amd.models.detection.yolov4 import PAN, Neck
    from oracle import yolov4 as ov
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    pan = PAN(64, torch.nn.Mish(inplace=True), torch.nn.BatchNorm2d, h.nn.DropBlock2d)
    h.nn.init.init_module(pan, "leaky_relu")
    x, up = _bf16(torch.randn((2, 64, 7, 7), generator=g)), _bf16(torch.randn((2, 64, 14, 14), generator=g))
    _block_vs_oracle(monkeypatch, pan, lambda m, xs: [m(xs[0], xs[1])], lambda sd, xs, cfg: [ov.pan(xs[0], xs[1], sd, "m", cfg)],
                     [x, up], 0.1 * 49 * 2, 13, 2e-2, 6e-2)
    torch.manual_seed(14)
    neck = Neck([128, 64, 32], torch.nn.Mish(inplace=True), torch.nn.BatchNorm2d, h.nn.DropBlock2d)
    feats = [_bf16(torch.randn((2, 32, 24, 24), generator=g)), _bf16(torch.randn((2, 64, 12, 12), generator=g)),
             _bf16(torch.randn((2, 128, 6, 6), generator=g))]
    _block_vs_oracle(monkeypatch, neck, lambda m, xs: list(m(xs)), lambda sd, xs, cfg: list(ov.neck(xs, sd, cfg, prefix="m")),
                     feats, 0.1 * 49, 15, 5e-2, 0.2)


def test_yolov4_head_logits_match_bf16_oracle(monkeypatch):
    """The three head branches up to the (padded, bias-carrying) output convs, without the loss."""
    import holocron_amd as h
    from holocron_amd.models.detection.yolov4 import Yolov4Head
    from holocron_amd.nn.convbn_op import run_conv_sequence
    from holocron_amd.ops.nhwc import cat_buffer, cat_cl
    from oracle import yolov4 as ov
    torch.manual_seed(16)
    head = Yolov4Head(4, None, torch.nn.Mish(inplace=True), torch.nn.BatchNorm2d, h.nn.DropBlock2d)
    gh = torch.Generator().manual_seed(17)
    for seq in (head.head1, head.head2_2, head.head3):
        seq[-1].weight.data = _bf16(torch.randn(seq[-1].weight.shape, generator=gh) * 0.05)
        seq[-1].bias.data = torch.randn(seq[-1].bias.shape, generator=gh) * 0.5
    feats = [_bf16(torch.randn((2, 128, 16, 16), generator=gh)), _bf16(torch.randn((2, 256, 8, 8), generator=gh)),
             _bf16(torch.randn((2, 512, 4, 4), generator=gh))]

    def run_hip(m, xs):
        o1 = run_conv_sequence(m.head1, xs[0])
        buf, (pa, _) = cat_buffer(2, [256, 256], 8, 8, "cuda")
        h2 = run_conv_sequence(m.head2_1, cat_cl([run_conv_sequence(m.pre_head2, xs[0], out=pa), xs[1]], buf))
        o2 = run_conv_sequence(m.head2_2, h2)
        buf, (pa, _) = cat_buffer(2, [512, 512], 4, 4, "cuda")
        o3 = run_conv_sequence(m.head3, cat_cl([run_conv_sequence(m.pre_head3, h2, out=pa), xs[2]], buf))
        return [o1, o2, o3]
    _block_vs_oracle(monkeypatch, head, run_hip, lambda sd, xs, cfg: list(ov.head_logits(xs, sd, cfg, prefix="m")), feats,
                     0.1 * 49, 18, 4e-2, 0.2)


def _golden_yolov4(gm):
    import holocron_amd as h
    torch.manual_seed(gm["seed"])
    m = h.models.detection.YOLOv4(gm["layout"], num_classes=gm["num_classes"], stem_channels=gm["stem_channels"])
    gh = torch.Generator().manual_seed(gm["head_seed"])
    for seq in (m.head.head1, m.head.head2_2, m.head.head3):
        seq[-1].weight.data = torch.randn(seq[-1].weight.shape, generator=gh) * 0.05
        seq[-1].bias.data = torch.randn(seq[-1].bias.shape, generator=gh) * 0.5
    for mod in m.modules():
        if isinstance(mod, h.nn.DropBlock2d):
            mod.p = gm["drop_p"]
    return m


def test_yolov4_reduced_model_train_step_matches_reference(golden, monkeypatch):
    """End to end against the reference's own numbers.  A randomly initialised 70-layer net with batch statistics over
    2 images amplifies a 1e-3 relative input perturbation to 5-15 % at the logits (measured on the fp32 oracle), so
    bf16 storage bounds this comparison to the 10 % level; the sharp checks are the per-block tests above."""
    from holocron_amd.nn import functional as Fh
    gm = golden("yolo.pt")["model"]
    m = _golden_yolov4(gm).cuda().train()
    rp = _Replay(gm["noise_seed"])
    monkeypatch.setattr(Fh, "_noise", rp.hook)
    losses = m(gm["x"].cuda(), _targets_to(gm["target"], "cuda"))
    assert rp.n == gm["n_draws"]                                # same number (and, by shape, order) of DropBlock draws
    for k, v in gm["losses"].items():
        assert losses[k].shape == v.shape
        assert abs(float(losses[k].sum()) - float(v.sum())) < 0.1 * max(abs(float(v.sum())), 0.1), (k, losses[k], v)
    sum(v.sum() for v in losses.values()).backward()
    params = dict(m.named_parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params.values())
    for n in ("head.head1.3.bias", "head.head1.3.weight"):      # one conv away from the loss: well conditioned
        got, gref = params[n].grad.float().cpu(), gm["grads"][n]
        assert float(F.cosine_similarity(got.flatten(), gref.flatten(), dim=0)) > 0.99, n
    for n in ("backbone.stem.1.running_mean", "backbone.stages.0.base_layer.1.running_var", "backbone.stages.1.transition.1.running_var"):
        assert rel_l2(m.state_dict()[n].cpu(), gm["running"][n]) < 1e-2, n
    m.eval()
    with torch.no_grad():
        dets = m(gm["x"].cuda())
    assert len(dets) == 2 and all(set(d) == {"boxes", "scores", "labels"} for d in dets)
    for d, nref in zip(dets, gm["n_detections"]):
        assert abs(d["boxes"].shape[0] - nref) <= max(5, 0.05 * nref)
        assert d["labels"].dtype == torch.int64 and d["boxes"].shape[1] == 4
    with pytest.raises(ValueError):
        m.train()(gm["x"].cuda())


def test_cspdarknet53_mish_forward_backward_smoke():
    import holocron_amd as h
    torch.manual_seed(0)
    m = h.models.cspdarknet53_mish(num_classes=10).cuda().train()
    for mod in m.modules():
        if isinstance(mod, h.nn.DropBlock2d):
            mod.p = 0.1 * 49 * 2
    x = torch.rand((2, 3, 64, 64), device="cuda")
    out = m(x)                                   # records the DropBlock call sequence
    assert out.shape == (2, 10) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    plan = m._hc_drop_plan
    ndrop = sum(isinstance(mod, h.nn.DropBlock2d) for mod in m.modules())
    assert plan.ready and len(plan.entries) == ndrop
    outs = []
    for _ in range(2):                           # one rand + one batched mask launch per forward from now on
        m.zero_grad()
        out = m(x)
        assert plan.cursor == ndrop and torch.isfinite(out).all()
        out.sum().backward()
        outs.append(out.detach())
    assert not torch.equal(outs[0], outs[1])     # fresh noise every step
    m(torch.rand((2, 3, 96, 96), device="cuda")).sum().backward()     # another resolution: the plan is re-recorded
    assert len(m._hc_drop_plan.entries) == ndrop


def test_yolov4_step_replays_from_a_graph_with_packed_targets(golden, monkeypatch):
    """the YOLOv4 training step faulted under hipGraph replay (its per-forward target packing became pageable
    memcpy nodes).  With the ground truth packed once (PackedTargets) the whole step - forward, four losses, backward - is captured
    and every replay reproduces the eager step; packing inside a capture is refused instead of recorded."""
    import holocron_amd as h
    from holocron_amd.models.detection.yolov4 import PackedTargets, YoloLayer
    gm = golden("yolo.pt")["model"]
    m = _golden_yolov4(gm).cuda().train()
    for mod in m.modules():                       # no DropBlock noise: eager and replayed steps must see the same function
        if hasattr(mod, "p") and mod.__class__.__name__ == "DropBlock2d":
            mod.p = 0.0
    x = gm["x"].cuda()
    tgt = _targets_to(gm["target"], "cuda")
    packed = PackedTargets(tgt, x.device)
    assert len(packed) == len(tgt)
    out = {}

    def step(t):
        for p in m.parameters():
            p.grad = None
        losses = m(x, t)
        sum(v.sum() for v in losses.values()).backward()
        out["loss"] = torch.stack([v.sum().detach() for v in losses.values()])

    step(tgt)
    step(packed)                                  # same numbers through the packed form
    torch.cuda.synchronize()
    ref_loss = out["loss"].clone()
    ref_grad = {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
    # two eager runs of this randomly initialised 70-layer net already differ by a few percent in a loss (atomics order -> bf16
    # rounding flips -> chaotic amplification, see the module docstring): the bound separates that from a stale / faulting replay
    def same(a, b):          # the objectness term alone moves by +-15 % between two eager runs; the sum by ~2 %
        return bool(torch.isfinite(a).all()) and torch.allclose(a, b, rtol=0.3, atol=1e-2) and abs(float(a.sum() - b.sum())) < 0.1 * float(b.sum())
    step(tgt)
    assert same(out["loss"], ref_loss)
    g = torch.cuda.CUDAGraph()
    with monkeypatch.context() as mp:             # what a capture would see, without poisoning a real one
        mp.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
        with pytest.raises(RuntimeError, match="before stream capture"):
            YoloLayer._pack_targets(tgt, x.device)
        with pytest.raises(RuntimeError, match="outside stream capture"):
            PackedTargets(tgt, x.device)
    step(packed)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        step(packed)
    torch.cuda.synchronize()
    for it in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert same(out["loss"], ref_loss), (it, out["loss"], ref_loss)
        for n, p in m.named_parameters():
            if n in ref_grad and n.startswith("head.head1.3."):          # one conv away from the loss: well conditioned
                e = float((p.grad.float() - ref_grad[n]).norm() / (ref_grad[n].norm() + 1e-12))
                assert e < 0.5, (it, n, e)
    packed.update(tgt)                            # same counts: refill in place
    with pytest.raises(ValueError):
        packed.update(tgt[:1] + tgt[:1] if len(tgt) > 1 and tgt[0]["boxes"].shape[0] != tgt[1]["boxes"].shape[0] else [{"boxes": tgt[0]["boxes"][:0], "labels": tgt[0]["labels"][:0]}] * len(tgt))


def test_batched_post_processing_equals_per_layer_post_processing():
    """The eval path of the head (post_process_scales: every (image, scale) NMS problem of the batch in one launch pair) returns what
    the three YoloLayers' own post_process_logits + the reference's per-image concatenation (yolov4.py:302-336, 603-609) return:
    the same detections in the same order, bit for bit - also with empty problems and images without any candidate."""
    from holocron_amd.models.detection.yolov4 import YoloLayer, post_process_scales
    torch.manual_seed(5)
    anchors = torch.tensor([[[12, 16], [19, 36], [40, 28]], [[36, 75], [76, 55], [72, 146]], [[142, 110], [192, 243], [459, 401]]],
                           dtype=torch.float32) / 608
    nc, N = 7, 5
    layers = [YoloLayer(anchors[i], num_classes=nc, scale_xy=s).cuda().eval() for i, s in enumerate((1.2, 1.1, 1.05))]
    outs = [torch.randn((N, 3 * (5 + nc), h, h), device="cuda") for h in (24, 12, 6)]
    for o in outs:
        o[3] -= 30.0                       # image 3: no candidate passes the objectness threshold at any scale
    outs[1][0] -= 30.0                     # image 0: an empty problem between two populated ones
    outs[2][1, :, :, :] = outs[2][1, :, :1, :1]     # image 1, coarsest scale: identical cells -> identical scores (ties) and heavy overlap
    ref = [l.post_process_logits(o) for l, o in zip(layers, outs)]
    ref = [{k: torch.cat([r[i][k] for r in ref], 0) for k in ("boxes", "scores", "labels")} for i in range(N)]
    got = post_process_scales(layers, outs)
    assert len(got) == N
    assert sum(int(d["boxes"].shape[0]) for d in ref) > 200 and ref[3]["boxes"].shape[0] == 0
    for a, b in zip(got, ref):
        for k in ("boxes", "scores", "labels"):
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
