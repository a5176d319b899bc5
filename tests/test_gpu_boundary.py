"""GPU: the round-2 boundary additions - differentiable box operators, the functional normalised convolution, backward through
fused units that run on their RUNNING statistics (eval mode / freeze_bn) and the trainers driving the HIP models."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _rand_boxes(n, g, scale=100.0):
    b = torch.rand((n, 4), generator=g) * scale
    b[:, 2:] = b[:, :2] + 1.0 + torch.rand((n, 2), generator=g) * scale * 0.5
    return b


@pytest.mark.parametrize("name", ["box_giou", "diou_loss", "ciou_loss", "iou_penalty", "aspect_ratio_consistency"])
def test_box_ops_gradients_match_reference_expressions(name):
    """the reference's functions are plain torch expressions (ops/boxes.py:33-211): their autograd gradient on CPU (oracle/boxes.py
    restates them) against hc_box_pairwise_bwd.  Includes tied coordinates (max / min split the gradient) and touching boxes."""
    import holocron_amd as h
    from oracle import boxes as ob
    g = torch.Generator().manual_seed(0)
    b1, b2 = _rand_boxes(37, g), _rand_boxes(23, g)
    b2[:5] = b1[:5]                                   # identical boxes: every max / min is a tie
    b2[5, :] = torch.tensor([b1[5, 2], b1[5, 1], b1[5, 2] + 7.0, b1[5, 3]])      # touching: intersection width exactly 0
    w = torch.randn((37, 23), generator=g)
    r1, r2 = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    ref = getattr(ob, name)(r1, r2)
    (ref * w).sum().backward()
    d1, d2 = b1.cuda().requires_grad_(True), b2.cuda().requires_grad_(True)
    out = getattr(h.ops.boxes, name)(d1, d2)
    assert out.requires_grad
    (out * w.cuda()).sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert rel_l2(d1.grad.cpu(), r1.grad) < 1e-5, rel_l2(d1.grad.cpu(), r1.grad)
    assert rel_l2(d2.grad.cpu(), r2.grad) < 1e-5, rel_l2(d2.grad.cpu(), r2.grad)


def test_functional_norm_conv2d_matches_module_and_reference_formula():
    import holocron_amd as h
    torch.manual_seed(0)
    x = torch.rand(2, 8, 9, 9).to(torch.bfloat16).float()
    w = (torch.randn(16, 8, 3, 3) * 0.2).to(torch.bfloat16).float()
    b = torch.randn(16)
    # functional.py:346-352: every patch normalised over its Cin*KH*KW entries (biased variance, eps inside the rsqrt)
    cols = F.unfold(x, 3, padding=1).transpose(1, 2)
    cols = (cols - cols.mean(-1, keepdim=True)) * (cols.var(-1, unbiased=False, keepdim=True) + 1e-14).rsqrt()
    ref = (cols @ w.view(16, -1).t() + b).transpose(1, 2).reshape(2, 16, 9, 9)
    wg = w.cuda().requires_grad_(True)
    out = h.nn.functional.norm_conv2d(x.cuda(), wg, b.cuda(), padding=1)
    assert rel_l2(out.float().cpu(), ref) < 4e-3
    out.float().square().mean().backward()
    assert wg.grad is not None and torch.isfinite(wg.grad).all()
    mod = h.nn.NormConv2d(8, 16, 3, padding=1).cuda()
    with torch.no_grad():
        mod.weight.copy_(w)
        mod.bias.copy_(b)
    assert torch.equal(mod(x.cuda()), h.nn.functional.norm_conv2d(x.cuda(), mod.weight, mod.bias, padding=1))


@pytest.mark.parametrize("cfg", [(48, 48, 1, True), (32, 64, 2, False), (96, 96, 1, True)])
def test_repblock_backward_on_running_statistics(cfg):
    """eval mode (what freeze_bn leaves a frozen BatchNorm in, trainer/utils.py:26-30) with trainable convs: the block normalises
    with its running statistics, so dy = a * dz without the batch-statistics terms.  Against torch-CPU autograd of the block."""
    import holocron_amd as h
    from oracle import repvgg as orv
    cin, cout, stride, ident = cfg
    g = torch.Generator().manual_seed(cin + cout)
    blk = h.models.RepBlock(cin, cout, stride, ident)
    sd = blk.state_dict()
    for k, v in sd.items():
        if v.dim() == 4:
            v.copy_((torch.randn(v.shape, generator=g) * (2.0 / (v.shape[0] * v.shape[2] * v.shape[3])) ** 0.5).to(torch.bfloat16).float())
        elif k.endswith("running_var") or k.endswith("weight"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith("running_mean") or k.endswith("bias"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.2)
    state = {k: v.clone() for k, v in sd.items()}
    x = torch.rand((4, cin, 14, 14), generator=g).to(torch.bfloat16).float()
    blk = blk.cuda().eval()
    xg = x.cuda().requires_grad_(True)
    out = blk(xg)
    r = (torch.rand(out.shape, generator=g) + 0.5).to(torch.bfloat16).float()
    (out.float() * r.cuda()).sum().backward()
    osd = {"blk." + k: v.clone() for k, v in state.items()}
    keys = orv.trainable_keys(osd)
    for k in keys:
        osd[k].requires_grad_(True)
    xe = x.clone().requires_grad_(True)
    o = orv.rep_block(xe, osd, "blk", stride, ident, False)
    gr = torch.autograd.grad((o * r).sum(), [xe] + [osd[k] for k in keys])
    assert rel_l2(out.float().cpu(), o.detach()) < 4e-3
    assert rel_l2(xg.grad.float().cpu(), gr[0]) < 3e-2            # ReLU-kink flips of a 4-image batch
    for k, gref in zip(keys, gr[1:]):
        got = dict(blk.named_parameters())[k[4:]].grad.cpu()
        assert rel_l2(got, gref) < 3e-2, (k, rel_l2(got, gref))
    for k, v in blk.state_dict().items():                         # nothing moved the running statistics
        if "running" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), state[k]), k


def test_classification_trainer_on_hip_model(tmp_path):
    """references/classification/train.py:216-227 in miniature: a small RepVGG, HIP AdaBelief, two epochs of two batches, frozen
    first stage (freeze_model + freeze_bn), sync-free evaluation."""
    import holocron_amd as h
    torch.manual_seed(0)
    cfg = dict(num_blocks=[1, 1, 1, 1, 1], planes=[16, 16, 32, 64, 64], width_multiplier=1, final_width_multiplier=1)
    model = h.models.RepVGG(**cfg)
    g = torch.Generator().manual_seed(1)
    batches = [(torch.rand(8, 3, 64, 64, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(2)]
    opt = h.optim.AdaBelief(model.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6)
    tr = h.trainer.ClassificationTrainer(model, batches, batches, torch.nn.CrossEntropyLoss(label_smoothing=0.1), opt, gpu=0,
                                         output_file=str(tmp_path / "ck.pth"))
    w_frozen = model.features[0][0].branches[0][0].weight.detach().clone()
    w_live = model.head.weight.detach().clone()
    tr.fit_n_epochs(2, 1e-3, freeze_until="features.0", sched_type="onecycle")
    assert tr.step == 4 and tr.epoch == 2 and math.isfinite(tr.min_loss)
    assert torch.equal(model.features[0][0].branches[0][0].weight.detach(), w_frozen)
    assert not torch.equal(model.head.weight.detach(), w_live)
    assert not model.features[0][0].branches[0][1].training                  # frozen BatchNorm runs on its running statistics
    met = tr.evaluate()
    assert set(met) == {"val_loss", "acc1", "acc5"} and 0.0 <= met["acc1"] <= met["acc5"] <= 1.0
    # same numbers as the reference's per-batch host arithmetic (trainer/classification.py:60-66)
    model.eval()
    c1 = c5 = n = 0
    with torch.no_grad():
        for x, t in batches:
            o = model(x.cuda()).float().cpu()
            c1 += int((o.argmax(1) == t).sum())
            c5 += int((o.topk(5, dim=1)[1] == t[:, None]).any(1).sum())
            n += x.shape[0]
    assert abs(met["acc1"] - c1 / n) < 1e-6 and abs(met["acc5"] - c5 / n) < 1e-6
    losses = tr.check_setup(num_it=4, plot=False)
    assert len(losses) == 4 and all(math.isfinite(v) for v in losses)


def test_detection_trainer_on_hip_yolov4():
    """references/detection/train.py:205-218 in miniature: the model returns the loss dict in training mode and detections in eval."""
    import holocron_amd as h
    from holocron_amd.models.detection.yolov4 import YOLOv4
    torch.manual_seed(0)
    model = YOLOv4([(64, 1), (128, 1), (256, 1), (512, 1), (1024, 1)], num_classes=5, stem_channels=16)   # the neck needs 256 / 512 / 1024
    g = torch.Generator().manual_seed(2)

    def sample():
        imgs = [torch.rand(3, 128, 128, generator=g) for _ in range(2)]
        tgts = []
        for _ in range(2):
            k = 2
            b = torch.rand(k, 4, generator=g)
            b[:, :2] *= b[:, 2:]
            tgts.append({"boxes": torch.cat([b[:, :2], b[:, :2] + (1 - b[:, :2]) * b[:, 2:]], 1).clamp(0, 1),
                         "labels": torch.randint(0, 5, (k,), generator=g)})
        return imgs, tgts
    batches = [sample() for _ in range(2)]
    opt = h.optim.AdaBelief(model.parameters(), lr=1e-4, betas=(0.95, 0.99), eps=1e-6)
    tr = h.trainer.DetectionTrainer(model, batches, batches, None, opt, gpu=0, skip_nan_loss=True)
    before = model.head.head1[-1].weight.detach().clone()
    tr.fit_n_epochs(1, 1e-4, sched_type="cosine")
    assert tr.step == 2
    met = tr.evaluate()
    assert set(met) == {"loc_err", "clf_err", "det_err", "val_loss"}
    assert not torch.equal(before, model.head.head1[-1].weight.detach())
