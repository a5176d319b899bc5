"""CPU: the drop-in surface the reference's training scripts bind to - names, signatures and behaviour of
`holocron.*` as imported by references/classification/train.py:30-36 and references/detection/train.py:29-32 - resolved through the
`holocron` alias of this repository.  No kernels run here (no GPU): the trainers are exercised with plain torch modules."""
import inspect
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_script_import_lists_resolve_through_the_alias():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == "holocron" or k.startswith("holocron.")]:
        if "reference" in (getattr(sys.modules[k], "__file__", "") or ""):
            pytest.skip("the reference itself is imported as `holocron` in this process")
    import holocron
    import holocron_amd
    assert holocron is holocron_amd
    # references/classification/train.py:30-36
    from holocron.models import classification
    from holocron.models.presets import CIFAR10 as CIF10
    from holocron.models.presets import IMAGENETTE
    from holocron.optim import AdaBelief, AdamP, AdEMAMix
    from holocron.trainer import ClassificationTrainer
    from holocron.utils.data import Mixup
    from holocron.utils.misc import find_image_size
    # references/detection/train.py:29-32
    from holocron.models import detection
    from holocron.trainer import DetectionTrainer
    assert callable(classification.__dict__["repvgg_a0"]) and callable(classification.__dict__["rexnet1_0x"])
    assert callable(detection.__dict__["yolov4"])
    assert len(IMAGENETTE.classes) == 10 and len(CIF10.classes) == 10 and len(IMAGENETTE.mean) == 3
    assert all(callable(o) for o in (AdaBelief, AdamP, AdEMAMix, Mixup, find_image_size, ClassificationTrainer, DetectionTrainer))
    # the same module objects under both names (no second copy of any state)
    import holocron.nn.functional as F1
    import holocron_amd.nn.functional as F2
    assert F1 is F2 and hasattr(F1, "norm_conv2d")
    # holocron/__init__.py:1 names
    for sub in ("models", "nn", "ops", "optim", "trainer", "transforms", "utils"):
        assert hasattr(holocron, sub), sub
    from holocron.trainer import (BinaryClassificationTrainer, SegmentationTrainer, Trainer, freeze_bn, freeze_model,
                                  split_normalization_params)
    assert issubclass(BinaryClassificationTrainer, ClassificationTrainer) and issubclass(SegmentationTrainer, Trainer)
    assert all(callable(f) for f in (freeze_bn, freeze_model, split_normalization_params))


def test_trainer_signature_matches_reference():
    """constructor and method names / parameters of trainer/core.py:26-104,277-451 (checked against the reference source when it
    is available in this container)"""
    from holocron_amd.trainer import Trainer
    want = ["self", "model", "train_loader", "val_loader", "criterion", "optimizer", "gpu", "output_file", "amp", "skip_nan_loss",
            "nan_tolerance", "gradient_acc", "gradient_clip", "on_epoch_end"]
    got = list(inspect.signature(Trainer.__init__).parameters)
    assert got[:len(want)] == want
    for name in ("set_device", "save", "load", "_fit_epoch", "to_cuda", "_to_cuda", "_backprop_step", "_get_loss", "_set_params",
                 "_reset_opt", "evaluate", "_eval_metrics_str", "_reset_scheduler", "fit_n_epochs", "find_lr", "plot_recorder",
                 "check_setup"):
        assert callable(getattr(Trainer, name)), name
    assert list(inspect.signature(Trainer.fit_n_epochs).parameters)[:6] == ["self", "num_epochs", "lr", "freeze_until", "sched_type",
                                                                            "norm_weight_decay"]
    if os.path.exists(os.path.join(REF, "holocron", "trainer", "core.py")):
        import ast
        tree = ast.parse(open(os.path.join(REF, "holocron", "trainer", "core.py")).read())
        cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Trainer"][0]
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef)]:
            ours = getattr(Trainer, fn.name, None)
            assert ours is not None, fn.name
            ref_args = [a.arg for a in fn.args.args]
            our_args = list(inspect.signature(ours).parameters)
            if isinstance(inspect.getattr_static(Trainer, fn.name), staticmethod):
                ref_args = [a for a in ref_args if a != "self"]
            assert our_args[:len(ref_args)] == ref_args, (fn.name, our_args, ref_args)


def _tiny():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                               torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(4, 6))


def test_freeze_helpers():
    from holocron_amd.trainer import freeze_bn, freeze_model, split_normalization_params
    m = _tiny()
    freeze_model(m, "1")                      # up to and including the BatchNorm
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert frozen == ["0.weight", "0.bias", "1.weight", "1.bias"]
    assert m[1].training is False and m[1].track_running_stats is False      # its statistics stop updating (utils.py:26-30)
    with pytest.raises(ValueError):
        freeze_model(m, "nope")
    freeze_model(m, None)
    assert all(p.requires_grad for p in m.parameters())
    norm, other = split_normalization_params(m)
    assert len(norm) == 2 and len(other) == 4
    with pytest.raises(ValueError):
        split_normalization_params(m, [int])
    m2 = _tiny()
    for p in m2[1].parameters():
        p.requires_grad_(False)
    freeze_bn(m2.train())
    assert not m2[1].training and m2[0].training


def test_helpers_match_reference_when_available():
    if not os.path.exists(os.path.join(REF, "holocron", "trainer", "utils.py")):
        pytest.skip("reference tree not present")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_trainer_utils", os.path.join(REF, "holocron", "trainer", "utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from holocron_amd.trainer import utils as ours
    for layer in (None, "0", "1", "5"):
        a, b = _tiny(), _tiny()
        ours.freeze_model(a, layer)
        ref.freeze_model(b, layer)
        assert [p.requires_grad for p in a.parameters()] == [p.requires_grad for p in b.parameters()]
        assert [m.training for m in a.modules()] == [m.training for m in b.modules()]
        na, oa = ours.split_normalization_params(a)
        nb, ob = ref.split_normalization_params(b)
        assert [tuple(p.shape) for p in na] == [tuple(p.shape) for p in nb] and [tuple(p.shape) for p in oa] == [tuple(p.shape) for p in ob]


def test_classification_trainer_cpu_loop(tmp_path):
    from holocron_amd.trainer import ClassificationTrainer
    m = _tiny()
    g = torch.Generator().manual_seed(1)
    batches = [(torch.randn(8, 3, 6, 6, generator=g), torch.randint(0, 6, (8,), generator=g)) for _ in range(6)]
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
    seen = []
    tr = ClassificationTrainer(m, batches, batches[:2], torch.nn.CrossEntropyLoss(), opt, gpu=None, gradient_acc=2,
                               gradient_clip=1.0, output_file=str(tmp_path / "ck.pth"), on_epoch_end=seen.append)
    before = [p.detach().clone() for p in m.parameters()]
    tr.fit_n_epochs(2, 0.05, norm_weight_decay=0.0)
    assert tr.epoch == 2 and tr.step == 12 and len(seen) == 2 and {"val_loss", "acc1", "acc5"} <= set(seen[0])
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, m.parameters()))
    assert len(opt.param_groups) == 2 and opt.param_groups[0]["weight_decay"] == 0.0       # norm parameters got their own group
    assert os.path.exists(tr.output_file)
    state = torch.load(tr.output_file, weights_only=False)
    assert set(state) == {"epoch", "step", "min_loss", "model"}                                # the reference's checkpoint keys
    tr.save(str(tmp_path / "full.pth"), with_optimizer=True)
    tr2 = ClassificationTrainer(_tiny(), batches, batches[:2], torch.nn.CrossEntropyLoss(),
                                torch.optim.SGD(_tiny().parameters(), lr=0.05, momentum=0.9), gpu=None)
    tr2.load(torch.load(str(tmp_path / "ck.pth"), weights_only=False))
    assert tr2.epoch == state["epoch"] and tr2.step == state["step"]
    # evaluate(): top-1 / top-5 like the reference's arithmetic on the host
    met = tr.evaluate()
    with torch.no_grad():
        m.eval()
        c1 = c5 = n = 0
        for x, t in batches[:2]:
            o = m(x)
            c1 += int((o.argmax(1) == t).sum())
            c5 += int((o.topk(5, dim=1)[1] == t[:, None]).any(1).sum())
            n += x.shape[0]
    assert abs(met["acc1"] - c1 / n) < 1e-9 and abs(met["acc5"] - c5 / n) < 1e-9
    # find_lr records an exponential sweep; frozen-everything raises like the reference
    tr.find_lr(num_it=5)
    assert len(tr.loss_recorder) == len(tr.lr_recorder) == 5 and tr.lr_recorder[1] > tr.lr_recorder[0]
    with pytest.raises(ValueError):
        tr.find_lr(num_it=100)
    losses = tr.check_setup(num_it=3, plot=False)
    assert len(losses) == 3
    for p in m.parameters():
        p.requires_grad_(False)
    with pytest.raises(AssertionError):
        tr._set_params()


def test_detection_assign_iou_and_metrics():
    from holocron_amd.trainer.detection import DetectionTrainer, assign_iou
    gt = torch.tensor([[0., 0., 10., 10.], [20., 20., 30., 30.], [0., 0., 9., 9.]])
    pred = torch.tensor([[0., 0., 10., 10.], [21., 21., 30., 30.], [100., 100., 110., 110.]])
    gi, pi = assign_iou(gt, pred, 0.5)
    assert sorted(zip(gi, pi)) == [(0, 0), (1, 1)]          # gt 2 also prefers pred 0 but gt 0 has the higher IoU
    s = DetectionTrainer._eval_metrics_str({"loc_err": 0.25, "clf_err": None, "det_err": 0.5})
    assert "25.00%" in s and "N/A" in s
    if os.path.exists(os.path.join(REF, "holocron", "trainer", "detection.py")):
        sys.path.insert(0, ROOT)
        from oracle import tv_ops
        torch.manual_seed(0)
        for _ in range(5):
            a = torch.rand(7, 4) * 50
            a[:, 2:] += a[:, :2] + 1
            b = a[torch.randperm(7)[:5]] + torch.rand(5, 4)
            b = torch.cat([b, b[:2] + 0.3])                  # duplicates: several ground truths compete for one prediction
            iou = tv_ops.box_iou(a, b).max(dim=1)
            kept = iou.values >= 0.5
            gi, pi = assign_iou(a, b, 0.5)
            assert len(set(pi)) == len(pi) and all(bool(kept[g]) for g in gi)
            for g_, p_ in zip(gi, pi):                       # each chosen pair is the best ground truth of its prediction
                rivals = [float(iou.values[k]) for k in range(a.shape[0]) if bool(kept[k]) and int(iou.indices[k]) == p_]
                assert abs(float(iou.values[g_]) - max(rivals)) < 1e-6


def test_transforms_resize_and_zoom_out():
    from holocron_amd.transforms import RandomZoomOut, Resize, ResizeMethod
    img = torch.rand(3, 40, 80)
    assert Resize((32, 32))(img).shape == (3, 32, 32)
    padded = Resize((32, 32), mode=ResizeMethod.PAD)(img)
    assert padded.shape == (3, 32, 32) and float(padded[:, :8].abs().max()) == 0.0 and float(padded[:, 8:24].abs().max()) > 0
    with pytest.raises(ValueError):
        Resize((32, 32), mode="pad")
    with pytest.raises(ValueError):
        Resize((32,))
    torch.manual_seed(0)
    z = RandomZoomOut((64, 64), scale=(0.3, 0.6))(img)
    assert z.shape == (3, 64, 64) and float((z == 0).float().mean()) > 0.3
    with pytest.raises(ValueError):
        RandomZoomOut((64, 64), scale=(0.9, 0.3))
