"""The headline configuration at its FULL size (repvgg_a0, batch 256, 3 x 224 x 224: BASELINE.json configs[1]) through
properties that need no oracle - the CPU restatement takes ~40 s per step at this size:

* a training step is invariant under a permutation of the batch (BatchNorm statistics, the loss and every parameter
  gradient are sums over the batch) - this runs every kernel of the step at the benchmark's launch geometry: the persistent
  48-channel kernels with 112 tiles per workgroup, the 41-way split weight gradients, the 1280-channel DMA kernels;
* in eval mode images are independent: the batch of 256 equals four batches of 64;
* the re-parametrised network (repvgg.py:75-107) equals the three-branch one.

Tolerances: activations are bf16 in HBM and the statistics are accumulated with atomics (order-dependent in the last bits),
so two runs of the SAME input already differ (measured inside the test); the bounds are small multiples of that floor and far
below what a wrong tile / halo / split would produce (O(1))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_repvgg_a0_bs256_properties():
    import holocron_amd as h
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = h.models.repvgg_a0(num_classes=10).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((256, 3, 224, 224), device=dev, generator=g)
    t = torch.randint(0, 10, (256,), device=dev, generator=g)
    names = ["features.0.0.branches.0.0.weight", "features.0.1.branches.0.0.weight", "features.1.0.branches.1.0.weight",
             "features.3.5.branches.0.0.weight", "features.4.1.branches.0.0.weight", "features.4.1.branches.0.1.weight",
             "features.4.1.branches.1.1.bias", "features.4.1.branches.2.weight", "head.weight", "head.bias"]
    params = dict(m.named_parameters())

    def step(xx, tt):
        for p in m.parameters():
            p.grad = None
        logits = m(xx)
        loss = torch.nn.functional.cross_entropy(logits, tt, label_smoothing=0.1)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().float(), float(loss), {n: params[n].grad.detach().float().clone() for n in names}

    lg0, loss0, gr0 = step(x, t)
    assert torch.isfinite(lg0).all() and loss0 == loss0
    lg1, loss1, gr1 = step(x, t)                      # the same input again: the noise floor of the atomics
    perm = torch.randperm(256, device=dev, generator=g)
    lgp, lossp, grp = step(x[perm].contiguous(), t[perm].contiguous())
    # Two runs of the same input differ by ~2e-2 on the 10 logits of this randomly initialised 28-block network (1-ulp bf16 flips
    # from the order of the statistics atomics, amplified layer by layer); the permuted run must stay within a small multiple of
    # that floor.  A wrong tile, halo or split would be O(1).
    floor = max(_rel(lg1, lg0), 1e-3)
    e_lg = _rel(lgp, lg0[perm])
    print(f"logits: floor {floor:.3e} permuted {e_lg:.3e}; loss {loss0:.5f} {loss1:.5f} {lossp:.5f}")
    assert floor < 0.1, floor
    assert e_lg < max(5e-2, 3 * floor), (e_lg, floor)
    assert abs(lossp - loss0) < max(1e-2, 3 * abs(loss1 - loss0)) * max(1.0, abs(loss0)), (lossp, loss0, loss1)
    # Gradients: only the last block and the head are stable run to run (measured floors: head 1e-2, last block's BatchNorm 4e-2,
    # its 1280 x 1280 conv 0.13, every earlier conv ~0.5: behind a few blocks of ReLU masks that flip with the last bf16 bit the
    # batch gradient of a randomly initialised network is noise-dominated - the fp32 reference shows the same sensitivity,
    # DESIGN.md §6), so each gradient is held to ITS OWN floor and the tight check needs at least two stable tensors.
    checked = 0
    report = []
    for n in names:
        assert torch.isfinite(grp[n]).all(), n
        e, f = _rel(grp[n], gr0[n]), max(_rel(gr1[n], gr0[n]), 1e-3)
        report.append((n, f, e))
        print(f"{n}: floor {f:.3e} permuted {e:.3e}")
    # only tensors that are stable run to run can say anything about the permutation: for the others (floor ~0.5: every conv
    # gradient in front of the last block) a bound of "3 x floor" passes a zero or an uncorrelated gradient, so none is asserted -
    # those kernels are compared with the fp32 reference layer by layer at this size in test_gpu_fullsize_layers.py, and the whole
    # step is checked for bit-reproducibility in test_repvgg_a0_bs256_deterministic_mode below
    for n, f, e in report:
        if f < 0.1:
            assert e < 3 * f + 1e-2, (n, e, f)
            checked += 1
    assert checked >= 2, report

    # eval: images are independent
    m.eval()
    with torch.no_grad():
        full = m(x).float()
        parts = torch.cat([m(x[i:i + 64].contiguous()).float() for i in range(0, 256, 64)])
    print(f"eval 4 x 64 vs 256: {_rel(parts, full):.3e}")
    assert _rel(parts, full) < 1e-2, _rel(parts, full)

    # re-parametrised network == three-branch network
    m2 = h.models.repvgg_a0(num_classes=10)
    m2.load_state_dict(m.state_dict())
    m2 = m2.to(dev).eval()
    m2.reparametrize()
    with torch.no_grad():
        rep = m2(x).float()
    print(f"re-parametrised vs three-branch: {_rel(rep, full):.3e}")
    assert _rel(rep, full) < 5e-2, _rel(rep, full)


def test_repvgg_a0_bs256_deterministic_mode():
    """hc_set_deterministic(1): every workgroup owns its slot of the statistics accumulators (32768 replicas instead of 128, added in
    a fixed order by the finalize kernels), the split reductions are single-writer.  Two training steps from the same state on the
    same input are then BIT-identical - loss, logits, every gradient, every updated parameter and running statistic - at the
    benchmark's size, i.e. with the launch geometries of the bench (SURVEY.md hard-part 3)."""
    import holocron_amd as h
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((256, 3, 224, 224), device=dev, generator=g)
    t = torch.randint(0, 10, (256,), device=dev, generator=g)
    h.set_deterministic(True)
    try:
        runs = []
        for _ in range(2):
            torch.manual_seed(0)
            m = h.models.repvgg_a0(num_classes=10).to(dev).train()
            opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6)
            out = []
            for _step in range(2):
                opt.zero_grad(set_to_none=True)
                logits = m(x)
                loss = torch.nn.functional.cross_entropy(logits, t, label_smoothing=0.1)
                loss.backward()
                grads = [p.grad.detach().clone() for p in m.parameters()]
                opt.step()
                out.append((loss.detach().clone(), logits.detach().clone(), grads))
            torch.cuda.synchronize()
            runs.append((out, {k: v.detach().clone() for k, v in m.state_dict().items()}))
        (o0, s0), (o1, s1) = runs
        for (l0, lg0, g0), (l1, lg1, g1) in zip(o0, o1):
            assert torch.equal(l0, l1) and torch.equal(lg0, lg1)
            for a, b in zip(g0, g1):
                assert torch.equal(a, b)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k
        assert torch.isfinite(o0[1][0])
    finally:
        h.set_deterministic(False)
