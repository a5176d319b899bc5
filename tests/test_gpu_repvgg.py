"""GPU: the RepVGG path (RepBlock fwd/bwd, whole-model train steps, eval, reparametrisation) against
the reference's golden vectors and the oracle.  bf16 activations: rel-L2 tolerances stated inline."""
import pytest
import torch

from conftest import close_frac, rel_l2

pytestmark = pytest.mark.gpu


def _mk_block(cfg, state):
    import holocron_amd as h
    cin, cout, stride, ident = cfg
    blk = h.models.RepBlock(cin, cout, stride, ident)
    blk.load_state_dict(state)
    return blk.cuda()


def test_repblock_train_matches_reference(golden):
    for c in golden("repblock.pt"):
        cin, cout, stride, ident = c["cfg"]
        blk = _mk_block(c["cfg"], c["state"]).train()
        x = c["x"].cuda().requires_grad_(cin % 16 == 0)
        out = blk(x)
        assert rel_l2(out.float().cpu(), c["out"]) < 4e-3, c["cfg"]
        (out.float() * c["r"].cuda()).sum().backward()
        # Gradients: an activation that sits within bf16 rounding of the ReLU kink flips its mask and
        # moves the gradients it feeds by O(1) (a handful of the 2*H*W*C elements per case).  So the
        # check is element-wise with a small allowed outlier fraction, not a norm.
        if cin % 16 == 0:
            scale = float(c["dx"].abs().mean())
            assert close_frac(x.grad.float().cpu(), c["dx"], 2e-2, 2e-2 * scale) > 0.97, c["cfg"]
        for n, p in blk.named_parameters():
            ref = c["dparams"][n]
            scale = float(ref.abs().mean())
            frac = close_frac(p.grad.cpu(), ref, 3e-2, 3e-2 * scale)
            assert frac > 0.9, (c["cfg"], n, frac)
        sd = blk.state_dict()
        for k, v in c["state_after"].items():
            if "running" in k:
                assert torch.allclose(sd[k].cpu(), v, rtol=2e-3, atol=2e-3), (c["cfg"], k)
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(v)


def test_repblock_eval_and_reparam_match_reference(golden):
    for c in golden("repblock.pt"):
        blk = _mk_block(c["cfg"], c["state_after"]).eval()
        with torch.no_grad():
            out = blk(c["x"].cuda())
            assert rel_l2(out.float().cpu(), c["out_eval"]) < 4e-3, c["cfg"]
            blk.reparametrize()
            assert torch.allclose(blk.branches.weight.cpu(), c["rep_weight"], rtol=1e-5, atol=1e-6)
            assert torch.allclose(blk.branches.bias.cpu(), c["rep_bias"], rtol=1e-4, atol=1e-5)
            rep = blk(c["x"].cuda())
            assert rel_l2(rep.float().cpu(), c["out_rep"]) < 6e-3, c["cfg"]


def test_repvgg_small_train_steps_match_reference(golden):
    import holocron_amd as h
    g = golden("repvgg_small.pt")
    m = h.models.RepVGG(**g["cfg"])
    m.load_state_dict(g["state"])
    m = m.cuda().train()
    opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6, weight_decay=0.0)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    x, t = g["x"].cuda(), g["target"].cuda()
    for si, step in enumerate(g["steps"]):
        opt.zero_grad()
        logits = m(x)
        loss = crit(logits, t)
        loss.backward()
        assert rel_l2(logits.float().cpu(), step["logits"]) < 3e-2, si
        assert abs(float(loss) - float(step["loss"])) < 3e-2 * max(1.0, abs(float(step["loss"])))
        worst = 0.0
        for n, p in m.named_parameters():
            ref = step["grads"][n]
            worst = max(worst, rel_l2(p.grad.cpu(), ref))
        assert worst < 0.15, worst        # deep bf16 chain with batch size 4: loose on the worst tensor
        opt.step()
    with torch.no_grad():
        m.eval()
        ev = m(x)
        m.reparametrize()
        ev_rep = m(x)
    assert rel_l2(ev_rep.float().cpu(), ev.float().cpu()) < 2e-2


def test_repvgg_a0_forward_vs_oracle_eval():
    """full-size architecture, small batch: logits against the CPU oracle on a shared state_dict"""
    import holocron_amd as h
    from oracle import repvgg as orv
    torch.manual_seed(0)
    m = h.models.repvgg_a0()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 4:
                p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(2, 3, 96, 96).to(torch.bfloat16).float()
    nb, a, b = orv.ARCH["repvgg_a0"]
    ch = orv.widths(orv.PLANES, a, b)
    with torch.no_grad():
        ref = orv.forward(sd, x, nb, ch, training=False)
        out = m.cuda().eval()(x.cuda())
    assert rel_l2(out.float().cpu(), ref) < 3e-2
