"""The multi-GPU step on one GPU: a process group of one rank over RCCL, the GradReducer forced on, and the step replayed as
two hipGraphs around one eager all-reduce (holocron_amd.parallel.GraphedStep) must train like the plain eager step.
(Runs last: it owns a process group.)"""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_blocks=[1, 1, 1, 1, 1], planes=[16, 16, 32, 64, 64], width_multiplier=1, final_width_multiplier=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(dev, x, t):
    import holocron_amd as h
    torch.manual_seed(0)
    m = h.models.RepVGG(**CFG).to(dev).train()
    opt = h.optim.AdaBelief(m.parameters(), lr=1e-3, betas=(0.95, 0.99), eps=1e-6)
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x), t, label_smoothing=0.1)
        loss.backward()
        loss_buf.copy_(loss.detach())

    return m, opt, fwd_bwd, loss_buf


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("comm_dtype,cut_backward", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_two_graph_dp_step_trains_like_eager(comm_dtype, cut_backward):
    import torch.distributed as dist
    from holocron_amd.parallel import BackwardCut, GradReducer, GraphedStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.rand((8, 3, 64, 64), device=dev, generator=g)
        t = torch.randint(0, 10, (8,), device=dev, generator=g)
        n_steps = 4

        # plain eager training
        m0, opt0, fb0, loss0 = _make(dev, x, t)
        init = [p.detach().clone() for p in m0.parameters()]
        losses0 = []
        for _ in range(n_steps):
            fb0()
            opt0.step()
            losses0.append(float(loss0.item()))

        # forced reducer + two graphs around the collective
        m1, opt1, fb1, loss1 = _make(dev, x, t)
        assert all(torch.equal(a, b) for a, b in zip(init, m1.parameters()))
        if cut_backward:                   # backward in two graphs, the rear bucket reduced behind the second one
            rear_mod = m1.features[-1][-1]
            rear = {id(p) for p in rear_mod.parameters()} | {id(p) for p in m1.head.parameters()}
            front_last = next(p for p in reversed(list(m1.parameters())) if id(p) not in rear)
            red = GradReducer(m1.parameters(), bucket_mb=64.0, comm_dtype=comm_dtype, force=True, new_bucket_at=[front_last])
            cut = BackwardCut(rear_mod)
            gs = GraphedStep([fb1, cut.continue_backward], opt1, red)
        else:
            red = GradReducer(m1.parameters(), bucket_mb=0.25, comm_dtype=comm_dtype, force=True)
            gs = GraphedStep(fb1, opt1, red)
        assert red.active and len(red.buckets) > 1 and red.flat.dtype == comm_dtype
        gs.capture()                       # runs step 1 eagerly (deferred reducer), then captures
        assert len(gs.graphs) == (2 if cut_backward else 1) and gs.final is not None and not red._hooks
        if cut_backward:
            assert [sp[0].data_ptr() for sp in gs.spans] == [b.flat.data_ptr() for b in red.buckets]
        losses1 = [float(loss1.item())]
        for _ in range(n_steps - 1):
            gs.run()
            losses1.append(float(loss1.item()))
        torch.cuda.synchronize()

        # same number of optimizer steps on both sides
        p0, p1 = next(iter(m0.parameters())), next(iter(m1.parameters()))
        assert opt0.state[p0]["step"] == opt1.state[p1]["step"] == n_steps
        # the loss trajectories agree (bf16 activations + atomics: not bit-equal)
        for a, b in zip(losses0, losses1):
            assert abs(a - b) < 5e-2 * max(1.0, abs(a)), (losses0, losses1)
        # the parameters moved, and in the same direction as the eager run
        d0 = torch.cat([(p.detach() - i).flatten() for p, i in zip(m0.parameters(), init)])
        d1 = torch.cat([(p.detach() - i).flatten() for p, i in zip(m1.parameters(), init)])
        assert float(d1.abs().max()) > 0
        assert _cos(d0, d1) > 0.9, _cos(d0, d1)
        # the communication buffer holds the last step's gradients (world = 1: sum == the gradient itself)
        got = torch.cat([v.float().flatten() for b in red.buckets for v in b.views])
        want = torch.cat([p.grad.float().flatten() for b in red.buckets for p in b.params])
        tol = 0.0 if comm_dtype == torch.float32 else 2.0 ** -8
        assert float((got - want).abs().max()) <= tol * float(want.abs().max()) + 0.0
    finally:
        if own:
            dist.destroy_process_group()
