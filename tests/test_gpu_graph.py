"""hipGraph replay must reproduce the eager step bit for bit: every kernel launch, scratch clear and reduction of a block's
forward + backward is captured once and replayed twice.  (hipMemsetAsync nodes did not survive this - the second replay read
stale workspaces - which is why the library clears scratch with a kernel, csrc/common.h hc_zero_async.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blocks():
    import holocron_amd as h
    from holocron_amd.models.classification.repvgg import RepBlock
    yield "mobileone 48->48 k4 s1", h.models.MobileOneBlock(48, 48, 4, 1), (8, 48, 32, 32)
    yield "mobileone 48->128 k2 s2", h.models.MobileOneBlock(48, 128, 2, 2), (8, 48, 32, 32)
    yield "rexblock se", h.models.ReXBlock(38, 50, 6, 2, use_se=True), (8, 38, 20, 20)
    yield "rexblock shortcut", h.models.ReXBlock(61, 61, 6, 1, use_se=True), (8, 61, 10, 10)
    yield "repblock identity", RepBlock(48, 48, 1), (8, 48, 28, 28)


def test_graph_replay_equals_eager():
    dev = torch.device("cuda:0")
    for name, m, shape in _blocks():
        torch.manual_seed(0)
        m = m.to(dev).train()
        x = torch.rand(shape, device=dev, requires_grad=True)
        holder = {}

        def step():
            for p in m.parameters():
                p.grad = None
            x.grad = None
            out = m(x)
            holder["r"] = holder.get("r", None) if holder.get("r", None) is not None else torch.rand(out.shape, device=dev)
            (out.float() * holder["r"]).sum().backward()
            holder["out"] = out.detach()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ref = {"out": holder["out"].float().clone(), "dx": x.grad.float().clone(),
               **{n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}}
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        for it in range(3):
            g.replay()
            torch.cuda.synchronize()
            got = {"out": holder["out"].float(), "dx": x.grad.float(), **{n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}}
            for k, v in ref.items():
                # atomics make the statistics / gate sums order-dependent in the last bits, which moves bf16 roundings (2^-8):
                # a replay that read a stale workspace is wrong at the O(1) level
                e = float((got[k] - v).norm() / (v.norm() + 1e-12))
                assert e < 2e-2, (name, "replay", it, k, e)


def test_wgrad_side_stream_matches_main_stream():
    """Weight gradients issued on the second stream (holocron_amd.ops.conv.set_wgrad_side_stream) must equal the ones of the
    single-stream step, eagerly and from a hipGraph replay (fork / join captured as a parallel branch), and autograd must
    have adopted the side-stream tensors as .grad (the join callback raises if it copied them on the main stream)."""
    import holocron_amd as h
    from holocron_amd.ops import conv as cv
    dev = torch.device("cuda:0")
    cfg = dict(num_blocks=[1, 2, 2, 1, 1], planes=[16, 16, 32, 64, 64], width_multiplier=1, final_width_multiplier=1)
    torch.manual_seed(0)
    m = h.models.RepVGG(**cfg).to(dev).train()
    x = torch.rand((8, 3, 64, 64), device=dev)
    t = torch.randint(0, 10, (8,), device=dev)

    def step():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(x), t).backward()

    def grads():
        return {n: p.grad.float().clone() for n, p in m.named_parameters()}

    def compare(ref, got, what):
        for k, v in ref.items():
            e = float((got[k] - v).norm() / (v.norm() + 1e-12))
            assert e < 2e-2, (what, k, e)

    was_on = cv.wgrad_side_stream_enabled()
    cv.set_wgrad_side_stream(False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # BatchNorm running statistics move every step but do not enter the training-mode gradients: same x, same weights
    ref = grads()
    cv.set_wgrad_side_stream(True)
    try:
        for it in range(2):
            step()
            torch.cuda.synchronize()
            compare(ref, grads(), f"eager side stream {it}")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        for it in range(3):
            for p in m.parameters():      # a replay that skipped the side branch would leave these zeros behind
                p.grad.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            compare(ref, grads(), f"replay side stream {it}")
    finally:
        cv.set_wgrad_side_stream(was_on)
