"""Shared timing loop of the secondary training benches: eager warm-up, optional hipGraph capture of the whole step (falls
back to eager launches when the capture fails), K timed steps bracketed by synchronisation."""
import time

import torch


def timed_training(model, opt, x, t, loss_fn, steps, warmup, use_graph=True):
    dev = x.device
    loss_buf = torch.zeros((), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), t)
        loss.backward()
        opt.step()
        loss_buf.copy_(loss.detach())

    for _ in range(max(2, warmup)):
        step()
    torch.cuda.synchronize()
    graph, note = None, "eager"
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                step()
            torch.cuda.synchronize()
            probe = next(p for p in model.parameters() if p.dim() == 2)
            before = probe.detach().clone()
            opt.advance_for_replay()
            gph.replay()
            torch.cuda.synchronize()
            if not torch.isfinite(loss_buf).item() or torch.equal(before, probe.detach()):
                raise RuntimeError("graph replay did not train")
            graph, note = gph, "hipGraph replay of the full step"
        except Exception as e:  # noqa: BLE001
            graph, note = None, f"eager (graph capture failed: {type(e).__name__}: {str(e)[:80]})"
            torch.cuda.synchronize()

    def run_step():
        if graph is not None:
            opt.advance_for_replay()
            graph.replay()
        else:
            step()

    for _ in range(warmup):
        run_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(loss_buf), note
