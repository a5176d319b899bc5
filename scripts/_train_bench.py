"""Shared timing loop of the secondary training benches: eager warm-up, optional hipGraph capture of the whole step (falls
back to eager launches when the capture fails), K timed steps bracketed by synchronisation."""
import time

import torch

from holocron_amd.parallel import GraphedStep


def timed_training(model, opt, x, t, loss_fn, steps, warmup, use_graph=True):
    dev = x.device
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), t)
        loss.backward()
        loss_buf.copy_(loss.detach())

    gstep = GraphedStep(fwd_bwd, opt)
    for _ in range(max(2, warmup)):
        gstep.eager()
    torch.cuda.synchronize()
    note = "eager"
    if use_graph:
        try:
            gstep.capture()
            probe = next(p for p in model.parameters() if p.dim() == 2)
            before = probe.detach().clone()
            gstep.run()
            torch.cuda.synchronize()
            if not torch.isfinite(loss_buf).item() or torch.equal(before, probe.detach()):
                raise RuntimeError("graph replay did not train")
            note = "hipGraph replay of the full step"
        except Exception as e:  # noqa: BLE001
            gstep.release()
            note = f"eager (graph capture failed: {type(e).__name__}: {str(e)[:80]})"
            torch.cuda.synchronize()

    for _ in range(warmup):
        gstep.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gstep.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(loss_buf), note
