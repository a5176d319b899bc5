"""Per-layer roofline of rexnet1_0x's 1 x 1 convolutions at batch 256 (channel counts as the padded model runs them): the forward
launch (with BatchNorm statistics) and the data-gradient launch of every expand / project convolution and of the last 1 x 1, timed
with HIP events; bytes = input + output maps (bf16) + weights, roof = max(bytes / 8 TB/s, flop / 2.5 PF).  Prints a table
(profiles/rNN_rexnet_pointwise_layers.txt): the per-layer evidence behind "the pointwise family sits at a quarter of its HBM roof"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import holocron_amd as h
from holocron_amd import _lib
from holocron_amd.models.classification.rexnet import ReXBlock
from holocron_amd.nn.mbconv_op import ceil16
from holocron_amd.ops import conv as cv

N = int(os.environ.get("PW_N", "256"))
dev = torch.device("cuda:0")
m = h.models.rexnet1_0x(num_classes=1000)
layers, H = [], 112
for mod in m.features:
    if isinstance(mod, ReXBlock):
        expand, dw, se, act, project = mod._plan()
        if expand is not None:
            layers.append(("expand", expand[0].in_channels, expand[0].out_channels, H))
        H = (H - 1) // dw[0].stride[0] + 1
        layers.append(("project", project[0].in_channels, project[0].out_channels, H))
last = [mm for mm in m.features if isinstance(mm, torch.nn.Conv2d)]
if last:
    layers.append(("last", last[-1].in_channels, last[-1].out_channels, H))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f"{'layer':<8} {'Cin->Cout (padded)':<24} {'map':>4} {'MB':>7} {'roof us':>8} | {'fwd us':>8} {'TB/s':>6} {'frac':>5} | {'dgrad us':>8} {'TB/s':>6} {'frac':>5}")
tot = [0.0, 0.0, 0.0]
g = torch.Generator(device=dev).manual_seed(0)
for kind, ci, co, Hh in layers:
    cip, cop = ceil16(ci), ceil16(co)
    x = cv.to_cl_bf16(torch.rand((N, cip, Hh, Hh), device=dev, generator=g) - 0.5)
    dy = cv.to_cl_bf16(torch.rand((N, cop, Hh, Hh), device=dev, generator=g) - 0.5)
    y, dx = cv.empty_cl(N, cop, Hh, Hh, dev), cv.empty_cl(N, cip, Hh, Hh, dev)
    wf = (torch.rand((cop, 1, cip), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    wb = (torch.rand((cip, 1, cop), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    stats = torch.zeros((_lib.stat_replicas(), 2, cop), device=dev)
    df = cv.fwd_desc(N, cip, Hh, Hh, cop, 1, 1, 1, 0)
    db = cv.fwd_desc(N, cop, Hh, Hh, cip, 1, 1, 1, 0)
    tf = timed(lambda: cv.launch_conv(df, x, wf, y, stats=stats))
    tb = timed(lambda: cv.launch_conv(db, dy, wb, dx))
    nbytes = (x.numel() + y.numel() + wf.numel()) * 2
    flop = 2.0 * N * Hh * Hh * cip * cop
    roof = max(nbytes / 8e12, flop / 2.5e15) * 1e6
    tot[0] += roof; tot[1] += tf; tot[2] += tb
    print(f"{kind:<8} {('%d->%d (%d->%d)' % (ci, co, cip, cop)):<24} {Hh:>4} {nbytes / 1e6:>7.1f} {roof:>8.1f} | {tf:>8.1f} {nbytes / tf / 1e6:>6.2f} {roof / tf:>5.2f} | "
          f"{tb:>8.1f} {nbytes / tb / 1e6:>6.2f} {roof / tb:>5.2f}")
print(f"total: roof {tot[0]:.0f} us per direction, forward {tot[1]:.0f} us ({tot[0] / tot[1]:.2f}), data gradient {tot[2]:.0f} us ({tot[0] / tot[2]:.2f})")
