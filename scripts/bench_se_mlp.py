"""hc_se_mlp_fwd / hc_se_mlp_bwd at the squeeze-excite shapes of rexnet1_0x, batch 256: time per call (HIP events).  Under
`rocprofv3 --kernel-trace` the per-kernel times come out by grid size (scripts/prof_round4.sh: summ)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from holocron_amd import _lib
from holocron_amd._lib import check, stream

SHAPES = [(256, 228, 256, 19), (256, 432, 448, 36), (256, 702, 704, 58), (256, 1044, 1088, 87)]
lib = _lib.load()
dev = torch.device("cuda:0")
for N, Cc, Cp, R in SHAPES:
    g = torch.Generator(device=dev).manual_seed(Cc)
    nf = int(lib.hc_se_mlp_part_floats(N, R))
    f = lambda *s: torch.rand(s, device=dev, generator=g)
    pooled, w1, gamma, beta, w2, b2 = f(N, Cp), f(R, Cc) - 0.5, f(R) + 0.5, f(R), f(Cc, R) - 0.5, f(Cc)
    rm, rv, nbt = f(R), f(R) + 0.5, torch.zeros((), dtype=torch.int64, device=dev)
    h1, part, stat, gbuf, part2 = f(N * R), f(nf), f(2 * R), f(N * R), f(nf)
    lg = torch.empty((N, Cp), dtype=torch.bfloat16, device=dev)
    dl = (f(N, Cp) - 0.5).to(torch.bfloat16)
    dpool, dw1, dw2, dgb, db2 = f(N, Cp), f(R, Cc), f(Cc, R), f(2, R), f(Cc)
    d = _lib.SeMlpDesc()
    d.pooled, d.w1, d.gamma, d.beta, d.w2, d.b2 = (t.data_ptr() for t in (pooled, w1, gamma, beta, w2, b2))
    d.running_mean, d.running_var, d.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
    d.h1, d.part, d.stat, d.logits = h1.data_ptr(), part.data_ptr(), stat.data_ptr(), lg.data_ptr()
    d.dl, d.g, d.part2, d.dpool, d.dw1, d.dw2 = (t.data_ptr() for t in (dl, gbuf, part2, dpool, dw1, dw2))
    d.dgamma, d.dbeta, d.db2 = dgb.data_ptr(), dgb.data_ptr() + 4 * R, db2.data_ptr()
    d.N, d.C, d.Cp, d.R, d.act, d.eps, d.momentum = N, Cc, Cp, R, 6, 1e-5, 0.1
    out = []
    for name, fn in (("fwd", lib.hc_se_mlp_fwd), ("bwd", lib.hc_se_mlp_bwd)):
        for _ in range(3):
            check(fn(C.byref(d), stream()), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            check(fn(C.byref(d), stream()), name)
        e1.record()
        torch.cuda.synchronize()
        out.append("%s %.1f us" % (name, e0.elapsed_time(e1) * 50.0))
    print("N %d C %d R %d: %s" % (N, Cc, R, ", ".join(out)))
