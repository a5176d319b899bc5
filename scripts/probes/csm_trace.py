"""Per-phase cycle sums of one wave of the pipelined small-channel conv (library built with -DHC_CSM_TRACE:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHC_CSM_TRACE -shared -o scripts/probes/libcsm_trace.so
holocron_amd/csrc/conv_small.hip holocron_amd/csrc/conv_resident.hip)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from holocron_amd.ops import conv as cv  # noqa: E402
from holocron_amd._lib import ptr, stream  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcsm_trace.so"))
lib.hc_conv_small.restype = C.c_int32
NAMES = ["loop back-edge + copies", "vmcnt(0) wait", "top barrier", "DMA issue", "row 0 MFMAs + stats + staging", "staging barrier",
         "stores", "rows 1-2 MFMAs"]
for (N, Cc, H, mode) in [(256, 48, 112, 0), (256, 48, 112, 1), (256, 48, 56, 0)]:
    x = cv.to_cl_bf16(torch.randn(N, Cc, H, H, device="cuda"))
    w3 = torch.randn(Cc, Cc, 3, 3, device="cuda"); w1 = torch.randn(Cc, Cc, 1, 1, device="cuda")
    y3 = cv.empty_cl(N, Cc, H, H, "cuda"); y1 = cv.empty_cl(N, Cc, H, H, "cuda")
    stats = torch.zeros(2, 128, 2, Cc, device="cuda")
    d = cv.conv_small_desc(N, H, H, Cc, Cc, mode)
    if mode == 0:
        wp3, wp1 = cv.pack_weight(w3, 0), cv.pack_weight(w1, 0)
        d.srcA, d.srcB, d.w3, d.w1 = ptr(x), None, ptr(wp3), ptr(wp1)
        d.w3_rstride, d.w1_rstride = 9 * d.C, d.C
        d.out3, d.out1, d.resid, d.stats3, d.stats1 = ptr(y3), ptr(y1), None, ptr(stats[0]), ptr(stats[1])
    else:
        wpd = torch.empty((Cc, 10, Cc), dtype=torch.bfloat16, device="cuda")
        cv.pack_weight(w3, 1, out=wpd, tap0=0, T=10); cv.pack_weight(w1, 1, out=wpd, tap0=9, T=10)
        d.srcA, d.srcB, d.w3 = ptr(y3), ptr(y1), ptr(wpd)
        d.w1 = ptr(wpd) + 9 * d.C * 2
        d.w3_rstride, d.w1_rstride = 10 * d.C, 10 * d.C
        d.out3, d.out1, d.resid, d.stats3, d.stats1 = ptr(x), None, ptr(x), None, None
    for _ in range(3):
        rc = lib.hc_conv_small(C.byref(d), C.c_void_p(stream()))
    torch.cuda.synchronize()
    assert rc == 0, rc
    out = (C.c_ulonglong * 8)()
    assert lib.hc_conv_small_trace(out) == 0
    tot = sum(out)
    print(f"--- N={N} C={Cc} {H}x{H} mode={mode}: {tot} cycles in the steady-state loop of wave 0 / block 8")
    for n, v in zip(NAMES, out):
        print(f"  {n:34s} {v:10d}  {100.0 * v / max(tot, 1):5.1f} %")
