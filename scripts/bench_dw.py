"""Depthwise 3x3 shape by shape at the rexnet1_0x batch-256 shapes: the strip kernels (HC_DW_TILE=0) against the LDS-tiled kernels -
stride-1 forward (with statistics), and the stride-2 forward / data gradient / weight gradient.  Re-runs itself in two child processes
(the switches are read once per process)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S2_SHAPES = [(256, 96, 112), (256, 256, 56), (256, 384, 28), (256, 768, 14)]      # channel counts as the padded model runs them
SHAPES = [(256, 32, 112), (256, 176, 56), (256, 304, 28), (256, 432, 14), (256, 96, 56), (256, 64, 112), (256, 128, 56)]


def child():
    import torch
    from holocron_amd import _lib
    from holocron_amd._lib import check, ptr, stream
    from holocron_amd.ops import conv as cv
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = {}
    for N, C, H in SHAPES:
        g = torch.Generator(device=dev).manual_seed(C)
        x = cv.to_cl_bf16(torch.rand((N, C, H, H), device=dev, generator=g) - 0.5)
        w = torch.randn((9, C), device=dev, generator=g)
        y = cv.empty_cl(N, C, H, H, dev)
        stats = torch.zeros((_lib.stat_replicas(), 2, C), device=dev)
        for _ in range(3):
            check(lib.hc_dw3x3_fwd(ptr(x), ptr(w), ptr(y), ptr(stats), N, H, H, C, 1, stream()), "dw")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(lib.hc_dw3x3_fwd(ptr(x), ptr(w), ptr(y), ptr(stats), N, H, H, C, 1, stream()), "dw")
        e1.record()
        torch.cuda.synchronize()
        out["%d@%d" % (C, H)] = {"us": e0.elapsed_time(e1) * 100.0, "crc": int(y.view(torch.int16).to(torch.int64).sum().item()),
                                 "mb": 2.0 * x.numel() * 2 / 1e6}
    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100.0

    def crc(t):
        return int(t.view(torch.int16).to(torch.int64).sum().item())

    for N, C, H in S2_SHAPES:
        g = torch.Generator(device=dev).manual_seed(C + 1)
        OH = (H - 1) // 2 + 1
        x = cv.to_cl_bf16(torch.rand((N, C, H, H), device=dev, generator=g) - 0.5)
        dy = cv.to_cl_bf16(torch.rand((N, C, OH, OH), device=dev, generator=g) - 0.5)
        w = torch.randn((9, C), device=dev, generator=g)
        y = cv.empty_cl(N, C, OH, OH, dev)
        dx = cv.empty_cl(N, C, H, H, dev)
        stats = torch.zeros((_lib.stat_replicas(), 2, C), device=dev)
        ws = torch.empty((lib.hc_dw3x3_wgrad_ws_bytes(C) // 4,), dtype=torch.float32, device=dev)
        dw = torch.zeros((C, 1, 3, 3), device=dev)
        mb = (x.numel() + y.numel()) * 2 / 1e6
        r = {"mb": mb}
        r["fwd"] = timed(lambda: check(lib.hc_dw3x3_fwd(ptr(x), ptr(w), ptr(y), ptr(stats), N, H, H, C, 2, stream()), "fwd"))
        r["dgrad"] = timed(lambda: check(lib.hc_dw3x3_dgrad(ptr(dy), ptr(w), None, ptr(dx), N, H, H, C, 2, stream()), "dgrad"))
        r["wgrad"] = timed(lambda: check(lib.hc_dw3x3_wgrad(ptr(x), ptr(dy), ptr(ws), ptr(dw), N, H, H, C, C, 2, 0, stream()), "wgrad"))
        r["crc"] = [crc(y), crc(dx)]
        r["dw"] = float(dw.double().abs().sum().item())
        out["s2 %d@%d" % (C, H)] = r
    print("RESULT " + json.dumps(out))


def run(env):
    e = dict(os.environ)
    e.update(env)
    e["DW_CHILD"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(r.stdout[-2000:] + r.stderr[-3000:])


if __name__ == "__main__":
    if os.environ.get("DW_CHILD") == "1":
        child()
    else:
        a, b = run({"HC_DW_TILE": "0"}), run({"HC_DW_TILE": "1"})
        print(f"{'shape':<10} {'MB':>7} {'strip us':>9} {'TB/s':>6} {'tile us':>9} {'TB/s':>6}  same bits")
        for k in a:
            if k.startswith("s2 "):
                continue
            print(f"{k:<10} {a[k]['mb']:>7.0f} {a[k]['us']:>9.1f} {a[k]['mb'] / a[k]['us']:>6.2f} {b[k]['us']:>9.1f} {b[k]['mb'] / b[k]['us']:>6.2f}  {a[k]['crc'] == b[k]['crc']}")
        print(f"{'stride 2':<12} {'MB':>6} | {'fwd us':>14} {'TB/s':>10} | {'dgrad us':>14} {'TB/s':>10} | {'wgrad us':>14} {'TB/s':>10} | same bits (fwd, dgrad)  dw rel diff")
        for k in a:
            if not k.startswith("s2 "):
                continue
            p, q = a[k], b[k]
            cols = " | ".join(f"{p[m]:>6.1f} ->{q[m]:>6.1f} {p['mb'] / p[m]:>4.2f}->{q['mb'] / q[m]:>4.2f}" for m in ("fwd", "dgrad", "wgrad"))
            print(f"{k:<12} {p['mb']:>6.0f} | {cols} | {p['crc'] == q['crc']}  {abs(p['dw'] - q['dw']) / max(p['dw'], 1e-9):.2e}")
