"""Big-tile gather-conv family (csrc/conv_gather.hip) against the 128 x 128 form, shape by shape: forward 3x3 / 1x1 convs (with
BatchNorm statistics) of the YOLOv4 608^2 batch-16, repvgg_a2 and repvgg_a0 stride-1 layers.  The dispatch switch is read once per
process, so this script re-runs itself: HC_CONV_BIG=0 (128 x 128 form) and the default (the family wherever its efficiency predicate picks it; the threshold knob of round 4 is gone)
and prints both times, the launch efficiency the predicate sees, and whether the two results are bit-identical."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # name, N, Cin, H, Cout, k
    ("yolo 64@152 3x3", 16, 64, 152, 64, 3), ("yolo 128@76 3x3", 16, 128, 76, 128, 3), ("yolo 128->256@76 3x3", 16, 128, 76, 256, 3),
    ("yolo 256@38 3x3", 16, 256, 38, 256, 3), ("yolo 256->512@38 3x3", 16, 256, 38, 512, 3), ("yolo 512@19 3x3", 16, 512, 19, 512, 3),
    ("yolo 512->1024@19 3x3", 16, 512, 19, 1024, 3), ("yolo 1024->512@19 1x1", 16, 1024, 19, 512, 1),
    ("yolo 512->256@38 1x1", 16, 512, 38, 256, 1), ("yolo 2048->512@19 1x1", 16, 2048, 19, 512, 1),
    ("a2 96@56 3x3 bs256", 256, 96, 56, 96, 3), ("a2 192@28 3x3 bs256", 256, 192, 28, 192, 3), ("a2 384@14 3x3 bs256", 256, 384, 14, 384, 3),
    ("a1 64@56 3x3 bs256", 256, 64, 56, 64, 3), ("a1 128@28 3x3 bs256", 256, 128, 28, 128, 3), ("a1 256@14 3x3 bs256", 256, 256, 14, 256, 3),
    ("a0 192@14 3x3 bs256", 256, 192, 14, 192, 3), ("a0 1280@7 3x3 bs256", 256, 1280, 7, 1280, 3),
]


def child():
    import torch
    from holocron_amd import _lib
    from holocron_amd.ops import conv as cv
    dev = torch.device("cuda:0")
    out = {}
    for name, N, Cin, H, Cout, k in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        x = cv.to_cl_bf16(torch.rand((N, Cin, H, H), device=dev, generator=g) - 0.5)
        w = ((torch.rand((Cout, k * k, Cin), device=dev, generator=g) - 0.5) * 0.1).to(torch.bfloat16)
        y = cv.empty_cl(N, Cout, H, H, dev)
        d = cv.fwd_desc(N, Cin, H, H, Cout, k, k, 1, k // 2)
        stats = torch.zeros((_lib.stat_replicas(), 2, Cout), dtype=torch.float32, device=dev)
        for _ in range(3):
            cv.launch_conv(d, x, w, y, stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            cv.launch_conv(d, x, w, y, stats=stats)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        stats.zero_()
        cv.launch_conv(d, x, w, y, stats=stats)
        yf = y.float()
        out[name] = {"us": us, "sum": float(yf.double().sum()), "abs": float(yf.double().abs().sum()),
                     "crc": int(y.view(torch.int16).to(torch.int64).sum().item()), "s1": float(stats[:, 0].double().sum()),
                     "s2": float(stats[:, 1].double().sum())}
    print("RESULT " + json.dumps(out))


def run(env):
    e = dict(os.environ)
    e.update(env)
    e["BIGTILE_CHILD"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(r.stdout[-2000:] + r.stderr[-4000:])


def main():
    a, b = run({"HC_CONV_BIG": "0"}), run({"HC_CONV_BIG": "1"})      # the family under its own dispatch predicate (efficiency >= 0.70)
    print(f"{'shape':<26} {'GFLOP':>7} {'128x128 us':>10} {'TF/s':>6} {'family us':>10} {'TF/s':>6} {'ratio':>6} {'tile':>9} {'eff':>5}  same bits / stats")
    for name, N, Cin, H, Cout, k in SHAPES:
        fl = 2.0 * N * H * H * Cout * Cin * k * k
        M = N * H * H
        if Cout <= 96: bc, bp = 96, 512
        elif Cout <= 128: bc, bp = 128, 512
        elif Cout <= 192: bc, bp = 192, 256
        elif 256 < Cout <= 384: bc, bp = 384, 128
        else: bc, bp = 256, 256
        ct, pt = -(-Cout // bc), -(-M // bp)
        tiles = ct * pt
        rounds = -(-tiles // 256)
        eff = Cout / (ct * bc) * M / (pt * bp) * tiles / (rounds * 256)
        ra, rb = a[name], b[name]
        same = ra["crc"] == rb["crc"]
        st = abs(ra["s1"] - rb["s1"]) <= 1e-5 * max(1.0, abs(ra["s1"])) and abs(ra["s2"] - rb["s2"]) <= 1e-5 * max(1.0, abs(ra["s2"]))
        big_used = (Cin * k * k // 32 >= 16) and Cin % 32 == 0 and Cout >= 64
        print(f"{name:<26} {fl / 1e9:>7.1f} {ra['us']:>10.1f} {fl / ra['us'] / 1e6:>6.0f} {rb['us']:>10.1f} {fl / rb['us'] / 1e6:>6.0f} "
              f"{ra['us'] / rb['us']:>6.2f} {bc:>4}x{bp:<4} {eff:>5.2f}  {same} / {st}" + ("" if big_used else "   (outside the family's rules)"))


if __name__ == "__main__":
    child() if os.environ.get("BIGTILE_CHILD") == "1" else main()
