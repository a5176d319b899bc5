"""Build libholocron_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The library is plain C ABI (include/holocron_hip.h) and has no torch dependency; it is built
in-tree (holocron_amd/lib/) so that it travels with the source snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libholocron_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = {
    # the epilogue's store loop is fully unrolled over the tile's accumulator quads (24-32 copies of its body); past LLVM's default
    # pragma-unroll budget the loop stays rolled, the accumulator indices become dynamic and the 96-384 accumulator registers of a wave
    # move to SCRATCH (round 6: the 192-channel tile lost 50 % that way when the inference epilogue grew the body -
    # tests/test_host_logic.py::test_gather_conv_accumulators_stay_in_registers keeps watch)
    "conv_gather.hip": ["-mllvm", "-pragma-unroll-threshold=131072"],
    "conv_pointwise.hip": [],
    "conv_small.hip": [],
    "conv_wgrad.hip": [],
    "conv_wgrad_tr.hip": [],
    "conv_wgrad_dma.hip": [],
    "conv_wgrad_rep.hip": [],
    "conv_rows.hip": [],
    "conv_rows48.hip": [],
    "conv_s2.hip": [],
    "rep_bn.hip": [],
    "optim.hip": [],
    "nhwc_ops.hip": [],
    "dwconv.hip": [],
    "se_mlp.hip": [],
    "mobileone.hip": [],
    # separate torch kernels in the reference round after every op: no fused multiply-add here
    "pointwise.hip": ["-ffp-contract=off"],
    "losses.hip": ["-ffp-contract=off"],
    "yolo.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _deps():
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "..", "include", "holocron_hip.h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(item):
    src, flags = item
    obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _deps()):
        return obj, False
    cmd = [HIPCC] + COMMON + flags + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(LIBDIR):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(LIBDIR, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(_compile, SOURCES.items()))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built" if rebuilt else "up to date", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
