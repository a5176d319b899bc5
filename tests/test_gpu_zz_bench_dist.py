"""bench.py's N > 1 code path on one GPU (HC_FORCE_DIST=1: process group of one rank over RCCL, GradReducer, three hipGraphs with the
all-reduces between them) against its single-graph path: in deterministic mode the two loss trajectories must agree BIT FOR BIT - the
fp32 wire, the bucket pack / unpack and the cut backward change where the work is launched from, never a value.
Both arms run with HC_WREP_DEFER=0: the deferred RepBlock weight gradients are GROUPED per backward call, and the cut backward of the
N > 1 path ends a call (and a group) at each cut - same-shaped blocks then land in groups of other sizes, whose split-K factor and
therefore fp32 summation order differ.  That is a rounding-order difference, not a value difference, but AdaBelief's first steps are
sign-like and amplify it to 5e-3 of the loss within a dozen steps; without grouping every launch of the two arms is the same launch.
(Runs late: each arm is a fresh process with its own process group.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(force_dist):
    env = dict(os.environ)
    env["HC_FORCE_DIST"] = "1" if force_dist else "0"
    env["HC_WREP_DEFER"] = "0"
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = "29541"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--profile-steps", "1",
           "--no-cpu-baseline", "--no-secondary", "--deterministic", "--loss-tail", "5", "--batch", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = next(ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{"))
    return json.loads(line)


def test_forced_dist_bench_step_matches_the_single_graph_step_bit_for_bit():
    single, dist = _bench(False), _bench(True)
    assert "hipGraph replay of the full step" in single["config"]["mode"], single["config"]["mode"]
    assert "all-reduce" in dist["config"]["mode"] and "hipGraphs" in dist["config"]["mode"], dist["config"]["mode"]
    a, b = single["config"]["loss_tail"], dist["config"]["loss_tail"]
    assert len(a) == 5 and a == b, (a, b)
    assert single["config"]["final_loss"] == dist["config"]["final_loss"]
    assert a[-1] < a[0] + 1e-3                   # ... and it trains
