"""`--gpus N` starts N ranks by itself: `parallel.ensure_ranks` driven on CPU through a stub with bench.py's
argument contract, gloo backend, world size 2; and a launcher whose WORLD_SIZE disagrees with --gpus is refused loudly."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "_stub_bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_1_without_launcher_runs_in_process():
    r = subprocess.run([sys.executable, STUB, "--gpus", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout) == {"n_gpus": 1, "steps": 2, "sum_of_ranks": 1.0}


def test_gpus_2_without_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, STUB, "--gpus", "2", "--steps", "3"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout) == {"n_gpus": 2, "steps": 3, "sum_of_ranks": 3.0}      # ranks 0 and 1 both took part: 1 + 2


def test_launcher_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, STUB, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "disagrees with WORLD_SIZE=2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_py_calls_ensure_ranks_before_touching_the_gpu():
    """bench.py and the shared driver of the secondary benches route --gpus through ensure_ranks (on a CPU-only host the mismatch
    is reported before the 'needs an MI355X' exit, i.e. the check really comes first)."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE=1" in r.stderr, r.stderr[-2000:]
    src = open(os.path.join(root, "scripts", "_train_bench.py")).read()
    assert "ensure_ranks(a.gpus" in src


@pytest.mark.parametrize("bad", ["0", "-2"])
def test_nonpositive_gpus_refused(bad):
    r = subprocess.run([sys.executable, STUB, "--gpus", bad], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus must be >= 1" in r.stderr
