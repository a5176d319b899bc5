#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py ) > gpurun_out/r2i/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2i/pytest.log | tail -3
timeout 200 python scripts/bench_ew.py 2>&1 | grep "C=" | cut -c1-250
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/r2i/bench.json
