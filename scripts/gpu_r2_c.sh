#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
echo "bench rc=$?"; cat gpurun_out/r2c/bench.json | cut -c1-1800
HC_WREP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/bench_nowrep.json 2> gpurun_out/r2c/bench_nowrep.err
echo "bench(no wrep) rc=$?"; cut -c1-400 gpurun_out/r2c/bench_nowrep.json
( time timeout 1200 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py ) > gpurun_out/r2c/pytest.log 2>&1
tail -15 gpurun_out/r2c/pytest.log
