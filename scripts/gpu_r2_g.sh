#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -s ) > gpurun_out/r2g/fullsize.log 2>&1
tail -25 gpurun_out/r2g/fullsize.log
( time timeout 1200 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py --deselect tests/test_gpu_fullsize.py ) > gpurun_out/r2g/pytest.log 2>&1
tail -6 gpurun_out/r2g/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?"; cut -c1-420 gpurun_out/r2g/bench.json
