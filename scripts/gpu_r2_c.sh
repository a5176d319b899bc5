#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
echo "== PF=2"; HC_WREP_PF=2 python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -8
( time timeout 1200 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py ) > gpurun_out/r2c/pytest.log 2>&1
tail -5 gpurun_out/r2c/pytest.log
( timeout 900 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2" ) 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
echo "bench rc=$?"; cat gpurun_out/r2c/bench.json | cut -c1-1700
