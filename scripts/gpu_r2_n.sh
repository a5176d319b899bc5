#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py tests/test_gpu_conv.py tests/test_gpu_conv_rows.py -q -x ) > gpurun_out/r2n/repvgg.log 2>&1
grep -E "passed|failed" gpurun_out/r2n/repvgg.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2_block" ) > gpurun_out/r2n/layers.log 2>&1
grep -E "passed|failed" gpurun_out/r2n/layers.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/r2n/bench.json
HC_CONV_ROWS48=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2n/bench_no48.json 2> gpurun_out/r2n/bench_no48.err; cut -c1-220 gpurun_out/r2n/bench_no48.json
