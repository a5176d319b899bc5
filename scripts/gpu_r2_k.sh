#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py tests/test_gpu_conv.py -q -x ) > gpurun_out/r2k/repvgg.log 2>&1
grep -E "passed|failed" gpurun_out/r2k/repvgg.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2_block" ) > gpurun_out/r2k/layers.log 2>&1
grep -E "passed|failed" gpurun_out/r2k/layers.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/r2k/bench.json
HC_STACK_FWD=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2k/bench_nostack.json 2> gpurun_out/r2k/bench_nostack.err; cut -c1-220 gpurun_out/r2k/bench_nostack.json
