#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "96@28 or 192@14 or 96_ or 192_" ) > gpurun_out/r2h/layers.log 2>&1
tail -5 gpurun_out/r2h/layers.log
( time timeout 1200 python -m pytest tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py tests/test_gpu_conv.py -q -x ) > gpurun_out/r2h/repvgg.log 2>&1
tail -5 gpurun_out/r2h/repvgg.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r2h/bench.json
HC_CONV_ROWS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2h/bench_norows.json 2> gpurun_out/r2h/bench_norows.err; cut -c1-300 gpurun_out/r2h/bench_norows.json
