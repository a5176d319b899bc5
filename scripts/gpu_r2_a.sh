#!/bin/bash
# round 2, GPU call A: full-size per-layer parity tests, the ZeroPool regression test, and a baseline bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
nproc > gpurun_out/r2a/host.txt; free -g >> gpurun_out/r2a/host.txt
( time timeout 1500 python -m pytest tests/test_gpu_fullsize_layers.py -x -q -s --durations=15 ) > gpurun_out/r2a/fullsize_layers.log 2>&1
echo "fullsize rc=$?" >> gpurun_out/r2a/summary.txt
( time timeout 600 python -m pytest tests/test_gpu_repvgg.py -x -q -s ) > gpurun_out/r2a/repvgg.log 2>&1
echo "repvgg rc=$?" >> gpurun_out/r2a/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo "bench rc=$?" >> gpurun_out/r2a/summary.txt
tail -5 gpurun_out/r2a/fullsize_layers.log; tail -3 gpurun_out/r2a/repvgg.log; cat gpurun_out/r2a/summary.txt; cat gpurun_out/r2a/bench.json
