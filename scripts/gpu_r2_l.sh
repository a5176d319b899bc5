#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py ) > gpurun_out/r2l/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r2l/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/r2l/bench.json
HC_CONV_STAGED_STORES=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench_ns.json 2> gpurun_out/r2l/bench_ns.err; cut -c1-220 gpurun_out/r2l/bench_ns.json
timeout 600 python scripts/bench_yolov4.py --batch 16 --steps 5 --warmup 3 > gpurun_out/r2l/yolo.json 2> gpurun_out/r2l/yolo.err; tail -1 gpurun_out/r2l/yolo.json | cut -c1-200
HC_CONV_STAGED_STORES=0 timeout 600 python scripts/bench_yolov4.py --batch 16 --steps 5 --warmup 3 > gpurun_out/r2l/yolo_ns.json 2> gpurun_out/r2l/yolo_ns.err; tail -1 gpurun_out/r2l/yolo_ns.json | cut -c1-200
