#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
timeout 900 python scripts/bench_yolov4.py --steps 5 --warmup 2 > gpurun_out/r2f/yolov4.json 2> gpurun_out/r2f/yolov4.err; echo "yolov4 rc=$?"; cut -c1-1800 gpurun_out/r2f/yolov4.json; tail -3 gpurun_out/r2f/yolov4.err
HC_FORCE_DIST=1 timeout 600 python scripts/bench_yolov4.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/yolov4_dist.json 2> gpurun_out/r2f/yolov4_dist.err; echo "yolov4 dist rc=$?"; cut -c1-700 gpurun_out/r2f/yolov4_dist.json; tail -3 gpurun_out/r2f/yolov4_dist.err
