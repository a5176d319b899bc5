#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
for v in 48 -1; do
  HC_EW_NT_MB=$v timeout 600 python scripts/bench_yolov4.py --batch 16 --steps 5 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
  HC_EW_NT_MB=$v timeout 600 python scripts/bench_rexnet.py --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
done
( time timeout 1500 python -m pytest tests/test_gpu_darknet.py tests/test_gpu_yolo.py tests/test_gpu_rexnet.py -q -x ) > gpurun_out/r2m/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r2m/pytest.log
