#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py tests/test_gpu_conv_rows.py tests/test_gpu_wgrad_rep.py -q -x ) > gpurun_out/r2o/repvgg.log 2>&1
grep -E "passed|failed" gpurun_out/r2o/repvgg.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2_block" ) > gpurun_out/r2o/layers.log 2>&1
grep -E "passed|failed" gpurun_out/r2o/layers.log
for m in 1 2 0 1; do HC_CONV_ROWS48=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows48=$m', round(d['value']), round(d['ms_per_step'],3))"; done
