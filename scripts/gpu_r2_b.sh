#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_wgrad_rep.py -q -x ) > gpurun_out/r2b/wrep.log 2>&1
tail -25 gpurun_out/r2b/wrep.log
( time timeout 1500 python -m pytest tests/test_gpu_fullsize_layers.py -q -s -k "block_vs" ) > gpurun_out/r2b/fullsize_c2.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/r2b/fullsize_c2.log | cut -c1-300 | tail -30
