#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_boundary.py -q -x ) > gpurun_out/r2e/boundary.log 2>&1
tail -30 gpurun_out/r2e/boundary.log
( timeout 900 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_fullsize_layers.py --deselect tests/test_gpu_boundary.py ) 2>&1 | tail -4
