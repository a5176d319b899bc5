#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_fullsize_layers.py -q -s -k "block_vs_oracles" ) > gpurun_out/r2a/fullsize_blocks.log 2>&1
grep -E "vs bf16|vs fp32|^(FAILED|ERROR)|passed|failed" gpurun_out/r2a/fullsize_blocks.log | tail -40
