#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wgrad_rep.py -q -x 2>&1 | tail -3
for dbg in 0 1 2; do echo "== HC_WREP_DBG=$dbg"; HC_WREP_DBG=$dbg python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
