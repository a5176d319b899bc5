#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wgrad_rep.py -q -x 2>&1 | tail -2
echo "== HV=2 default"; python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== HV=1"; HC_WREP_HV=1 python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -4
echo "== HV=2 compute only"; HC_WREP_DBG=2 python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -4
echo "== HV=2 dma only"; HC_WREP_DBG=1 python scripts/bench_wrep.py --no-old 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -4
