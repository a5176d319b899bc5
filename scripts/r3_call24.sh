#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3x; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > $O/tests.log; tail -5 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
