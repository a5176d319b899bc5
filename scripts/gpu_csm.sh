#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/csm; O=gpurun_out/csm
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -k "conv_small" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for pipe in 1 0; do echo "== pipe $pipe"; HC_CONV_SMALL_PIPE=$pipe timeout 100 python scripts/bench_layers.py small 0 2 2>&1 | grep "small-fwd(3x3\|small-dgrad\|no stats   "; done | tee $O/pipe.txt
bash scripts/pmc_traffic.sh small 0 2>&1 | tail -7
