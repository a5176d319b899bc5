#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/csm; O=gpurun_out/csm
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -k "conv_small" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
HC_CONV_SMALL_PIPE=1 timeout 100 python scripts/bench_layers.py small 0 2 2>&1 | grep "small-fwd(3x3\|small-dgrad\|no stats   " | tee $O/pipe.txt
