#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3k; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_fullsize_bn.py -k "graph_with_packed or bn_passes" -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
HC_EW_ROUNDS=0 timeout 120 python scripts/bench_ew.py 2>&1 | grep -v amdgpu > $O/ew0.txt; HC_EW_ROUNDS=1 timeout 120 python scripts/bench_ew.py 2>&1 | grep -v amdgpu > $O/ew1.txt; paste -d'\n' $O/ew0.txt $O/ew1.txt | cut -c1-330
for cfg in "HC_EW_ROUNDS=0" "HC_EW_ROUNDS=1" "HC_STACK_MAXC=96"; do env $cfg timeout 300 python bench.py --steps 150 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg: $(cut -c90-135 $O/bench_$cfg.json)"; done
