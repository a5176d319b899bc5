#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3g; mkdir -p $O; rm -rf $O/*
HC_CONV_S2_V=2 timeout 600 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_fullsize_layers.py -k "conv_s2_forward or (c2_conv_passes_vs and s2)" -x -q > $O/tests_v2.log 2>&1; echo "tests_v2 rc=$?" >> $O/tests_v2.log; grep -E "passed|failed|rc=" $O/tests_v2.log | tail -3
for cfg in "HC_CONV_S2_V=1" "HC_CONV_S2_V=2" "HC_CONV_S2_V=2 HC_CONV_S2_R=1" "HC_CONV_S2_V=2 HC_CONV_S2_DBG=1"; do
  echo "== $cfg"; env $cfg timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu | head -3
done > $O/s2_bench.txt 2>&1; cat $O/s2_bench.txt
HC_CONV_S2_V=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_v1.json 2> $O/bench_v1.err; echo "v1: $(cut -c90-135 $O/bench_v1.json)"
HC_CONV_S2_V=2 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_v2.json 2> $O/bench_v2.err; echo "v2: $(cut -c90-135 $O/bench_v2.json)"
