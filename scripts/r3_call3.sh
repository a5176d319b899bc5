#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3c; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_conv_s2.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
HC_CONV_S2_V=1 timeout 600 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_fullsize_layers.py -k "conv_s2_forward or (c2_conv_passes_vs and s2)" -x -q > $O/tests_v1.log 2>&1; echo "tests_v1 rc=$?" >> $O/tests_v1.log; grep -E "passed|failed|rc=" $O/tests_v1.log | tail -3
for cfg in "" "HC_CONV_S2_V=1" "HC_CONV_S2_V=1 HC_CONV_S2_R=1" "HC_CONV_S2_DBG=1" "HC_CONV_S2_DBG=2" "HC_CONV_S2_DBG=4" "HC_CONV_S2_DBG=3" "HC_CONV_S2_V=1 HC_CONV_S2_DBG=1"; do
  echo "== $cfg"; env $cfg timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu
done > $O/s2_bench.txt 2>&1; cat $O/s2_bench.txt
HC_CONV_S2=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_off.json 2> $O/bench_off.err; echo "off: $(cut -c90-135 $O/bench_off.json)"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_on.json 2> $O/bench_on.err; echo "on: $(cut -c90-135 $O/bench_on.json)"
HC_CONV_S2_V=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_v1.json 2> $O/bench_v1.err; echo "v1: $(cut -c90-135 $O/bench_v1.json)"
