#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3d; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_conv_rows.py tests/test_gpu_fullsize_layers.py tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py -k "not c4_ and not c3_" -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
for cfg in "" "HC_CONV_S2_R=1" "HC_CONV_S2_DBG=1" "HC_CONV_S2_DBG=4"; do
  echo "== $cfg"; env $cfg timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu
done > $O/s2_bench.txt 2>&1; cat $O/s2_bench.txt
timeout 100 python scripts/check_rows.py > $O/rows.txt 2>&1; grep -v amdgpu $O/rows.txt | tail -4
HC_CONV_S2=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_off.json 2> $O/bench_off.err; echo "off: $(cut -c90-135 $O/bench_off.json)"
HC_CONV_S2_DGRAD=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_nodg.json 2> $O/bench_nodg.err; echo "nodgrad: $(cut -c90-135 $O/bench_nodg.json)"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_on.json 2> $O/bench_on.err; echo "on: $(cut -c90-135 $O/bench_on.json)"
