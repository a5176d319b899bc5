#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3b; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_fullsize_bn.py -k "s2 or fp8_stem" -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
HC_CONV_S2=0 timeout 200 python scripts/bench_s2.py > $O/s2_off.txt 2>&1; cat $O/s2_off.txt | grep -v amdgpu
timeout 200 python scripts/bench_s2.py > $O/s2_on.txt 2>&1; cat $O/s2_on.txt | grep -v amdgpu
HC_CONV_S2_R=1 timeout 200 python scripts/bench_s2.py > $O/s2_on_r1.txt 2>&1; cat $O/s2_on_r1.txt | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_fullsize_layers.py tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py -k "c2_conv_passes_vs or c2_block or repvgg or fullsize" -x -q > $O/tests2.log 2>&1; echo "tests2 rc=$?" >> $O/tests2.log; grep -E "passed|failed|rc=" $O/tests2.log | tail -3
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
