#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3e; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_fullsize_layers.py tests/test_gpu_repvgg.py -k "s2 or 3@224 or repvgg" -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu > $O/s2_bench.txt; tail -4 $O/s2_bench.txt
HC_CONV_S2_STEM_WGRAD=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_off.json 2> $O/bench_off.err; echo "off: $(cut -c90-135 $O/bench_off.json)"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_on.json 2> $O/bench_on.err; echo "on: $(cut -c90-135 $O/bench_on.json)"
