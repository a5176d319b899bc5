#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/q; O=gpurun_out/q; rm -f $O/*
timeout 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_repvgg.py tests/test_gpu_graph.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for pipe in 1 0 1 0; do HC_CONV_SMALL_PIPE=$pipe timeout 200 python bench.py --no-cpu-baseline > $O/b_$pipe.json 2> $O/b_$pipe.err; echo "pipe=$pipe $(cut -c90-125 $O/b_$pipe.json)"; done
