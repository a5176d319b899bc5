#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3a; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_fullsize_bn.py tests/test_gpu_fullsize_layers.py -k "bn or deterministic or fp8 or se_scale or spp" -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
timeout 120 python scripts/bench_ew.py > $O/bench_ew.txt 2>&1
bash scripts/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/mfma_util.txt $O/ 2>/dev/null; head -40 $O/mfma_util.txt
