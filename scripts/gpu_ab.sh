#!/bin/bash
# One GPU-box call: targeted tests of the new paths, then same-box A/B of the bench modes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
O=gpurun_out/ab
rm -f $O/*
timeout 400 python -m pytest tests/test_gpu_graph.py tests/test_gpu_zz_dp_graph.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
run() { name=$1; shift; ( "$@" timeout 240 python bench.py --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err ); echo "$name rc=$? $(cut -c90-130 $O/$name.json) lines=$(wc -l < $O/$name.json)"; }
for i in 1 2 3; do run dist$i env HC_FORCE_DIST=1; done
EXTRA=--no-graph run dist_eager env HC_FORCE_DIST=1
run base env
tail -3 $O/tests.log
