#!/bin/bash
# Final verification of a round: the whole GPU suite, smoke(), the headline bench (plain and the forced multi-GPU code path).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/fin; O=gpurun_out/fin; rm -f $O/*
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 100 python scripts/bench_layers.py small 0 2 2>&1 | grep "small-fwd(3x3\|small-dgrad"
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(cut -c90-130 $O/bench.json)"
HC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_dist.json 2> $O/bench_dist.err; echo "dist rc=$? $(cut -c90-130 $O/bench_dist.json)"
