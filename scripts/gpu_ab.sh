#!/bin/bash
# One GPU-box call: targeted tests of the new paths, then same-box A/B of the bench modes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
O=gpurun_out/ab
rm -f $O/*
timeout 400 python -m pytest tests/test_gpu_graph.py tests/test_gpu_zz_dp_graph.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
run() { name=$1; shift; ( "$@" timeout 240 python bench.py --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err ); echo "$name rc=$? $(cut -c90-130 $O/$name.json) lines=$(wc -l < $O/$name.json)"; }
for i in 1 2 3 4; do run dist$i env HC_FORCE_DIST=1 HC_WGRAD_STREAM=0; done
for i in 1 2 3; do run dist_side$i env HC_FORCE_DIST=1 HC_WGRAD_STREAM=1; done
EXTRA=--no-graph run dist_eager env HC_FORCE_DIST=1
run base env HC_WGRAD_STREAM=0
run side env HC_WGRAD_STREAM=1
( timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > $O/torchrun.json 2> $O/torchrun.err ); echo "torchrun rc=$? $(cut -c90-130 $O/torchrun.json) lines=$(wc -l < $O/torchrun.json)"
tail -3 $O/tests.log
