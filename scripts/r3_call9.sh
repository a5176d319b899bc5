#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3i; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_fullsize_bn.py tests/test_gpu_repvgg.py tests/test_gpu_bn_zmask.py tests/test_gpu_darknet.py tests/test_gpu_rexnet.py -k "graph_with_packed or bn_passes or bn_act or repvgg or zmask or darknet or rexnet" -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
HC_CONV_TILE256=2 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_convs.py tests/test_gpu_fullsize_layers.py -k "not c3_ and not deterministic and not block_vs" -x -q > $O/tests_t256.log 2>&1; echo "tests_t256 rc=$?" >> $O/tests_t256.log; grep -E "passed|failed|rc=" $O/tests_t256.log | tail -3
for m in 0 1; do echo "== HC_CONV_TILE256=$m"; HC_CONV_TILE256=$m timeout 120 python scripts/bench_layers.py fwd 7 8 2>&1 | grep -v amdgpu; HC_CONV_TILE256=$m timeout 120 python scripts/bench_layers.py dgrad 7 8 2>&1 | grep -v amdgpu; done > $O/layers.txt 2>&1; cat $O/layers.txt
HC_CONV_TILE256=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_t0.json 2> $O/bench_t0.err; echo "t0: $(cut -c90-135 $O/bench_t0.json)"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_t1.json 2> $O/bench_t1.err; echo "t1: $(cut -c90-135 $O/bench_t1.json)"
HC_CONV_TILE256=0 timeout 300 python scripts/bench_yolov4.py --batch 16 --steps 8 --warmup 3 --no-cpu-baseline > $O/yolo_t0.json 2>/dev/null; echo "yolo t0: $(cut -c90-140 $O/yolo_t0.json)"
HC_CONV_TILE256=1 timeout 300 python scripts/bench_yolov4.py --batch 16 --steps 8 --warmup 3 --no-cpu-baseline > $O/yolo_t1.json 2>/dev/null; echo "yolo t1: $(cut -c90-140 $O/yolo_t1.json)"
