#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab2; mkdir -p $O; rm -f $O/*
( HC_WGRAD_SIDE_STREAM=1 timeout 500 python -m pytest tests/test_gpu_darknet.py tests/test_gpu_yolo.py tests/test_gpu_rexnet.py tests/test_gpu_repvgg.py tests/test_gpu_graph.py -x -q > $O/tests_side.log 2>&1; echo "tests rc=$?" >> $O/tests_side.log ) 
tail -3 $O/tests_side.log
for v in 0 1 0 1; do
  HC_WGRAD_SIDE_STREAM=$v timeout 200 python scripts/bench_yolov4.py --batch 16 --steps 6 --warmup 3 > $O/yolo_$v.json 2>> $O/yolo.err; echo "yolo side=$v $(cut -c1-200 $O/yolo_$v.json | tail -1)"
  HC_WGRAD_SIDE_STREAM=$v timeout 200 python scripts/bench_rexnet.py > $O/rex_$v.json 2>> $O/rex.err; echo "rex side=$v $(cut -c1-200 $O/rex_$v.json | tail -1)"
done
