cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c11
timeout 900 python -m pytest tests/test_gpu_darknet.py tests/test_gpu_yolo.py tests/test_gpu_yolo_v1.py tests/test_gpu_convs.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c11/tests.log
timeout 300 python scripts/bench_yolov4.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c11/yolov4_bench.json 2> gpurun_out/c11/yolov4.err
cat gpurun_out/c11/tests.log; cut -c1-300 gpurun_out/c11/yolov4_bench.json; tail -3 gpurun_out/c11/yolov4.err
