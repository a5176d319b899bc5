cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5
timeout 900 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_pointwise.py tests/test_gpu_boundary.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c5/tests.log
timeout 400 python scripts/bench_yolov4.py --eval --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c5/yolov4_eval.json 2> gpurun_out/c5/yolov4_eval.err
timeout 300 python bench.py --no-cpu-baseline --profile-steps 2 > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
cat gpurun_out/c5/tests.log; cat gpurun_out/c5/yolov4_eval.json; tail -3 gpurun_out/c5/yolov4_eval.err; cut -c1-330 gpurun_out/c5/bench.json
