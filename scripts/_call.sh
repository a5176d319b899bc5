cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c3/tests.log
timeout 400 python scripts/bench_yolov4.py --eval --batch 16 --steps 10 --warmup 3 > gpurun_out/c3/yolov4_eval.json 2> gpurun_out/c3/yolov4_eval.err
timeout 400 python scripts/bench_repvgg_fp8.py > gpurun_out/c3/fp8.json 2> gpurun_out/c3/fp8.err
cat gpurun_out/c3/tests.log; cat gpurun_out/c3/yolov4_eval.json; tail -3 gpurun_out/c3/yolov4_eval.err; cut -c1-1500 gpurun_out/c3/fp8.json; tail -3 gpurun_out/c3/fp8.err
