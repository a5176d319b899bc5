cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c16
timeout 1200 python -m pytest tests/test_gpu_rexnet.py tests/test_gpu_mobileone.py tests/test_gpu_convs.py tests/test_gpu_fullsize_bn.py tests/test_gpu_boundary.py tests/test_gpu_yolo.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c16/tests.log
timeout 300 python scripts/bench_rexnet.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c16/rexnet.json 2>/dev/null
timeout 300 python scripts/bench_mobileone.py 2>/dev/null | cut -c1-250 > gpurun_out/c16/mobileone.txt
HC_PAD_WIDE_FROM=0 timeout 300 python scripts/bench_mobileone.py 2>/dev/null | cut -c1-250 >> gpurun_out/c16/mobileone.txt
cat gpurun_out/c16/tests.log; cut -c1-250 gpurun_out/c16/rexnet.json; cat gpurun_out/c16/mobileone.txt
