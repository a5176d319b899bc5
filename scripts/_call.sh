cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7
timeout 900 python -m pytest tests/test_gpu_rexnet.py tests/test_gpu_convs.py tests/test_gpu_conv.py tests/test_gpu_fullsize_layers.py -x -q -m gpu -k "not c2_" 2>&1 | tail -8 > gpurun_out/c7/tests.log
timeout 300 python scripts/bench_rexnet.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/rexnet_bench.json 2> gpurun_out/c7/rexnet.err
MODEL=rexnet1_0x SHAPES=0 timeout 300 python scripts/prof_small_ops.py > gpurun_out/c7/rexnet_ops.txt 2>&1
cat gpurun_out/c7/tests.log; cut -c1-300 gpurun_out/c7/rexnet_bench.json; tail -3 gpurun_out/c7/rexnet.err; grep -v "Warn\|amdgpu\|^\[W" gpurun_out/c7/rexnet_ops.txt | cut -c1-180 | head -40
