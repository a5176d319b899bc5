cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c14
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_repvgg.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c14/tests.log
timeout 300 python bench.py --no-cpu-baseline --profile-steps 1 --steps 150 > gpurun_out/c14/bench.json 2> gpurun_out/c14/bench.err
HC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c14/bench_dist.json 2>> gpurun_out/c14/bench.err
cat gpurun_out/c14/tests.log; cut -c1-300 gpurun_out/c14/bench.json; cut -c1-300 gpurun_out/c14/bench_dist.json
