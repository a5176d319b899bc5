cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c21
timeout 1200 python -m pytest tests/test_gpu_fullsize_layers.py -q -m gpu -k "other_repvgg" 2>&1 | tail -12 > gpurun_out/c21/tests.log
timeout 300 python bench.py --arch repvgg_a2 --no-cpu-baseline --steps 50 --profile-steps 2 > gpurun_out/c21/a2.json 2> gpurun_out/c21/a2.err
timeout 300 python bench.py --arch repvgg_a1 --no-cpu-baseline --steps 50 --profile-steps 2 > gpurun_out/c21/a1.json 2> gpurun_out/c21/a1.err
cat gpurun_out/c21/tests.log; cut -c1-400 gpurun_out/c21/a2.json; tail -3 gpurun_out/c21/a2.err; cut -c1-300 gpurun_out/c21/a1.json
