cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_pointwise.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/c2/tests.log
HC_TORCH_LOSS=1 timeout 300 python bench.py --no-cpu-baseline --profile-steps 2 > gpurun_out/c2/bench_torchloss.json 2> gpurun_out/c2/bench.err
timeout 300 python bench.py --no-cpu-baseline --profile-steps 2 > gpurun_out/c2/bench.json 2>> gpurun_out/c2/bench.err
HC_LOSS=1 timeout 300 python scripts/prof_small_ops.py > gpurun_out/c2/small_ops.txt 2>&1
cat gpurun_out/c2/tests.log; cut -c1-330 gpurun_out/c2/bench_torchloss.json;  cut -c1-330 gpurun_out/c2/bench.json; grep -v "^\[W\|Warn\|amdgpu" gpurun_out/c2/small_ops.txt | cut -c1-220 | head -70
