cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c20
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/c20/tests.log
timeout 300 python scripts/bench_mobileone.py --no-cpu-baseline > gpurun_out/c20/mobileone.json 2> gpurun_out/c20/mobileone.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c20/smoke.log 2>&1
cat gpurun_out/c20/tests.log; cut -c1-300 gpurun_out/c20/mobileone.json; tail -2 gpurun_out/c20/mobileone.err; tail -1 gpurun_out/c20/smoke.log
