cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c9
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/c9/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c9/smoke.log 2>&1
cat gpurun_out/c9/tests.log; tail -2 gpurun_out/c9/smoke.log
