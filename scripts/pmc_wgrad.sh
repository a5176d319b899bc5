#!/bin/bash
# PMC counters for one microbenchmarked layer (own run, no tracing flags besides kernel-trace)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc -o p1 --output-format csv -- python scripts/bench_layers.py "$@" > gpurun_out/pmc/run1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc -o p2 --output-format csv -- python scripts/bench_layers.py "$@" > gpurun_out/pmc/run2.log 2>&1
ls gpurun_out/pmc
tail -3 gpurun_out/pmc/run1.log
