#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3m; mkdir -p $O; rm -rf $O/*
timeout 300 python scripts/find_small_kernels.py 2>&1 | grep -v amdgpu > $O/small.txt; head -60 $O/small.txt
