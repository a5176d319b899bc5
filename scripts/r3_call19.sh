#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3s; mkdir -p $O; rm -rf $O/*
HC_CONV_BIG=4 HC_CONV_BIG_MINS=16 timeout 400 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2 and 1280" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests4.log; echo "BIG=4"; cat $O/tests4.log
HC_CONV_BIG=3 HC_CONV_BIG_MINS=16 timeout 400 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2 and 1280" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests3.log; echo "BIG=3 mins16"; cat $O/tests3.log
cd /tmp; export TMPDIR=/tmp
for v in 0 3 4; do
HC_CONV_BIG=$v HC_CONV_BIG_MINS=16 timeout 300 rocprofv3 --kernel-trace -d $O/p$v -o m --output-format csv -- python $R/scripts/bench_block1280.py > $O/run$v.log 2>&1
python - $O/p$v $v <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_gather" in r["Kernel_Name"]]
print("BIG=" + v, " ".join(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f}" for r in rows[12:24]))
PY
rm -rf $O/p$v
done 2>&1 | tee $O/summary.txt
