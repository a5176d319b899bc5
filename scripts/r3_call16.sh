#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3p; mkdir -p $O; rm -rf $O/*
for v in 1 2; do
HC_CONV_BIG=$v timeout 400 python -m pytest tests/test_gpu_fullsize_layers.py -q -x -k "c2 and 1280" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests$v.log; echo "BIG=$v"; cat $O/tests$v.log
done
cd /tmp; export TMPDIR=/tmp
for v in 0 1 2; do
HC_CONV_BIG=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum -d $O/p$v -o m --output-format csv -- python $R/scripts/bench_block1280.py > $O/run$v.log 2>&1
python - $O/p$v $v <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        d = rows[(r["Dispatch_Id"], r["Kernel_Name"], r.get("Grid_Size", ""))]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("BIG=" + v)
n = 0
for (did, name, grid), d in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    if "conv_gather" not in name: continue
    n += 1
    if n <= 6: continue      # first iteration = warm-up
    if n > 12: break
    act = d.get("GRBM_GUI_ACTIVE", 0.0) / 8
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print(f"  {name[30:78]:48s} {d['ns']/1e3:8.1f} us  mfma_util {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(act*1024+1e-9):5.2f}  ta_busy {d.get('TA_BUSY_avr',0)/(act+1e-9):5.2f}  l2 hit {h/(h+m+1e-9):5.3f} req {h+m:.3g}")
PY
rm -rf $O/p$v
done 2>&1 | tee $O/summary.txt
