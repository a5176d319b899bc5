#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3o; mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
HC_CONV_BIG=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $O/p$v -o m --output-format csv -- python $R/scripts/bench_block1280.py > $O/run$v.log 2>&1
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
for (did, name, grid), d in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    if "conv_gather" not in name: continue
    act = d.get("GRBM_GUI_ACTIVE", 0.0) / 8
    print(f"  {name[30:75]:45s} grid {grid:>8s} {d['ns']/1e3:8.1f} us  mfma_util {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(act*1024+1e-9):5.2f}  lds_conflict/active {d.get('SQ_LDS_BANK_CONFLICT',0)/(d.get('SQ_LDS_IDX_ACTIVE',1)+1e-9):5.2f}  lds_active/cycle {d.get('SQ_LDS_IDX_ACTIVE',0)/(act*256+1e-9):5.2f} wait_lds {d.get('SQ_WAIT_INST_LDS',0):.3g} insts_lds {d.get('SQ_INSTS_LDS',0):.3g}")
PY
rm -rf $O/p$v
done 2>&1 | tee $O/summary.txt
