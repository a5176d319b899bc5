#!/bin/bash
# The 1280-channel gather-conv layers (scripts/bench_block1280.py) under rocprofv3: per launch duration, matrix-pipe busy fraction, wave
# wait split, for the 128 x 128 tile (HC_CONV_BIG=0), the four-wave (=2) and the eight-wave (=1, default) 256 x 256 tile.  One PMC pass
# per variant, --kernel-trace only.  Prints to stdout and gpurun_out/pmc_block1280/summary.txt.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_block1280; mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
for v in 0 2 1; do
HC_CONV_BIG=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/p$v -o m --output-format csv -- python $R/scripts/bench_block1280.py > $O/run$v.log 2>&1
python - $O/p$v $v <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        d = rows[(r["Dispatch_Id"], r["Kernel_Name"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("HC_CONV_BIG=" + v)
n = 0
for (did, name), d in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    if "conv_gather" not in name: continue
    n += 1
    if n <= 6 or n > 12: continue          # second of the four iterations (the first is warm-up)
    act = d.get("GRBM_GUI_ACTIVE", 0.0) / 8          # summed over the 8 XCDs by rocprofv3 on gfx950
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    cfg = name[name.index("<"):name.index(">") + 1] if "<" in name else name[:40]
    print(f"  {cfg:34s} {d['ns']/1e3:8.1f} us  eff_GHz {act/d['ns']:.2f}  mfma_busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(act*1024+1e-9):5.2f}  "
          f"wait_any {d.get('SQ_WAIT_ANY',0)/wc:5.2f}  wait_inst {d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f}  active_inst {d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}")
PY
rm -rf $O/p$v
done 2>&1 | tee $O/summary.txt
