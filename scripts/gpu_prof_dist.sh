#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pd; mkdir -p $O; rm -rf $O/*
HC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o p --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/run.log 2>&1
python - $O <<'PY'
import csv, glob, sys, collections
O=sys.argv[1]
f = glob.glob(O + "/t/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(lambda: [0, 0]); tot=0
for r in csv.DictReader(open(f)):
    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")[:150]
    per[name][0]+=1; per[name][1]+=t; tot+=t
with open(O+"/dist_kernels.txt","w") as fh:
    for k,(n,t) in sorted(per.items(), key=lambda kv:-kv[1][1]):
        if any(s in k.lower() for s in ("foreach","multi_tensor","copy","ccl","elementwise","adabelief","fill")):
            fh.write(f"{k:<150} {n:>6} {t/1e6:>9.3f} ms {t/n/1e3:>9.2f} us\n")
    fh.write(f"TOTAL {tot/1e6:.3f} ms\n")
PY
rm -rf $O/t; tail -3 $O/run.log | cut -c1-300
