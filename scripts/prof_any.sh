#!/bin/bash
# rocprofv3 kernel-trace of an arbitrary bench script: prof_any.sh <tag> <script> [args...]; writes gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag; mkdir -p $out; rm -rf $out/*
timeout 500 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/$@ > $out/run.log 2>&1
grep '^{' $out/run.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':<120} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f"{r['Name'][:120]:<120} {int(r['Calls']):>7} {float(r['TotalDurationNs'])/1e6:>10.3f} {float(r['AverageNs'])/1e3:>10.2f} {100*float(r['TotalDurationNs'])/tot:>6.2f}")
print("total_ms", tot/1e6)
PY
