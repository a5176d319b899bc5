#!/bin/bash
# Headline bench under rocprofv3 --kernel-trace --stats (eager launches so that every dispatch is a trace record): per-kernel and
# per-launch-geometry summaries into gpurun_out/prof_head/<prefix>_*.  Usage: prof_head.sh <prefix, e.g. r02_mid>
P=${1:-r02}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_head; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/t_head -o p --output-format csv -- python $R/bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline > $O/t_head.log 2>&1
python - "$O/t_head" "$O/$P" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline  (MI355X; eager steps incl. warm-up + the instrumented roofline step)" <<'PY'
import csv, glob, sys, collections
d, out, header = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
per, grid, tot = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0]), 0
for r in csv.DictReader(open(f)):
    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = (name[:name.index("(")] if "(" in name else name)[:110]
    per[name][0] += 1; per[name][1] += t
    k = name + f"  grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']})"
    grid[k][0] += 1; grid[k][1] += t
    tot += t
for agg, suffix, top in ((per, "kernel_stats.txt", 60), (grid, "kernel_stats_by_grid.txt", 90)):
    with open(out + "_" + suffix, "w") as fh:
        fh.write("# " + header + "\n")
        fh.write(f"{'kernel':<135} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            fh.write(f"{k:<135} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>10.2f} {100.0 * t / tot:>6.2f}\n")
        fh.write(f"TOTAL kernel time {tot / 1e6:.3f} ms\n")
PY
grep '^{' $O/t_head.log | tail -1 > $O/${P}_bench_under_rocprof.json
rm -rf $O/t_head
