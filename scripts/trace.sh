#!/bin/bash
# rocprofv3 kernel trace of one command, summarised per kernel and per (kernel, grid):  bash scripts/trace.sh <out prefix> "<header>" <command ...>
# writes <out prefix>_kernel_stats.txt and <out prefix>_kernel_stats_by_grid.txt (the raw trace is deleted)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
out=$1; hdr=$2; shift 2
T=$R/gpurun_out/_trace_$$; mkdir -p $T $(dirname $out)
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $T -o p --output-format csv -- "$@" > $T.log 2>&1)
python - "$T" "$out" "$hdr" <<'PY'
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
for agg, suffix in ((per, "kernel_stats.txt"), (grid, "kernel_stats_by_grid.txt")):
    with open(out + "_" + suffix, "w") as fh:
        fh.write("# " + header + "\n")
        fh.write(f"{'kernel':<135} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
            fh.write(f"{k:<135} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>10.2f} {100.0 * t / tot:>6.2f}\n")
        fh.write(f"TOTAL kernel time {tot / 1e6:.3f} ms\n")
PY
tail -5 $T.log; rm -rf $T $T.log
