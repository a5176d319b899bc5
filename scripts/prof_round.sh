#!/bin/bash
# Round-end measurements (prefix $1, default r02_final): headline bench (JSON line incl. roofline + cpu_baseline), its rocprofv3
# kernel trace (per kernel and per launch geometry), the forced multi-GPU code path, the secondary configurations, the per-shape
# tables of the two new kernels.  Everything lands in gpurun_out/final/ (copy to profiles/).
P=${1:-r02_final}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; rm -rf $O/*
summ() {  # summ <trace dir> <out prefix> <header>
python - "$1" "$2" "$3" <<'PY'
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
}
# ---- headline
timeout 900 python $R/bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log | tail -1 > $O/${P}_bench.json
timeout 500 rocprofv3 --kernel-trace --stats -d $O/t_head -o p --output-format csv -- python $R/bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline > $O/t_head.log 2>&1
summ $O/t_head $O/$P "rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline  (MI355X; eager steps incl. warm-up + the instrumented roofline step)"
grep '^{' $O/t_head.log | tail -1 > $O/${P}_bench_under_rocprof.json
HC_FORCE_DIST=1 timeout 300 python $R/bench.py --no-cpu-baseline > $O/dist.log 2>&1; grep '^{' $O/dist.log | tail -1 > $O/${P}_bench_forced_dist.json
# ---- secondary configurations: bench line + eager trace
sec() {  # sec <tag> <script> "<bench args>" "<trace args or empty>"
  timeout 600 python $R/scripts/$2 $3 > $O/$1.log 2>&1; grep '^{' $O/$1.log | tail -1 > $O/${P}_$1_bench.json
  if [ -n "$4" ]; then
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/t_$1 -o p --output-format csv -- python $R/scripts/$2 $4 > $O/t_$1.log 2>&1
    summ $O/t_$1 $O/${P}_$1 "rocprofv3 --kernel-trace --stats -- python scripts/$2 $4  (MI355X)"
  fi
}
sec yolov4 bench_yolov4.py "--batch 16 --steps 5 --warmup 3" "--batch 16 --steps 3 --warmup 1 --no-cpu-baseline"
sec rexnet bench_rexnet.py "--steps 10 --warmup 3" "--steps 3 --warmup 1 --no-graph --no-cpu-baseline"
# ---- per-shape tables
( cd $R; timeout 300 python scripts/bench_wrep.py --no-old > $O/${P}_wgrad_rep_shapes.txt 2>&1; timeout 200 python scripts/check_rows.py > $O/${P}_conv_rows_shapes.txt 2>&1; timeout 200 python scripts/bench_ew.py 2>&1 | grep "C=" > $O/${P}_bn_passes.txt )
rm -rf $O/t_head $O/t_yolov4 $O/t_rexnet
ls -la $O | head -40
