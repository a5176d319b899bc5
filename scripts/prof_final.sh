#!/bin/bash
# Round-end measurements: headline bench (JSON line incl. roofline + cpu_baseline), its rocprofv3 kernel trace (per kernel and per
# launch geometry), and the secondary configurations.  Everything lands in gpurun_out/final/ (copy to profiles/).
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
for agg, suffix in ((per, "kernel_stats.txt"), (grid, "kernel_stats_by_grid.txt")):
    with open(out + "_" + suffix, "w") as fh:
        fh.write("# " + header + "\n")
        fh.write(f"{'kernel':<135} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
            fh.write(f"{k:<135} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>10.2f} {100.0 * t / tot:>6.2f}\n")
        fh.write(f"TOTAL kernel time {tot / 1e6:.3f} ms\n")
PY
}
# ---- headline
timeout 600 python $R/bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log | tail -1 > $O/r01_final_bench.json
timeout 500 rocprofv3 --kernel-trace --stats -d $O/t_head -o p --output-format csv -- python $R/bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline > $O/t_head.log 2>&1
summ $O/t_head $O/r01_final "rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline  (MI355X; eager steps incl. warm-up + the instrumented roofline step)"
grep '^{' $O/t_head.log | tail -1 > $O/r01_final_bench_under_rocprof.json
# ---- secondary configurations: bench line (graph replay where the script captures) + eager trace
sec() {  # sec <tag> <script> "<bench args>" "<trace args or empty>"
  timeout 400 python $R/scripts/$2 $3 > $O/$1.log 2>&1; grep '^{' $O/$1.log | tail -1 > $O/r01_$1_bench.json
  if [ -n "$4" ]; then
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/t_$1 -o p --output-format csv -- python $R/scripts/$2 $4 > $O/t_$1.log 2>&1
    summ $O/t_$1 $O/r01_$1 "rocprofv3 --kernel-trace --stats -- python scripts/$2 $4  (MI355X)"
  fi
}
sec yolov4 bench_yolov4.py "--batch 16 --steps 5 --warmup 3" "--batch 16 --steps 3 --warmup 1"
sec rexnet bench_rexnet.py "--steps 10 --warmup 3" "--steps 3 --warmup 1 --no-graph"
sec mobileone bench_mobileone.py "--steps 10 --warmup 3" "--steps 3 --warmup 1 --no-graph"
sec repvgg_a2_fp8 bench_repvgg_fp8.py "" ""
rm -rf $O/t_head $O/t_yolov4 $O/t_rexnet $O/t_mobileone
ls -la $O | head -40
