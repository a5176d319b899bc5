#!/bin/bash
# Round-6 end measurements.  Everything lands in gpurun_out/final6/ (copied to profiles/r06_*).  PMC passes first: the bench lines
# read the traffic files they produce (bench.py / scripts/_train_bench.py look for profiles/r06_pmc_*_traffic.json).
# SECTIONS="rexnet" (or any subset of: headline rexnet yolov4 fp8 mobileone misc) re-measures one configuration after a change to its kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/final6; mkdir -p $O
SECTIONS=${SECTIONS:-"headline rexnet yolov4 fp8 mobileone misc"}
want() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
[ "$SECTIONS" = "headline rexnet yolov4 fp8 mobileone misc" ] && rm -rf $O/*
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
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
            fh.write(f"{k:<135} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>10.2f} {100.0 * t / tot:>6.2f}\n")
        fh.write(f"TOTAL kernel time {tot / 1e6:.3f} ms\n")
PY
}
trace() {  # trace <tag> <header> <command ...>
  tag=$1; hdr=$2; shift 2
  (cd /tmp; export TMPDIR=/tmp; timeout 500 rocprofv3 --kernel-trace --stats -d $O/t_$tag -o p --output-format csv -- "$@" > $O/t_$tag.log 2>&1)
  summ $O/t_$tag $O/r06_final_$tag "$hdr"; rm -rf $O/t_$tag
}
# ---- headline: PMC traffic + MFMA utilisation, then the bench line, then the eager kernel trace
if want headline; then
bash scripts/pmc_families.sh step python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $O/pmc_step.log 2>&1
cp gpurun_out/pmc_step/traffic.json profiles/r06_pmc_step_traffic.json; cp gpurun_out/pmc_step/traffic.json $O/r06_pmc_step_traffic.json
bash scripts/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/mfma_util.txt $O/r06_mfma_util.txt
timeout 600 python bench.py > $O/r06_final_bench.json 2> $O/bench.err
HC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > $O/r06_final_bench_forced_dist.json 2> $O/bench_dist.err
# two eager kernel traces of the headline: SERIALISED (everything on one stream: per-kernel durations are the kernels' own) and
# CONCURRENT (bench.py's default: finished weight-gradient groups on a second stream - durations of kernels that ran beside them are
# inflated, the file says so in its header)
HC_WREP_SIDE=0 HC_WGRAD_STREAM=0 trace headline_serialised "SERIALISED: HC_WREP_SIDE=0 HC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline --profile-steps 1  (MI355X; one stream: per-kernel durations are undisturbed)" python $R/bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline --profile-steps 1
trace headline_concurrent "CONCURRENT (weight gradients on a second stream, bench.py's default): rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline --profile-steps 1  (MI355X; durations of kernels that overlap the side stream are inflated - use the serialised file for per-kernel numbers)" python $R/bench.py --steps 12 --warmup 3 --no-graph --no-cpu-baseline --profile-steps 1
timeout 200 python scripts/bench_stem.py 2>&1 | grep -v amdgpu > $O/r06_final_stem_fused.txt
( for e in "HC_WGRAD_KNOCKOUT=0" "HC_WGRAD_KNOCKOUT=1" "HC_WGRAD_STREAM=0 HC_WREP_SIDE=0" "HC_STEM_FUSED=0"; do env $e timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', round(d['ms_per_step'],3), 'ms/step')"; done ) > $O/r06_final_step_knockouts.txt
timeout 120 python scripts/bench_ew.py 2>&1 | grep -v amdgpu > $O/r06_final_bn_passes.txt
timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu > $O/r06_final_conv_s2_shapes.txt
timeout 120 python scripts/check_rows.py 2>&1 | grep -v amdgpu > $O/r06_final_conv_rows_shapes.txt
timeout 200 python scripts/bench_wrep.py 2>&1 | grep -v amdgpu > $O/r06_final_wgrad_rep_shapes.txt
fi
# ---- secondary configurations: PMC traffic, bench line (graph replay), eager trace
if want rexnet; then
bash scripts/pmc_families.sh rexnet python scripts/bench_rexnet.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc_rexnet.log 2>&1
cp gpurun_out/pmc_rexnet/traffic.json profiles/r06_pmc_rexnet_traffic.json; cp gpurun_out/pmc_rexnet/traffic.json $O/r06_pmc_rexnet_traffic.json
timeout 400 python scripts/bench_rexnet.py --steps 20 --warmup 5 > $O/r06_final_rexnet_bench.json 2> $O/rexnet.err
trace rexnet "rocprofv3 --kernel-trace --stats -- python scripts/bench_rexnet.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline  (MI355X)" python $R/scripts/bench_rexnet.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline
timeout 300 python scripts/bench_dw.py 2>&1 | grep -v amdgpu > $O/r06_final_dw_tile_shapes.txt
timeout 300 python scripts/layer_table.py --model rexnet1_0x 2>/dev/null > $O/r06_final_rexnet_layer_table.txt
( for e in "HC_CONV_SHORT=1024" "HC_CONV_SHORT=0" "HC_CONV_SHORT=1024" "HC_CONV_SHORT=0"; do env $e timeout 300 python scripts/bench_rexnet.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')"; done ) > $O/r06_rexnet_ab.txt
fi
if want yolov4; then
bash scripts/pmc_families.sh yolov4 python scripts/bench_yolov4.py --batch 16 --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc_yolov4.log 2>&1
cp gpurun_out/pmc_yolov4/traffic.json profiles/r06_pmc_yolov4_traffic.json; cp gpurun_out/pmc_yolov4/traffic.json $O/r06_pmc_yolov4_traffic.json
timeout 500 python scripts/bench_yolov4.py --batch 16 --steps 10 --warmup 3 > $O/r06_final_yolov4_bench.json 2> $O/yolov4.err
timeout 500 python scripts/bench_yolov4.py --eval --batch 16 --steps 10 --warmup 3 > $O/r06_final_yolov4_eval_bench.json 2> $O/yolov4_eval.err
timeout 300 python scripts/bench_bigtile.py 2>&1 | grep -v amdgpu > $O/r06_final_bigtile_family_shapes.txt
timeout 300 python scripts/layer_table.py --model yolov4 2>/dev/null > $O/r06_final_yolov4_layer_table.txt
HC_WGRAD_DEFER=0 timeout 300 python scripts/layer_table.py --model yolov4 2>/dev/null > $O/r06_yolov4_layer_table_ungrouped_wgrad.txt
( for e in "HC_WGRAD_DEFER=1" "HC_WGRAD_DEFER=0" "HC_WGRAD_DEFER=1" "HC_WGRAD_DEFER=0"; do env $e timeout 300 python scripts/bench_yolov4.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')"; done
  OLDD="HC_CONV_DEEP=0 HC_CONV_SHORT_FIRST=0 HC_CONV_SHORT_ALL=1 HC_CONV_C32=0 HC_CONV_CLSFAST=0 HC_CSP_SPLIT=0 HC_SPP_TILE=0"
  for e in "HC_X=1" "$OLDD" "HC_X=1" "$OLDD"; do env $e timeout 300 python scripts/bench_yolov4.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train [$e]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')"; done
  for e in "HC_X=1" "$OLDD" "HC_X=1" "$OLDD"; do env $e timeout 300 python scripts/bench_yolov4.py --eval --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('eval [$e]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/pass')"; done
  for e in "HC_INFER_FUSED=1" "HC_INFER_FUSED=0" "HC_INFER_FUSED=1" "HC_INFER_FUSED=0"; do env $e timeout 300 python scripts/bench_yolov4.py --eval --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('eval $e', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/pass')"; done ) > $O/r06_yolov4_ab.txt
bash scripts/trace.sh $O/r06_final_yolov4_eval "rocprofv3 --kernel-trace --stats -- python scripts/bench_yolov4.py --eval --steps 5 --warmup 2 --no-cpu-baseline  (MI355X; 8 eval passes of 16 images)" python $R/scripts/bench_yolov4.py --eval --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
trace yolov4 "rocprofv3 --kernel-trace --stats -- python scripts/bench_yolov4.py --batch 16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline  (MI355X)" python $R/scripts/bench_yolov4.py --batch 16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline
fi
if want fp8; then
bash scripts/pmc_families.sh repvgg_a2_fp8 python scripts/bench_repvgg_fp8.py --steps 2 --warmup 1 > $O/pmc_fp8.log 2>&1
cp gpurun_out/pmc_repvgg_a2_fp8/traffic.json profiles/r06_pmc_repvgg_a2_fp8_traffic.json; cp gpurun_out/pmc_repvgg_a2_fp8/traffic.json $O/r06_pmc_repvgg_a2_fp8_traffic.json
timeout 300 python scripts/bench_repvgg_fp8.py > $O/r06_final_repvgg_a2_fp8_bench.json 2> $O/fp8.err
fi
if want mobileone; then
bash scripts/pmc_families.sh mobileone python scripts/bench_mobileone.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc_mobileone.log 2>&1
cp gpurun_out/pmc_mobileone/traffic.json profiles/r06_pmc_mobileone_traffic.json; cp gpurun_out/pmc_mobileone/traffic.json $O/r06_pmc_mobileone_traffic.json
timeout 400 python scripts/bench_mobileone.py > $O/r06_final_mobileone_bench.json 2> $O/mobileone.err
fi
if want misc; then
timeout 200 ./scripts/probes/fill_probe.bin rows 2>&1 > $O/r06_fill_probe_rows.txt
timeout 200 python scripts/probe_whole_models.py 2>/dev/null > $O/r06_whole_model_probe.txt
ls -la $O | head -60
cut -c1-300 $O/r06_final_bench.json
timeout 200 python scripts/fixture_fracs.py 2>&1 | grep -v amdgpu > $O/fixture_fracs.txt; tail -1 $O/fixture_fracs.txt
timeout 400 python -m pytest tests/test_gpu_yolo.py -q -x 2>&1 | tail -5 > $O/yolo_tests.log; cat $O/yolo_tests.log
fi
