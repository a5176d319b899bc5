cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --profile-steps 1 --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), d['config']['mode'][:150])"; }
run single A=1
run dist HC_FORCE_DIST=1
run dist_cut40 HC_FORCE_DIST=1 HC_BENCH_CUT0=4:0
run dist_cut38 HC_FORCE_DIST=1 HC_BENCH_CUT0=3:8
run dist_cut311 HC_FORCE_DIST=1 HC_BENCH_CUT0=3:11
run dist_cut38_2seg HC_FORCE_DIST=1 HC_BENCH_CUT0=3:8 HC_BENCH_CUTS=1
run dist_2seg HC_FORCE_DIST=1 HC_BENCH_CUTS=1
