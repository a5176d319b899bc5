cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c18
runr() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_rexnet.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3))"; }
runh() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --profile-steps 1 --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3))"; }
runf() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_repvgg_fp8.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['bf16_ms'],3))"; }
runr rex_fill0 HC_CONV_FILL=0
runr rex_fill400 A=1
runr rex_fill0b HC_CONV_FILL=0
runr rex_fill400b A=1
runh head_fill0 HC_CONV_FILL=0
runh head_fill400 A=1
runf fp8_fill0 HC_CONV_FILL=0
runf fp8_fill400 A=1
