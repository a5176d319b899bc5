cd $GRAFT_REPO_ROOT
runr() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_rexnet.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3))"; }
runy() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_yolov4.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3))"; }
runr rex_base A=1
runr rex_ew512 HC_EW_FLOOR=512
runr rex_ew2048 HC_EW_FLOOR=2048
runr rex_ew4096 HC_EW_FLOOR=4096
runr rex_nt16 HC_EW_NT_MB=16
runy yolo_base A=1
runy yolo_ew512 HC_EW_FLOOR=512
runy yolo_ew2048 HC_EW_FLOOR=2048
runy yolo_ew4096 HC_EW_FLOOR=4096
runy yolo_side HC_WGRAD_SIDE_STREAM=1
runr rex_side HC_WGRAD_SIDE_STREAM=1
