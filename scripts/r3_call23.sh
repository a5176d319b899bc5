#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3w; mkdir -p $O; rm -rf $O/*
HC_CONV_PIPE=1 timeout 500 python -m pytest tests/test_gpu_fullsize_layers.py tests/test_gpu_darknet.py -q -x -k "c4 or darknet" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests.log; cat $O/tests.log
for v in 0 1 0 1; do
HC_CONV_PIPE=$v timeout 300 python scripts/bench_yolov4.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/y$v.json 2> $O/y$v.err
python - $O/y$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f=d["roofline"]["families"]["conv_gather"]
print("yolov4 PIPE="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"],1), "conv_gather ms", round(f["ms_per_step"],3), "TF", round(f["tflops"]))
PY
done
for v in 0 1; do
HC_CONV_PIPE=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/b$v.json 2> $O/b$v.err
python - $O/b$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("repvgg PIPE="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"]))
PY
done
