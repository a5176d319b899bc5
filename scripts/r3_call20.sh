#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3t; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_fullsize_layers.py tests/test_gpu_fullsize.py tests/test_gpu_repvgg.py -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/tests.log; cat $O/tests.log
for v in 0 1 0 1; do
HC_CONV_BIG=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench$v.json 2> $O/bench$v.err
python - $O/bench$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f=d["roofline"]["families"]["conv_gather"]
print("BIG="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"]), "conv_gather ms", round(f["ms_per_step"],3), "TF", round(f["tflops"]))
PY
done
