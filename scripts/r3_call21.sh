#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3u; mkdir -p $O; rm -rf $O/*
echo skip tests
for v in -1 0 -1 0; do
HC_WDMA_NSPLIT=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench$v.json 2> $O/bench$v.err
python - $O/bench$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f=d["roofline"]["families"]["conv_wgrad"]
print("NSPLIT="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"]), "conv_wgrad ms", round(f["ms_per_step"],3), "TF", round(f["tflops"]))
PY
done
