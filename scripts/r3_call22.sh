#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3v; mkdir -p $O; rm -rf $O/*
for v in 0 1 0 1; do
HC_WREP_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench$v.json 2> $O/bench$v.err
python - $O/bench$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("WREP_SIDE="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"]))
PY
done
