#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3y; mkdir -p $O; rm -rf $O/*
timeout 300 python -m pytest tests/test_gpu_fullsize_bn.py -q -x -s -k "c2_repblock" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -16 > $O/tests_bn.log; cat $O/tests_bn.log | cut -c1-330
timeout 400 python -m pytest tests/test_gpu_repvgg.py tests/test_gpu_fullsize.py tests/test_gpu_bn_zmask.py -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -4 > $O/tests2.log; cat $O/tests2.log
for v in 0 1 0 1; do
HC_BN_FUSED_BWD=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench$v.json 2> $O/bench$v.err
python - $O/bench$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f=d["roofline"]["families"]
print("FUSED="+sys.argv[2], "ms/step", round(d["ms_per_step"],3), "img/s", round(d["value"]), "bn_elementwise ms", round(f["bn_elementwise"]["ms_per_step"],3), "bn_finalize ms", round(f.get("bn_finalize",{}).get("ms_per_step",0),3), "loss", d["config"].get("final_loss"))
PY
done
