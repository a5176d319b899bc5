#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r3l; mkdir -p $O; rm -rf $O/*
timeout 300 python -m pytest tests/test_gpu_multi_copy.py tests/test_gpu_zz_dp_graph.py -q -x 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
HC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_dist.json 2> $O/bench_dist.err
for f in bench bench_dist; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
(cd /tmp; export TMPDIR=/tmp; HC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o p --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 1 > $O/t.log 2>&1)
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r3l"
f=glob.glob(O+"/t/**/*kernel_trace.csv", recursive=True)[0]
per=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(f)):
    t=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); n=r["Kernel_Name"][:90]
    per[n][0]+=1; per[n][1]+=t
with open(O+"/dist_kernels.txt","w") as fh:
    for k,(n,t) in sorted(per.items(), key=lambda kv:-kv[1][1])[:70]:
        fh.write(f"{k:<92} {n:>6} {t/1e6:>9.3f} ms {t/n/1e3:>9.1f} us\n")
PY
rm -rf $O/t
grep -i "multi_copy\|foreach\|copy\|mul" $O/dist_kernels.txt | head
timeout 300 python scripts/bench_rexnet.py --steps 20 --warmup 5 > $O/r03_final_rexnet_bench.json 2> $O/rexnet.err; cut -c1-200 $O/r03_final_rexnet_bench.json
