#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3e; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_conv_s2.py tests/test_gpu_fullsize_layers.py tests/test_gpu_repvgg.py -k "s2 or 224 or repvgg" -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 200 python scripts/bench_s2.py 2>&1 | grep -v amdgpu > $O/s2_bench.txt; tail -4 $O/s2_bench.txt
cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_s2.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3e/prof/**/*kernel_trace.csv", recursive=True)
if f:
    per = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]; n = n[:n.index("(")] if "(" in n else n
        per[n[-70:]][0] += 1; per[n[-70:]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]: print(f"{k:<72} {n:>5} {t/n/1e3:>9.1f} us")
PY
rm -rf $O/prof
HC_CONV_S2_STEM_WGRAD=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_off.json 2> $O/bench_off.err; echo "off: $(cut -c90-135 $O/bench_off.json)"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_on.json 2> $O/bench_on.err; echo "on: $(cut -c90-135 $O/bench_on.json)"
