#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3j; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_fullsize_bn.py tests/test_gpu_repvgg.py tests/test_gpu_bn_zmask.py tests/test_gpu_darknet.py tests/test_gpu_rexnet.py tests/test_gpu_fullsize.py tests/test_gpu_mobileone.py -k "graph_with_packed or bn_passes or bn_act or repvgg or zmask or darknet or rexnet or fullsize or mobileone" -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench: $(cut -c90-135 $O/bench.json)"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3j/prof/**/*kernel_trace.csv", recursive=True)
if f:
    per = collections.defaultdict(lambda: [0, 0]); tot = 0
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[n[:80]][0] += 1; per[n[:80]][1] += t; tot += t
    with open("gpurun_out/r3j/kernel_stats.txt", "w") as fh:
        for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]: fh.write(f"{k:<82} {n:>5} {t/1e6:>9.3f} ms {t/n/1e3:>9.1f} us\n")
        fh.write(f"TOTAL {tot/1e6:.3f} ms\n")
    print("\n".join(l for l in open("gpurun_out/r3j/kernel_stats.txt").read().splitlines() if "finalize" in l or "TOTAL" in l))
PY
rm -rf $O/prof
