#!/bin/bash
# HBM-side traffic of the conv kernels on one micro-benchmarked layer: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes (no tracing domains besides --kernel-trace).  usage: pmc_traffic.sh <bench_layers args>
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2; rm -rf gpurun_out/pmc2/*
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc2 -o f --output-format csv -- python scripts/bench_layers.py "$@" > gpurun_out/pmc2/run_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc2 -o w --output-format csv -- python scripts/bench_layers.py "$@" > gpurun_out/pmc2/run_w.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc2/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in agg.items():
        if "conv_gather" in k or "wgrad" in k or "conv_small" in k:
            print(f.split("/")[-1][:3], k, c, "per launch:", round(v / n, 1), "launches", n)
PY
