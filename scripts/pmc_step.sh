#!/bin/bash
# HBM-side traffic per launch of the conv kernel families over the headline training step (bench.py, eager launches so that every
# dispatch is counted): two separate PMC passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md
# prescribes.  FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes for 16-byte-per-lane reads); both counters
# are in KiB.  Writes gpurun_out/pmc_step/traffic.json (copy to profiles/).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_step; rm -rf gpurun_out/pmc_step/*
CMD="python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_step -o f --output-format csv -- $CMD > gpurun_out/pmc_step/run_f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_step -o w --output-format csv -- $CMD > gpurun_out/pmc_step/run_w.log 2>&1
python - <<'PY'
import csv, glob, collections, json
fam = lambda k: ("conv_gather" if "conv_gather_kernel" in k else "conv_rows" if "conv_rows_kernel" in k else
                 "conv_small" if ("conv_small" in k or "conv_resident" in k) else
                 "conv_wgrad" if (("wgrad" in k or "wrep_kernel" in k) and "reduce" not in k and "dw3x3" not in k) else
                 "bn_elementwise" if ("rep_apply_kernel" in k or "rep_bwd_apply" in k or "rep_bwd_reduce" in k or "channel_stats" in k) else None)
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("gpurun_out/pmc_step/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k is None:
            continue
        a = agg[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline",
       "note": "bytes per launch, averaged over every launch of the family (warm-up + timed steps); read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB"}
for k in ("conv_gather", "conv_rows", "conv_small", "conv_wgrad", "bn_elementwise"):
    fr, fn = agg.get((k, "FETCH_SIZE"), [0, 0]); wr, wn = agg.get((k, "WRITE_SIZE"), [0, 0])
    if fn and wn:
        out[k] = {"launches_counted": fn, "read_bytes_per_launch": 2 * 1024 * fr / fn, "write_bytes_per_launch": 1024 * wr / wn,
                  "hbm_bytes_per_launch": 2 * 1024 * fr / fn + 1024 * wr / wn}
json.dump(out, open("gpurun_out/pmc_step/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
