#!/bin/bash
# HBM-side traffic per launch of every kernel family over one training-step command (eager launches so that every dispatch is
# counted): two separate PMC passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
# FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes for 16-byte-per-lane reads); both counters are in KiB.
#   usage: pmc_families.sh <name> <command ...>      -> gpurun_out/pmc_<name>/traffic.json  (copy to profiles/rNN_pmc_<name>_traffic.json)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NAME=$1; shift
O=gpurun_out/pmc_$NAME; mkdir -p $O; rm -rf $O/*
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o f --output-format csv -- "$@" > $O/run_f.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o w --output-format csv -- "$@" > $O/run_w.log 2>&1
python - "$O" "$*" <<'PY'
import csv, glob, collections, json, sys
O, cmd = sys.argv[1], sys.argv[2]
def fam(k):
    if "stem_fused_kernel<1>" in k: return "bn_elementwise"          # the stem's fused passes are filed like bench.py files them
    if "stem_fused_kernel<2>" in k: return "conv_wgrad"
    if "stem_bwd_" in k: return None
    if "cs2::" in k: return "conv_wgrad" if "wgrad" in k and "reduce" not in k else ("conv_s2" if "reduce" not in k else None)
    if "conv_gather_kernel" in k or "conv_pw_kernel" in k: return "conv_gather"
    if "conv_rows" in k: return "conv_rows"
    if "conv_small" in k or "conv_resident" in k: return "conv_small"
    if ("wgrad" in k or "wrep_kernel" in k) and "reduce" not in k and "dw3x3" not in k: return "conv_wgrad"
    if "dw3x3" in k or "dwrep" in k: return "dwconv"
    if any(t in k for t in ("rep_apply_kernel", "rep_bwd_apply", "rep_bwd_reduce", "channel_stats", "bn_act_", "se_scale", "msbn_apply", "msbn_bwd")): return "bn_elementwise"
    if "adabelief" in k: return "optimizer"
    if "pack_weight" in k: return "weight_pack"
    return None
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k is None: continue
        a = agg[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- " + cmd,
       "note": "bytes per launch, averaged over every launch of the family (warm-up + timed steps); read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB"}
for k in sorted({k for k, _ in agg}):
    fr, fn = agg.get((k, "FETCH_SIZE"), [0, 0]); wr, wn = agg.get((k, "WRITE_SIZE"), [0, 0])
    if fn and wn:
        out[k] = {"launches_counted": fn, "read_bytes_per_launch": 2 * 1024 * fr / fn, "write_bytes_per_launch": 1024 * wr / wn,
                  "hbm_bytes_per_launch": 2 * 1024 * fr / fn + 1024 * wr / wn}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
