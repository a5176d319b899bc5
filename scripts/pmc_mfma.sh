#!/bin/bash
# MFMA utilisation per kernel over the headline training step (eager launches so that every dispatch is counted): one PMC pass,
# --kernel-trace only (no other trace domain: MI355X_MICROARCH.md / gpurun rule).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles the
# matrix pipe of a SIMD is busy (summed over SIMDs), GRBM_GUI_ACTIVE the cycles the kernel was on the chip:
#     mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (the gfx94x MfmaUtil formula, counter_defs.yaml; rocprofv3
#     reports GRBM_GUI_ACTIVE summed over the 8 XCDs on gfx950: GUI_ACTIVE / dispatch time came out at 19 GHz = 8 x 2.4 in the first run)
# SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 = executed bf16 MFMA flops (padding included) -> executed TFLOP/s over the dispatch time.
# Writes gpurun_out/pmc_mfma/mfma_util.txt (copy to profiles/).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_mfma; mkdir -p $O; rm -rf $O/*
CMD="${PMC_CMD:-python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1}"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O -o m --output-format csv -- $CMD > $O/run.log 2>&1
python - "$O" "$CMD" <<'PY'
import csv, glob, sys, collections
O, cmd = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        d = rows[(r["Dispatch_Id"], r["Kernel_Name"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for (_, name), d in rows.items():
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")
    k = (k[:k.index("(")] if "(" in k else k)[:90]
    a = agg[k]
    a["n"] += 1
    for c, v in d.items():
        a[c] += v
with open(O + "/mfma_util.txt", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- {cmd}\n")
    fh.write("# mfma_util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); exec_TF = MOPS_BF16 * 512 / time (padding included); eff_GHz = GUI_ACTIVE / 8 / time\n")
    fh.write(f"{'kernel':<92} {'calls':>6} {'total_ms':>9} {'avg_us':>8} {'mfma_util':>9} {'exec_TF/s':>9} {'eff_GHz':>7}\n")
    tot_ns = tot_busy = tot_act = tot_mops = 0.0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        ns, act = a["ns"], a.get("GRBM_GUI_ACTIVE", 0.0)
        busy, mops = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        tot_ns += ns; tot_busy += busy; tot_act += act; tot_mops += mops
        if a["n"] and ns > 0 and (busy > 0 or ns > 2e5):
            util = busy / (act / 8 * 1024) if act else 0.0
            fh.write(f"{k:<92} {int(a['n']):>6} {ns / 1e6:>9.3f} {ns / a['n'] / 1e3:>8.1f} {100 * util:>8.1f}% {mops * 512 / ns / 1e3:>9.1f} {act / 8 / ns if ns else 0:>7.2f}\n")
    fh.write(f"{'ALL KERNELS':<92} {'':>6} {tot_ns / 1e6:>9.3f} {'':>8} {100 * tot_busy / (tot_act / 8 * 1024) if tot_act else 0:>8.1f}% {tot_mops * 512 / tot_ns / 1e3 if tot_ns else 0:>9.1f}\n")
print(open(O + "/mfma_util.txt").read()[:6000])
PY
