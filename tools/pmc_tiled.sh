#!/bin/bash
# HBM-side counters and LDS / issue counters of the tiled sweep (C5 by default), separate --pmc passes, nothing else traced.
# Usage on the GPU box: bash tools/pmc_tiled.sh [c5|c3]   -> gpurun_out/pmc_tiled_<cfg>.json
set -e
CFG=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_tiled_$CFG
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 tools/tiled_bench.py $CFG pmc > "$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.log"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/tiled_bench.py $CFG pmc > "$OUT/stats.log" 2>&1
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "tiled_sweep_kernel" in kn:
            acc["T" if ("Lb1E" in kn or ", true," in kn) else "K"][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {m: {k: sum(v) / len(v) for k, v in d.items()} for m, d in acc.items()}
st = {}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tiled_sweep_kernel" in r["Name"]:
            st["T" if ("Lb1E" in r["Name"] or ", true," in r["Name"]) else "K"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
json.dump({"config": "$CFG", "counters_avg_per_launch": res, "kernel_stats": st, "note": "K = k-stream form, T = temperature form; FETCH_SIZE / WRITE_SIZE in KiB"}, open("gpurun_out/pmc_tiled_$CFG.json", "w"), indent=1)
print(json.dumps({"stats": st, "K": res.get("K"), "T": res.get("T")}, indent=1))
PY
find "$OUT" -name "*.csv" -size +200k -delete
