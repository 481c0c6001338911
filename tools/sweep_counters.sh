#!/bin/bash
# SQ / LDS counters of the sweep kernel (separate --pmc passes, nothing else traced). Usage on the GPU box:
#   bash tools/sweep_counters.sh  -> gpurun_out/sweep_counters.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sweep_ctr
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 5 --solve-chunks 0 --no-cpu --no-pmc --no-tiled --sustain-seconds 0 > "$OUT/p$i.log" 2>&1
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kin::sweep_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open("gpurun_out/sweep_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -delete
