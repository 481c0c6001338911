#!/bin/bash
# HBM-side counters of the drop-in batched sweep (kin_rhs_batched_klib_dev = permute in + tiled sweep + permute out at C5),
# separate --pmc passes, nothing else traced; per CALL = the sum over its kernels. Usage on the GPU box: bash tools/pmc_dropin.sh [c5|cut]
set -e
CFG=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_dropin_$CFG
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 tools/dropin_bench.py $CFG pmc > "$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.log"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/dropin_bench.py $CFG pmc > "$OUT/stats.log" 2>&1
python3 tools/dropin_bench.py $CFG > "$OUT/timing.json" 2> "$OUT/timing.err"
python3 - <<PY
import csv, glob, json, collections
names = ("permute_staged_kernel", "tiled_sweep_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for n in names:
            if n in r["Kernel_Name"]:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
calls = 4   # tools/dropin_bench.py pmc: 1 + 3 calls
per_call = {c: sum(sum(acc[n].get(c, [])) for n in names) / calls for c in ("FETCH_SIZE", "WRITE_SIZE")}
st = {}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for n in names:
            if n in r["Name"]:
                st[r["Name"][:90]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
t = json.loads(open("$OUT/timing.json").read().strip().splitlines()[-1])
hbm = (2 * per_call["FETCH_SIZE"] + per_call["WRITE_SIZE"]) * 1024
res = {"config": "$CFG", "timing_hip_events": t, "counters_per_call_KiB": per_call, "kernel_stats": st,
       "hbm_bytes_per_call": hbm, "traffic_over_algorithmic": hbm / t["algorithmic_bytes"],
       "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the kernels of one call (the guide's gfx950 correction for wide streaming reads; the permutation kernels read 8 bytes per lane, an access width the guide calls uncalibrated: their share may be over-counted by up to 2x)"}
json.dump(res, open("gpurun_out/pmc_dropin_$CFG.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +200k -delete
