#!/bin/bash
# Where a solve's wall-clock goes on the device: rocprofv3 kernel trace of tools/solve_stats.py (C3, 20 chunks, 4 solves),
# summed kernel time, idle time between kernels (histogram of the gaps), per-kernel duration and gap before it.
# Usage on the GPU box: bash tools/solve_gaps.sh  -> gpurun_out/solve_gaps.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/solve_gaps
rm -rf "$OUT" && mkdir -p "$OUT"
SOLVE_REPEATS=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/solve_stats.py 10000 50000 20 > "$OUT/solve.log" 2>&1
python3 - <<'PY'
import csv, glob, json, collections
f = glob.glob("gpurun_out/solve_gaps/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kin::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the second solve only (the first contains the symbolic analysis): split at the largest gap between kernels
starts = [int(r["Start_Timestamp"]) for r in rows]; ends = [int(r["End_Timestamp"]) for r in rows]
gaps = [starts[i + 1] - ends[i] for i in range(len(rows) - 1)]
cut = max(range(len(gaps)), key=lambda i: gaps[i]) + 1
rows, starts, ends = rows[cut:], starts[cut:], ends[cut:]
gaps = [max(0, starts[i + 1] - ends[i]) for i in range(len(rows) - 1)]
span = ends[-1] - starts[0]
busy = sum(e - s for s, e in zip(starts, ends))
edges = [0, 1000, 2000, 3000, 5000, 10000, 20000, 50000, 10 ** 12]
hist = collections.OrderedDict()
for lo, hi in zip(edges[:-1], edges[1:]):
    g = [x for x in gaps if lo <= x < hi]
    hist[f"{lo / 1e3:g}-{hi / 1e3:g} us" if hi < 10 ** 12 else f">= {lo / 1e3:g} us"] = {"count": len(g), "total_ms": sum(g) / 1e6}
per = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(rows):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kin::", "")
    per[n][0] += 1; per[n][1] += ends[i] - starts[i]
    if i > 0: per[n][2] += gaps[i - 1]
top = sorted(per.items(), key=lambda kv: -kv[1][1])[:12]
rec = [l for l in open("gpurun_out/solve_gaps/solve.log") if l.startswith("{")]
out = {"workload": "C3 (10k / 50k), static 1000 K, 20 chunks of 1 ms, second solve on a warm handle, under rocprofv3 --kernel-trace",
       "solve_record": json.loads(rec[-1]) if rec else None, "kernels": len(rows), "span_ms": span / 1e6, "kernel_time_ms": busy / 1e6,
       "device_busy_fraction": busy / span, "idle_between_kernels_ms": sum(gaps) / 1e6, "gap_histogram": hist,
       "per_kernel": {n: {"launches": v[0], "avg_us": v[1] / v[0] / 1e3, "avg_gap_before_us": v[2] / v[0] / 1e3, "share_of_kernel_time": v[1] / busy} for n, v in top}}
json.dump(out, open("gpurun_out/solve_gaps.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernels", "span_ms", "kernel_time_ms", "device_busy_fraction", "idle_between_kernels_ms", "gap_histogram")}, indent=1))
PY
rm -rf "$OUT/trace"
