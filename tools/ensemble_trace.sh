#!/bin/bash
# What limits an ensemble of solves on one GPU: rocprofv3 kernel trace of tools/ensemble_scaling.py at K = 1 and K = 4,
# per-kernel duration under contention, dispatch rate, busy fraction (union of kernel intervals) and mean concurrency.
# Usage on the GPU box: bash tools/ensemble_trace.sh -> gpurun_out/ensemble_trace.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ens_trace
rm -rf "$OUT" && mkdir -p "$OUT"
for K in 1 4; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/k$K" -- python3 tools/ensemble_scaling.py 2 $K > "$OUT/k$K.log" 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
res = {}
for K in (1, 4):
    f = glob.glob(f"gpurun_out/ens_trace/k{K}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kin::", ""))
            for r in csv.DictReader(open(f)) if "kin::" in r["Kernel_Name"]]
    rows.sort()
    # the timed window = the last solve of every handle: take the final third of the trace by time (warm-up solves come first)
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * 2 // 3
    win = [r for r in rows if r[0] >= t_lo]
    span = win[-1][1] - win[0][0]
    # union of intervals and mean concurrency
    ev = sorted([(s, 1) for s, e, _ in win] + [(e, -1) for s, e, _ in win])
    busy = 0; depth = 0; last = ev[0][0]; area = 0
    for t, d in ev:
        if depth > 0: busy += t - last
        area += depth * (t - last)
        depth += d; last = t
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in win:
        per[n][0] += 1; per[n][1] += e - s
    top = sorted(per.items(), key=lambda kv: -kv[1][1])[:10]
    res[str(K)] = {"kernels_in_window": len(win), "window_ms": span / 1e6, "dispatches_per_s": len(win) / (span / 1e9),
                   "busy_fraction": busy / span, "mean_kernels_in_flight_while_busy": area / max(busy, 1),
                   "per_kernel_avg_us": {n: round(v[1] / v[0] / 1e3, 2) for n, v in top},
                   "per_kernel_share_of_kernel_time": {n: round(v[1] / sum(x[1] for x in per.values()), 3) for n, v in top}}
json.dump(res, open("gpurun_out/ensemble_trace.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +100k -delete
