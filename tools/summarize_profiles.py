"""Turns the rocprofv3 output of tools/collect_profiles.sh into the small files committed under
profiles/ (kernel-stats CSV of the kin:: kernels + a JSON with the PMC traffic of the sweep kernel,
corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB and FETCH_SIZE
counts half of a wide coalesced read on gfx950)."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if "kin::" in r["Name"]]
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(keep)


def pmc(sub, name):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "kin::sweep_" in r["Kernel_Name"] and r["Counter_Name"] == name]


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
sweep = [r for r in rows if "kin::sweep_" in r["Name"]][0]
line = [l for l in open(os.path.join(out, "bench_stats.log")) if l.startswith("{")][-1]
bench = json.loads(line)
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --solve-chunks 20 --cpu-solve-chunks 0 --no-pmc --sustain-seconds 0.5 ; "
               "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 5 --solve-chunks 0 --no-cpu --no-pmc --sustain-seconds 0",
    "workload": bench["config"]["workload"],
    "kernel": sweep["Name"].split("(")[0],
    "calls": int(sweep["Calls"]), "avg_ns_rocprof": float(sweep["AverageNs"]),
    "avg_launch_ms_bench_events": bench["roofline"]["avg_launch_ms"],
    "FETCH_SIZE_KiB_avg": sum(fetch) / len(fetch), "WRITE_SIZE_KiB_avg": sum(write) / len(write),
    "hbm_bytes_per_launch_corrected": (2.0 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024.0,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "bench_line_under_profiler": bench,
}
timed = glob.glob(os.path.join(out, "stats_timed", "**", "*kernel_stats.csv"), recursive=True)
if timed:      # warm-up + timed launches only (no sustained leg): the figure to hold against roofline.achieved
    trow = [r for r in csv.DictReader(open(timed[0])) if "kin::sweep_" in r["Name"]][0]
    tline = [l for l in open(os.path.join(out, "bench_stats_timed.log")) if l.startswith("{")][-1]
    summary["timed_launches_only"] = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --solve-chunks 0 --no-cpu --no-pmc --sustain-seconds 0",
                                      "calls": int(trow["Calls"]), "avg_ns_rocprof": float(trow["AverageNs"]),
                                      "avg_launch_ms_bench_events": json.loads(tline)["roofline"]["avg_launch_ms"]}
summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch_corrected"] / summary["algorithmic_bytes_per_launch"]
json.dump(summary, open(os.path.join(dst, f"{tag}_sweep_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line_under_profiler"}, indent=1))


# ---- C5 large-N sweep and the C4-size rate table (tools/run_configs.py c5sweep table)
def pmc_of(sub, name, kernel):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]


cfg_stats = glob.glob(os.path.join(out, "cfg_stats", "**", "*kernel_stats.csv"), recursive=True)
if cfg_stats:
    crow = list(csv.DictReader(open(cfg_stats[0])))
    recs = [json.loads(l) for l in open(os.path.join(out, "cfg_stats.log")) if l.startswith("{")]
    cfg = {"command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python3 tools/run_configs.py c5sweep table",
           "records_under_profiler": recs, "kernels": []}
    for kern, alg in (("kin::sweep_big_kernel", 20 * 250000 + 1024 * (8 * 250000 + 16 * 50000)),
                      ("kin::rate_table_kernel", 16 * 50000 + 8 * 14001 + 8 * 14001 * 50000)):
        row = [r for r in crow if kern in r["Name"]][0]
        fe, wr = pmc_of("cfg_fetch", "FETCH_SIZE", kern), pmc_of("cfg_write", "WRITE_SIZE", kern)
        hbm = (2.0 * sum(fe) / len(fe) + sum(wr) / len(wr)) * 1024.0
        avg_ns = float(row["AverageNs"])
        cfg["kernels"].append({"kernel": kern, "calls": int(row["Calls"]), "avg_ns_rocprof": avg_ns,
                               "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / avg_ns,
                               "frac_of_8TBps": alg / avg_ns / 8000.0,
                               "FETCH_SIZE_KiB_avg": sum(fe) / len(fe), "WRITE_SIZE_KiB_avg": sum(wr) / len(wr),
                               "hbm_bytes_per_launch_corrected": hbm, "traffic_over_algorithmic": hbm / alg})
    json.dump(cfg, open(os.path.join(dst, f"{tag}_c5_table_pmc.json"), "w"), indent=1)
    print(json.dumps(cfg["kernels"], indent=1))


# ---- the implicit solve (tools/solve_stats.py 10000 50000 20 = warm-up solve + timed solve)
sol = glob.glob(os.path.join(out, "solve_stats", "**", "*kernel_stats.csv"), recursive=True)
if sol:
    srows = list(csv.DictReader(open(sol[0])))
    with open(os.path.join(dst, f"{tag}_solve_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=srows[0].keys())
        w.writeheader()
        w.writerows([r for r in srows if "kin::" in r["Name"]])
    rec = [l for l in open(os.path.join(out, "solve_stats.log")) if l.startswith("{")]
    if rec:
        open(os.path.join(dst, f"{tag}_solve_stats.json"), "w").write(rec[-1])


# ---- round 4: resident integrator (300 species, 20 chunks) and lockstep ensemble (10k species, K = 16)
for sub, log, name in (("resident_stats", "resident_stats.log", "resident"), ("ensemble_stats", "ensemble_stats.log", "ensemble_lockstep")):
    st = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    srows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=srows[0].keys())
        w.writeheader()
        w.writerows([r for r in srows if "kin::" in r["Name"]])
    rec = [l for l in open(os.path.join(out, log)) if l.startswith("{")]
    if rec:
        open(os.path.join(dst, f"{tag}_{name}_stats.json"), "w").write(rec[-1])
