"""Kernel-trace analysis helper: python tools/gj_trace.py <kernel_trace.csv> - per-kernel mean duration and the mean gap to
the previous kernel of the stream, for the Gauss-Jordan chain and the Newton chain (rocprofv3 --kernel-trace output)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kin::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name] += e - s
    if prev_end is not None and s - prev_end < 50000:      # gaps above 50 us are host waits, not launch overhead
        gap[name] += s - prev_end
        cnt[name] += 1
    prev_end = e
tot = sum(dur.values())
for name in sorted(dur, key=lambda n: -dur[n])[:14]:
    n = sum(1 for r in rows if r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kin::", "") == name)
    print(f"{name[:48]:48s} n={n:6d} dur {dur[name] / n / 1e3:7.2f} us  gap-before {gap[name] / max(cnt[name], 1) / 1e3:6.2f} us  share {dur[name] / tot * 100:5.1f} %")
