#!/usr/bin/env python3
"""Kernel-trace CSV (rocprofv3 --kernel-trace) -> how the device's time divides: nothing running, the persistent recurrence
alone, the recurrence with a co-tenant kernel, other kernels without a recurrence.  python tools/timeline_summary.py <dir> [skip_ms]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "lstm" if "lstm_persistent" in n else "other", n))
rows.sort()
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
t_begin = rows[0][0] + skip
rows = [r for r in rows if r[0] >= t_begin]
events = []
for s, e, k, _ in rows:
    events.append((s, 1, k))
    events.append((e, -1, k))
events.sort()
cnt = {"lstm": 0, "other": 0}
acc = {"idle": 0, "lstm alone": 0, "lstm + co-tenant": 0, "other only": 0}
prev = events[0][0]
for t, d, k in events:
    dt = t - prev
    if cnt["lstm"] and cnt["other"]:
        acc["lstm + co-tenant"] += dt
    elif cnt["lstm"]:
        acc["lstm alone"] += dt
    elif cnt["other"]:
        acc["other only"] += dt
    else:
        acc["idle"] += dt
    cnt[k] += d
    prev = t
total = events[-1][0] - events[0][0]
print(f"span {total / 1e6:.2f} ms over {len(rows)} kernel launches")
for k, v in acc.items():
    print(f"  {k:18s} {v / 1e6:8.2f} ms  {100.0 * v / total:5.1f} %")
by = {}
for s, e, k, n in rows:
    import re
    m = re.search(r"([A-Za-z_0-9]+_kernel\w*)", n)
    key = m.group(1) if m else n[:50]
    by.setdefault(key, [0, 0])
    by[key][0] += 1
    by[key][1] += e - s
for key, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {d / 1e6:8.2f} ms in {c:5d} launches (mean {d / c / 1e3:8.1f} us)  {key}")
