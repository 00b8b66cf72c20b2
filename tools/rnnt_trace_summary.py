#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of tools/rnnt_probe.py: mean duration per (kernel, grid size)."""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:50]
            acc[(name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 50:
        continue
    print(f"{name:32s} grid {grid:>8s} n={len(v):6d} mean {sum(v) / len(v) / 1e3:7.2f} us  total {sum(v) / 1e6:7.2f} ms")
