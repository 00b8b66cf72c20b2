#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(\w+_kernel\d?)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    if not any(k in name for k in ("lstm", "gemm", "maskconv")):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:20s} n={len(v):4d} mean={sum(v) / len(v):.4g}")
