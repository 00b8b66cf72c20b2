"""Median duration per group of consecutive launches of the CTC alpha kernels in a rocprofv3 kernel trace (csv path, group size)."""
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ctc_alpha" in r["Kernel_Name"]]
g = int(sys.argv[2]) if len(sys.argv) > 2 else 23
for i in range(0, len(rows), g):
    grp = rows[i:i + g]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp]
    print(grp[0]["Kernel_Name"][22:62], len(grp), "%.1f us" % (statistics.median(d) / 1000))
