"""Kernel sequence of one steady-state streaming chunk from a rocprofv3 kernel trace: the launches between two consecutive
occurrences of the chunk's first convolution kernel late in the run (csv path [, marker substring])."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "nchw_to_cl_split_kernel"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < 4:
    idx = [i for i, r in enumerate(rows) if "ft_to_tf_split_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%5.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
print("chunk: %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
