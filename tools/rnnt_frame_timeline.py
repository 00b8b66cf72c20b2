#!/usr/bin/env python3
"""One frame of the RNN-T beam decode as a timeline (rocprofv3 --kernel-trace CSV of tools/rnnt_beam_probe.py): for the launches
of a frame in the middle of the last decode, start / end relative to the frame's first kernel, duration and queue -- whether
the side stream's G launches really run beside the joint / round kernels.
    python tools/rnnt_frame_timeline.py <dir with *kernel_trace.csv> [frame index, default: the middle] [frames per decode, default 501]"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: (re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0]
ends = [i for i, r in enumerate(rows) if "frame_end" in r["Kernel_Name"]]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 501       # frames of one decode (T' of the probe's clips)
k = int(sys.argv[2]) if len(sys.argv) > 2 else frames // 2
k = len(ends) - frames + k       # the last decode's frames
lo, hi = ends[k - 1] + 1, ends[k]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi + 3]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:8.2f} .. {e / 1e3:8.2f} us  {(e - s) / 1e3:6.2f}  q{r.get('Queue_Id', '?'):>3s}  {name(r)}")
