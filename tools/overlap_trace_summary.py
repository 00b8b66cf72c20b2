#!/usr/bin/env python3
"""Kernel-trace CSV of tools/overlap_probe.py (rocprofv3 --kernel-trace) -> were projection GEMMs and the persistent LSTM
actually resident at the same time, and how long did each take alone vs overlapped?"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            kind = "lstm" if "lstm_persistent" in n else ("gemm" if "gemm_nt_bf16x3_kernel4" in n else None)
            if kind:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
lstm = [r for r in rows if r[2] == "lstm"]
gemm = [r for r in rows if r[2] == "gemm" and r[3] >= 500000]     # the full-size projections (not FC-sized launches)


def overlap(a, others):
    return sum(max(0, min(a[1], o[1]) - max(a[0], o[0])) for o in others)


for name, ks, others in (("lstm", lstm, gemm), ("gemm", gemm, lstm)):
    alone = [k[1] - k[0] for k in ks if overlap(k, others) < 0.05 * (k[1] - k[0])]
    mixed = [(k[1] - k[0], overlap(k, others)) for k in ks if overlap(k, others) >= 0.05 * (k[1] - k[0])]
    print(f"{name}: {len(ks)} launches; alone: {len(alone)} (mean {sum(alone) / max(len(alone), 1) / 1e3:.1f} us); "
          f"overlapped with the other kind: {len(mixed)} (mean duration {sum(m[0] for m in mixed) / max(len(mixed), 1) / 1e3:.1f} us, "
          f"mean overlapped time {sum(m[1] for m in mixed) / max(len(mixed), 1) / 1e3:.1f} us)")
