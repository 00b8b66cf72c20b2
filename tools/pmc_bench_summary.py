#!/usr/bin/env python3
"""rocprofv3 --pmc counter CSVs of tools/pmc_bench.sh -> the JSON bench.py reads (profiles/r03_pmc_bench.json).

Per kernel and launch: HBM bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies
128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact; both are reported in KB), MFMA FLOP = SQ_INSTS_MFMA x FLOP
of the kernel's MFMA shape, L2 hit rate, MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES).
The split-operand GEMM (gemm_nt_bf16x3_kernel4, the LDS-DMA kernel) is reported per contraction length: its five full-grid launches per step are, in dispatch order,
layer 1 (K = 640) and layers 2-5 (K = 2048)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, precision = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "f16x3")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# FLOP per MFMA wave-instruction of each kernel's shape
MFMA_FLOP = {"lstm_persistent_split2_kernel": 2 * 16 * 16 * 32, "lstm_persistent_wide2_kernel": 2 * 16 * 16 * 32, "gemm_nt_bf16x3_kernel2": 2 * 32 * 32 * 16,
             "gemm_nt_bf16x3_kernel4": 2 * 32 * 32 * 16,
             "lstm_persistent_f32x2_kernel": 2 * 16 * 16 * 4, "gemm_nt_f32_kernel": 2 * 32 * 32 * 2,
             "maskconv_cl_kernel": 2 * 32 * 32 * 16}

rows = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [(dispatch, grid, value)]
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(\w+_kernel\d?)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:60]
            rows[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))


def mean(v):
    return sum(v) / len(v) if v else None


def summarise(counters):
    c = {k: mean([x[2] for x in v]) for k, v in counters.items()}
    out = {"launches_sampled": max(len(v) for v in counters.values()), "counters": {k: round(v, 1) for k, v in c.items()}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES"):
        out["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"]), 4)
    return out


kernels = {}
for name, counters in rows.items():
    if not any(k in name for k in ("lstm", "gemm", "maskconv", "gru")):
        continue
    if name == ("gemm_nt_f32_kernel" if precision == "f32" else "gemm_nt_bf16x3_kernel4"):   # the projection GEMM of this mode
        # full-grid launches of the ONE-batch leg: M = 16032 -> 63 x 32 tiles of 512 threads (f32: the largest grid seen);
        # the 64-utterance launches of the two-batches-per-forward passes have a larger grid and land in "@other"
        grids = sorted({g for v in counters.values() for _, g, _ in v})
        grid_max = 63 * 32 * 512 if (precision != "f32" and 63 * 32 * 512 in grids) else grids[-1]
        split = {"@K640": defaultdict(list), "@K2048": defaultdict(list), "@other": defaultdict(list)}
        for cname, v in counters.items():
            full = sorted(x for x in v if x[1] == grid_max)
            for i, x in enumerate(full):
                split["@K640" if i % 5 == 0 else "@K2048"][cname].append(x)
            for x in v:
                if x[1] != grid_max:
                    split["@other"][cname].append(x)
        for sfx, cs in split.items():
            if cs:
                kernels[name + sfx] = summarise(cs)
                kernels[name + sfx]["mfma_flop"] = kernels[name + sfx]["counters"].get("SQ_INSTS_MFMA", 0) * MFMA_FLOP[name]
        continue
    if name == "lstm_persistent_wide2_kernel":
        # one launch serves one batch group (128 workgroups of 512 threads: the one-batch leg) or two side by side (256)
        by_groups = {"@1group": defaultdict(list), "@2groups": defaultdict(list)}
        for cname, v in counters.items():
            for x in v:
                by_groups["@2groups" if x[1] >= 256 * 512 else "@1group"][cname].append(x)
        for sfx, cs in by_groups.items():
            if cs:
                kernels[name + sfx] = summarise(cs)
                kernels[name + sfx]["mfma_flop"] = kernels[name + sfx]["counters"].get("SQ_INSTS_MFMA", 0) * MFMA_FLOP[name]
        continue
    kernels[name] = summarise(counters)
    if name in MFMA_FLOP:
        kernels[name]["mfma_flop"] = kernels[name]["counters"].get("SQ_INSTS_MFMA", 0) * MFMA_FLOP[name]

import bench  # noqa: E402  (source digest only; no GPU use)
print(json.dumps({
    "source": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_*) over "
              "`python3 bench.py --steps 2 --warmup 1` (tools/pmc_bench.sh); per-launch means",
    "precision": precision, "source_sha16": bench.source_sha16(),
    "corrections": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts 64 B per 128-B request "
                   "(MI355X_MICROARCH.md, HBM); strided 4-byte reads are not separately calibrated",
    "kernels": kernels}, indent=1))
