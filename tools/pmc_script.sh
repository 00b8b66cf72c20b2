#!/bin/bash
# Counter passes over one python script (GPU box): tools/pmc_script.sh <tag> <script.py> [args]
# Each pass is its own rocprofv3 run (PMC only, no tracing flags); the program after `--` is python3 itself.
# Writes gpurun_out/<tag>_pmc_summary.txt: per kernel name the mean of every counter over its dispatches (kernels whose name
# holds one of the comma-separated PMC_KERNELS substrings).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=$ROOT/$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $SCRIPT $ARGS > $OUT.$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT.$name.log; return 1; }
}
ARGS="$@"
pass p1 FETCH_SIZE || exit 1
pass p2 WRITE_SIZE || exit 1
pass p3 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS || exit 1
pass p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
python3 - "$OUT" "${PMC_KERNELS:-ctc,pred_layer,gemm_nt_bf16x3_tile64,lookahead,gru_persistent,lstm_persistent_split2}" > $ROOT/gpurun_out/${TAG}_pmc_summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not any(s in k for s in sys.argv[2].split(",")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        # FETCH_SIZE / WRITE_SIZE: units of 64 B (32 B on some parts: MI355X_MICROARCH.md prescribes x2 for FETCH_SIZE on gfx950), summed over XCDs per dispatch row
        print("   %-22s mean %.4g over %d dispatch rows" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
cat $ROOT/gpurun_out/${TAG}_pmc_summary.txt
