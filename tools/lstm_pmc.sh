#!/bin/bash
# Counter passes over tools/lstm_probe.py (one config-2 BiLSTM layer; run on the GPU box): bash tools/lstm_pmc.sh <tag>
# Each pass is its own rocprofv3 run (PMC only, no tracing), bounded by `timeout`, and prints a progress line.
TAG=${1:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/lstm_pmc_$TAG
cd /tmp && export TMPDIR=/tmp
export MS_LSTM_STAMPS=0   # the shipped kernel, not the stamped variant
pass() {
  local name=$1; shift
  echo "[lstm_pmc] pass $name: $*"
  timeout -k 10 150 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/lstm_probe.py > $OUT.$name.log 2>&1 || { echo "[lstm_pmc] pass $name failed or timed out"; return 1; }
}
mkdir -p $OUT
pass p1 TCC_HIT_sum TCC_MISS_sum || exit 1
pass p2 FETCH_SIZE || exit 1
pass p3 WRITE_SIZE || exit 1
pass p4 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/lstm_pmc_$TAG.txt
find $OUT -name "*.csv" -size +1M -delete
cat $GRAFT_REPO_ROOT/gpurun_out/lstm_pmc_$TAG.txt
