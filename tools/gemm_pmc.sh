#!/bin/bash
# Counter passes over tools/gemm_probe.py (run on the GPU box): bash tools/gemm_pmc.sh <tag>
# Each pass is its own rocprofv3 run (PMC only, no tracing), bounded by `timeout`, and prints a progress line.
TAG=${1:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  echo "[gemm_pmc] pass $name: $*"
  timeout -k 10 150 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > $OUT.$name.log 2>&1 || { echo "[gemm_pmc] pass $name failed or timed out"; return 1; }
}
mkdir -p $OUT
pass p1 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA || exit 1
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY || exit 1
pass p3 TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG.txt
find $OUT -name "*.csv" -size +1M -delete
cat $GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG.txt
