#!/bin/bash
# Counter passes over tools/gemm_probe.py (run on the GPU box): PROBE_VARIANTS=2,4 bash tools/gemm_pmc.sh <tag>
# Each pass is its own rocprofv3 run (PMC only, no tracing), bounded by `timeout`; the program after `--` is python3 itself.
TAG=${1:-base}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/gemm_pmc_$TAG
cd /tmp && export TMPDIR=/tmp
export PROBE_ROUNDS=${PROBE_ROUNDS:-2}
pass() {
  local name=$1; shift
  echo "[gemm_pmc] pass $name: $*"
  timeout -k 10 150 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/tools/gemm_probe.py > $OUT.$name.log 2>&1 || { echo "[gemm_pmc] pass $name failed or timed out"; return 1; }
}
mkdir -p $OUT
pass p1 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY || exit 1
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE || exit 1
pass p3 FETCH_SIZE || exit 1
pass p4 TCC_HIT_sum TCC_MISS_sum || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ROOT/tools/gemm_probe.py > $OUT.trace.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT > $ROOT/gpurun_out/gemm_pmc_$TAG.txt
grep -h "gemm_nt" $OUT/trace/*kernel_stats.csv | cut -c1-60,150-400 >> $ROOT/gpurun_out/gemm_pmc_$TAG.txt
find $OUT -name "*.csv" -size +1M -delete
cat $ROOT/gpurun_out/gemm_pmc_$TAG.txt
