#!/bin/bash
# Counter passes over tools/gemm_probe.py (run on the GPU box): bash tools/gemm_pmc.sh <tag>
set -e
TAG=${1:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d $OUT/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY -d $OUT/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum FETCH_SIZE -d $OUT/p3 -o p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD WRITE_SIZE -d $OUT/p4 -o p4 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG.txt
find $OUT -name "*.csv" -size +1M -delete
cat $GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG.txt
