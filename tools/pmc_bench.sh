#!/bin/bash
# Counter passes over bench.py itself (HEAD's chained 5-layer stack, the binary the bench times), on the GPU box:
#     bash tools/pmc_bench.sh [tag] [precision]
# Each pass is its own rocprofv3 run (PMC only -- no tracing flags), bounded by `timeout`; the program after `--` is
# python3 itself.  Writes gpurun_out/pmc_bench_<tag>.json (copy it to profiles/r06_pmc_bench.json) and a text summary.
TAG=${1:-r06}
PREC=${2:-f16x3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_bench_$TAG
cd /tmp && export TMPDIR=/tmp
[ "$PREC" != "f16x3" ] && export MS_PRECISION=$PREC
# the layer-by-layer schedule: every recurrence launch is a whole layer (501 steps), which is what the roofline is stated on
# (the overlapped schedule cuts a layer into time segments: its launches are shorter and run beside a GEMM)
export MS_RNN_OVERLAP=0
pass() {
  local name=$1; shift
  echo "[pmc_bench] pass $name: $*"
  timeout -k 10 240 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-child --no-frontend --no-ragged --no-legs --reps 0 --detail-path /dev/null --in-flight 1 > $OUT.$name.log 2>&1 || { echo "[pmc_bench] pass $name failed or timed out"; tail -5 $OUT.$name.log; return 1; }
}
mkdir -p $OUT
pass p1 FETCH_SIZE || exit 1
pass p2 WRITE_SIZE || exit 1
pass p3 TCC_HIT_sum TCC_MISS_sum || exit 1
pass p4 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
# the two-batches-per-forward mode (pipeline.PairedBatches): the wide-workgroup recurrence and the 64-utterance GEMMs
passw() {
  local name=$1; shift
  echo "[pmc_bench] pass $name: $*"
  PROBE_STEPS=2 timeout -k 10 240 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/tools/n64_probe.py > $OUT.$name.log 2>&1 || { echo "[pmc_bench] pass $name failed or timed out"; tail -5 $OUT.$name.log; return 1; }
}
if [ "$PREC" == "f16x3" ] || [ "$PREC" == "bf16x3" ]; then
  passw w1 FETCH_SIZE || exit 1
  passw w2 WRITE_SIZE || exit 1
  passw w3 TCC_HIT_sum TCC_MISS_sum || exit 1
  passw w4 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
fi
python3 $ROOT/tools/pmc_bench_summary.py $OUT $PREC > $ROOT/gpurun_out/pmc_bench_$TAG.json || exit 1
find $OUT -name "*.csv" -size +1M -delete
cat $ROOT/gpurun_out/pmc_bench_$TAG.json
