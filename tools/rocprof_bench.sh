#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py (GPU box): tools/rocprof_bench.sh <tag> [--in-flight 1|2]
# writes gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_bench_line_under_rocprof.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-f32-child --no-frontend --no-ragged --no-legs "$@" > $ROOT/gpurun_out/${TAG}_bench_line_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}.err || { tail -5 $ROOT/gpurun_out/${TAG}.err; exit 1; }
cp $(find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $ROOT/gpurun_out/prof_$TAG -name "*.csv" -size +1M -delete
head -8 $ROOT/gpurun_out/${TAG}_kernel_stats.csv
