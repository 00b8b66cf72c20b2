#!/bin/bash
# rocprofv3 --kernel-trace --stats over ONE leg of tools/bench_configs.py (GPU box):
#   tools/rocprof_configs.sh <tag> <leg> [ENV=VALUE ...]      leg in: ds1 ctc beam rnnt stream frontend
# writes gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_line_under_rocprof.json.  The program is the interpreter itself
# (no env / bash -c hop between rocprofv3 and python: the profiler initialises the GPU before the program starts).
TAG=$1; LEG=$2; shift 2
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $ROOT/tools/bench_configs.py $LEG --no-cpu --line > $ROOT/gpurun_out/${TAG}_line_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}.err || { tail -5 $ROOT/gpurun_out/${TAG}.err; exit 1; }
cp $(find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $ROOT/gpurun_out/prof_$TAG -name "*.csv" -size +1M -delete
head -12 $ROOT/gpurun_out/${TAG}_kernel_stats.csv
