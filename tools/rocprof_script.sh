#!/bin/bash
# rocprofv3 --kernel-trace --stats over one python script (GPU box): tools/rocprof_script.sh <tag> <script.py> [args]
# writes gpurun_out/<tag>_kernel_stats.csv and the script's stdout to gpurun_out/<tag>.out
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $SCRIPT "$@" > $ROOT/gpurun_out/${TAG}.out 2> $ROOT/gpurun_out/${TAG}.err || { tail -5 $ROOT/gpurun_out/${TAG}.err; exit 1; }
cp $(find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
[ -z "$KEEP_TRACE" ] && find $ROOT/gpurun_out/prof_$TAG -name "*.csv" -size +1M -delete
cut -c1-160 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | head -12
