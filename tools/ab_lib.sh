#!/bin/bash
# Same-box A/B of two builds of libms_hotpath.so: tools/ab_lib.sh <lib A> <lib B> [rounds] [probe script + args]
# (alternating runs; absolute numbers move +-3 % between boxes, so only interleaved comparisons are trusted)
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; shift 3
PROBE=${@:-tools/lstm_layer_time.py}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    MS_HOTPATH_LIB=$(realpath "$L") timeout -k 10 300 python $PROBE 2>&1 | grep -v amdgpu.ids | tail -${AB_TAIL:-1} || exit 1
  done
done
