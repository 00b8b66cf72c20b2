#!/bin/bash
# GPU box: correctness of the LSTM paths, then stamps + layer wall time for the one-stream and
# two-stream split kernels.
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout 200 -x -k "rnn or lstm or ds2 or ds1" -s 2>&1 | grep -v amdgpu.ids | tail -5 || exit 1
echo "== two-stream"; timeout -k 10 200 python tools/lstm_probe.py 2>&1 | grep -v amdgpu.ids
echo "== one-stream"; MS_LSTM_ONE_STREAM=1 timeout -k 10 200 python tools/lstm_probe.py 2>&1 | grep -v amdgpu.ids
