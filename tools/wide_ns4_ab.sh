#!/bin/bash
# same-box A/B: 64 sequences as two groups of two 16-row streams on 256 workgroups (shipped) against ONE group of four
# streams on 128 workgroups (MS_LSTM_WIDE_NS=4), with and without the early request for h
cd "$(dirname "$0")/.."
echo "== correctness with MS_LSTM_WIDE_NS=4"
MS_LSTM_WIDE_NS=4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "wide or paired" 2>&1 | tail -2
for r in 1 2; do
echo -n "N=64 2 groups x 2 streams: "; PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 4 streams, early    : "; MS_LSTM_WIDE_NS=4 PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 4 streams, plain    : "; MS_LSTM_WIDE_NS=4 MS_LSTM_WIDE_AHEAD=0 PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
done
echo -n "N=128 2 groups x 4 streams, early: "; MS_LSTM_WIDE_NS=4 PROBE_N=128 python tools/lstm_layer_time.py 2>&1 | tail -1
echo "== stamps, four streams, early request"
MS_LSTM_WIDE_NS=4 python tools/wide_stamps.py 2>&1 | grep -v amdgpu.ids
