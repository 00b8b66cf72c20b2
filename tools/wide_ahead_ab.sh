#!/bin/bash
# same-box A/B of the wide-workgroup LSTM kernel with the early request for h (MS_LSTM_WIDE_AHEAD=1): parity, then layer times
cd "$(dirname "$0")/.."
echo "== correctness with MS_LSTM_WIDE_AHEAD=1"
MS_LSTM_WIDE_AHEAD=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q -k "rnn or lstm or cfg2 or ds2 or shard or utter or wide" 2>&1 | tail -2
echo "== correctness with MS_LSTM_WIDE_NS=4"
MS_LSTM_WIDE_NS=4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "wide or paired" 2>&1 | tail -2
for r in 1 2; do
echo -n "N=32 plain: "; PROBE_N=32 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=32 ahead: "; MS_LSTM_WIDE_AHEAD=1 PROBE_N=32 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 plain (2 groups x 2 streams): "; PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 ahead (2 groups x 2 streams): "; MS_LSTM_WIDE_AHEAD=1 PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 4 streams, early            : "; MS_LSTM_WIDE_NS=4 PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
done
echo "== stamps, ahead"
MS_LSTM_WIDE_AHEAD=1 python tools/wide_stamps.py 2>&1 | grep -v amdgpu.ids
echo "== stamps, four streams"
MS_LSTM_WIDE_NS=4 python tools/wide_stamps.py 2>&1 | grep -v amdgpu.ids | tail -3
