#!/bin/bash
# same-box A/B of the wide-workgroup LSTM kernel with k-step-wise tag check + MFMA (MS_LSTM_WIDE_KPIPE=1): parity, layer times, stamps
cd "$(dirname "$0")/.."
echo "== correctness with MS_LSTM_WIDE_KPIPE=1"
MS_LSTM_WIDE_KPIPE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_pipeline.py -m gpu -x -q -k "rnn or lstm or cfg2 or ds2 or shard or utter or wide or paired" 2>&1 | tail -2
for r in 1 2; do
echo -n "N=32 plain: "; PROBE_N=32 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=32 kpipe: "; MS_LSTM_WIDE_KPIPE=1 PROBE_N=32 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 plain: "; PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=64 kpipe: "; MS_LSTM_WIDE_KPIPE=1 PROBE_N=64 python tools/lstm_layer_time.py 2>&1 | tail -1
done
echo "== stamps, kpipe"
MS_LSTM_WIDE_KPIPE=1 python tools/wide_stamps.py 2>&1 | grep -v amdgpu.ids
