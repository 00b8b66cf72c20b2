cd "$(dirname "$0")/.."
echo "== correctness with MS_LSTM_WIDE=1"
MS_LSTM_WIDE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q -k "rnn or lstm or cfg2 or ds2 or shard or utter" 2>&1 | tail -2
for n in 32 64; do for r in 1 2; do
echo -n "N=$n shipped: "; PROBE_N=$n python tools/lstm_layer_time.py 2>&1 | tail -1
echo -n "N=$n wide   : "; MS_LSTM_WIDE=1 PROBE_N=$n python tools/lstm_layer_time.py 2>&1 | tail -1
done; done
