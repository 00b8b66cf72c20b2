cd "$(dirname "$0")/.."
for r in 1 2; do
python tools/n64_probe.py 2>&1 | tail -1
MS_LSTM_WIDE=1 python tools/n64_probe.py 2>&1 | tail -1
PROBE_MODES=par PROBE_ROUNDS=3 python tools/pipeline_probe.py 2>&1 | tail -1
done
