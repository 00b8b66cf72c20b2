cd "$(dirname "$0")/.."
rocm-smi --showpower --showmaxpower 2>&1 | grep -i "power\|cap" | head -8
(for i in $(seq 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Average Graphics Package Power\|Current Socket\|sclk" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/r03o_power_trace.txt &
SM=$!
sleep 1
PROBE_MODES=seq PROBE_ROUNDS=3 PROBE_STEPS=60 python tools/pipeline_probe.py 2>&1 | tail -1
sleep 1
PROBE_MODES=par PROBE_ROUNDS=3 PROBE_STEPS=60 python tools/pipeline_probe.py 2>&1 | tail -1
wait $SM
