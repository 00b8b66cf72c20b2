#!/usr/bin/env python3
"""Diagnostic: one config-2 BiLSTM layer (H=1024, N=32, T=501) with in-kernel wall-clock
stamps (MS_LSTM_STAMPS=1): where does a time step go?  Run on the GPU box."""
import os
import sys
import time

os.environ.setdefault("MS_LSTM_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T, In = 1024, 32, 501, int(os.environ.get("PROBE_IN", "2048"))
torch.manual_seed(0)
m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
x = torch.randn(T, N, In, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
lib = _lib.load()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m((x, lens))
    torch.cuda.synchronize()
    print(f"layer wall {1e3 * (time.perf_counter() - t0):.3f} ms")
off = lib.ms_rnn_debug_offset(0, T, N, In, H, 2)
ws = m._workspace.buf
dbg = ws[off:off + 256 * 8 * 8].view(torch.int64).reshape(256, 8).cpu().double()
two_stream = os.environ.get("MS_LSTM_ONE_STREAM") != "1" and os.environ.get("MS_PRECISION") != "f32"
if two_stream:
    # two-stream kernel: slots 0-3 = cell waves (0, 1), slots 4-7 = waves 2, 3; each slot sums 2 waves
    names = ["wait h (tags)", "re-requests", "mfma+lds write+barrier", "cell+publish"]
    for grp, label in ((0, "waves 0-1"), (4, "waves 2-3")):
        per = dbg[:, grp:grp + 4] / 2.0 / T * 10.0
        per[:, 1] = dbg[:, grp + 1] / 2.0 / T / 2.0          # slot 1 counts failed tag checks: per wave and stream-step
        print(label)
        for k, nm in enumerate(names):
            c = per[:, k]
            unit = "per stream-step" if k == 1 else "ns"
            print(f"  {nm:24s} mean {c.mean():8.2f} {unit}  min {c.min():8.2f}  max {c.max():8.2f}")
        print(f"  sum mean {(per[:, 0] + per[:, 2] + per[:, 3]).mean():.1f} ns per step (both streams)")
else:
    per_wave = dbg[:, :4] / 4.0 / T * 10.0  # ns per step (4 waves add up; 100 MHz ticks = 10 ns)
    names = ["wait_flags", "mfma_loop", "reduce+cell", "publish"]
    for k, nm in enumerate(names):
        c = per_wave[:, k]
        print(f"{nm:12s} mean {c.mean():8.1f} ns  min {c.min():8.1f}  max {c.max():8.1f}")
    print(f"sum mean {per_wave.sum(1).mean():.1f} ns per step")
