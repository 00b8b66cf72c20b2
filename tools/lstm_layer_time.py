#!/usr/bin/env python3
"""One config-2 BiLSTM layer (H = 1024, N = 32, T = 501, In = 2048) timed with the library's own HIP-event spans: mean
projection and recurrence time per call.  For same-box A/B runs of two library builds (tools/ab_lib.sh)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T, In = 1024, int(os.environ.get("PROBE_N", "32")), 501, int(os.environ.get("PROBE_IN", "2048"))
REPS = int(os.environ.get("PROBE_REPS", "40"))
torch.manual_seed(0)
lib = _lib.load()
m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
m.check_status = False
x = torch.randn(T, N, In, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
if os.environ.get("PROBE_RAGGED") == "1":   # lengths ~U[T / 2, T], sorted (the bench's ragged workload)
    lens = torch.sort(torch.randint(T // 2, T + 1, (N,), generator=torch.Generator().manual_seed(5)), descending=True).values
    lens[0] = T
for _ in range(5):
    m((x, lens))
torch.cuda.synchronize()
ms = (ctypes.c_float * 9)()
cnt = (ctypes.c_int * 9)()
lib.ms_prof_enable(1)
lib.ms_prof_read(ms, cnt)
for _ in range(REPS):
    m((x, lens))
torch.cuda.synchronize()
lib.ms_prof_read(ms, cnt)
lib.ms_prof_enable(0)
_lib.check(lib.ms_rnn_status(_lib.ptr(m._workspace.buf), _lib.stream_ptr()), "status")
print(f"rows {int(lens.sum())} of {T * N}; ", end="")
print(f"{os.path.basename(_lib.LIB_PATH)}: recurrence {ms[1] / max(cnt[1], 1):.4f} ms  projection {ms[0] / max(cnt[0], 1):.4f} ms  "
      f"({cnt[1]} launches)")
