#!/usr/bin/env python3
"""Where does the persistent LSTM's time step go when a projection GEMM shares its CUs?  In-kernel wall-clock stamps
(MS_LSTM_STAMPS=1) of one config-2 BiLSTM layer, alone and with the 4-wave co-tenant GEMM running on a second stream."""
import os
import sys
import time

os.environ["MS_LSTM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T, In = 1024, 32, 501, 2048
torch.manual_seed(0)
lib = _lib.load()
m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
m.check_status = False
x = torch.randn(T, N, In, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
M, K, NN = T * N, In, 8192
xa = torch.randn(M, K, device="cuda")
w = torch.randn(NN, K, device="cuda") * 0.02
y = torch.empty(M, NN, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
names = ["wait h (tags)", "repeated requests", "mfma+lds write+barrier", "cell+publish"]


def stamps():
    off = lib.ms_rnn_debug_offset(0, T, N, In, H, 2)
    buf = m._workspace.buf
    dbg = buf[off:off + 256 * 8 * 8].view(torch.int64).reshape(256, 8).cpu().double()
    out = {}
    for grp, label in ((0, "cell waves 0-1"), (4, "waves 2-3")):
        per = dbg[:, grp:grp + 4] / 2.0 / T * 10.0      # ns per step (both streams), 100 MHz ticks
        per[:, 1] = dbg[:, grp + 1] / 2.0 / T / 2.0     # slot 1 counts failed tag checks: per wave and stream-step
        out[label] = [float(per[:, k].mean()) for k in range(4)] + [float((per[:, 0] + per[:, 2] + per[:, 3]).mean())]
    return out


for variant, label in ((0, "alone"), (7, "with the 4-wave co-tenant GEMM, 3 DMA pieces at the head of a step (variant 7, round 2) on a second stream"),
                       (10, "with the 4-wave co-tenant GEMM, one DMA piece per MFMA group (variant 10, round 3) on a second stream"),
                       (2, "with the 8-wave register-staged GEMM (cannot be co-resident) on a second stream")):
    lib.ms_gemm_set_variant(variant if variant else 0)
    for it in range(3):
        torch.cuda.synchronize()
        if variant:
            with torch.cuda.stream(side):
                for _ in range(4):
                    _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), M, K, NN, 0, 0.0, 0.0,
                                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")
        t0 = time.perf_counter()
        m((x, lens))
        torch.cuda.current_stream().synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
    s = stamps()
    print(f"{label}: layer call (projection + recurrence) {wall:.3f} ms")
    for grp, v in s.items():
        print(f"   {grp:15s} " + "  ".join(f"{n} {t:7.2f}" + (" per stream-step" if k == 1 else " ns") for k, (n, t) in enumerate(zip(names, v[:4]))) + f"   sum {v[4]:7.1f} ns per step")
lib.ms_gemm_set_variant(0)
