#!/usr/bin/env python3
"""Where does a stream-step of the wide-workgroup LSTM kernel go?  In-kernel wall-clock stamps (MS_LSTM_STAMPS=1, the
diagnostic instantiation of lstm_persistent_wide2_kernel) of one config-2 BiLSTM layer: one batch group alone, one group
beside the regular projection GEMM on a second stream, two groups.  ns per stream-step (16 rows), mean over workgroups."""
import os
import sys
import time

os.environ["MS_LSTM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, T, In = 1024, 501, 2048
torch.manual_seed(0)
lib = _lib.load()
M, K, NN = T * 32, In, 8192
xa = torch.randn(M, K, device="cuda")
w = torch.randn(NN, K, device="cuda") * 0.02
y = torch.empty(M, NN, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
names = ["wait h", "re-requests", "mfma+lds write", "barrier", "cell+publish"]


def run(N, gemm):
    m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
    m.check_status = False
    x = torch.randn(T, N, In, device="cuda")
    lens = torch.full((N,), T, dtype=torch.int64)
    wall = 0.0
    for it in range(3):
        torch.cuda.synchronize()
        if gemm:
            with torch.cuda.stream(side):
                for _ in range(4):
                    _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), M, K, NN, 0, 0.0, 0.0,
                                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")
        t0 = time.perf_counter()
        m((x, lens))
        torch.cuda.current_stream().synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
    ns = 2                                  # 16-row streams per workgroup
    nwg = 128 * ((N + 31) // 32)
    off = lib.ms_rnn_debug_offset(0, T, N, In, H, 2)
    dbg = m._workspace.buf[off:off + nwg * 16 * 8].view(torch.int64).reshape(nwg, 16).cpu().double()
    print(f"N = {N}{' beside the regular GEMM' if gemm else ''}: layer call (projection + recurrence) {wall:.3f} ms, {nwg} workgroups x {ns} streams")
    for base, label in ((0, "cell waves 0-3 (3 k-steps)"), (8, "waves 4-7 (5 k-steps)")):
        per = dbg[:, base:base + 5] / 4.0 / (ns * T) * 10.0      # 4 waves per class; ns per stream-step (100 MHz ticks)
        per[:, 1] = dbg[:, base + 1] / 4.0 / (ns * T)            # a count per wave and stream-step
        tot = per[:, 0] + per[:, 2] + per[:, 3] + per[:, 4]
        print(f"  {label}: " + ", ".join(f"{n} {float(per[:, k].mean()):.{2 if k == 1 else 0}f}" + ("" if k == 1 else " ns") for k, n in enumerate(names))
              + f" | sum {float(tot.mean()):.0f} ns per stream-step (min {float(tot.min()):.0f}, max {float(tot.max()):.0f})")


run(32, False)
run(32, True)
run(64, False)
