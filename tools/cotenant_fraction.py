#!/usr/bin/env python3
"""How long does the persistent recurrence of a config-2 layer take when a co-tenant projection GEMM covers only a FRACTION of
it?  (What a faster co-tenant would buy: the recurrence runs beside the GEMM for the GEMM's duration and alone afterwards.)
A layer with a 32-wide input (its own projection is ~20 us, the recurrence is that of H = 1024) on the main stream; a K = 2048
split GEMM (co-tenant form) with M = frac x 16032 rows on a second stream, released by an event recorded just before the layer."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T, In = 1024, 32, 501, 32
torch.manual_seed(0)
lib = _lib.load()
m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
m.check_status = False
x = torch.randn(T, N, In, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
M, K, NN = T * N, 2048, 8192
xa = torch.randn(M, K, device="cuda")
w = torch.randn(NN, K, device="cuda") * 0.02
y = torch.empty(M, NN, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
ms = (ctypes.c_float * 9)()
cnt = (ctypes.c_int * 9)()
variant = int(os.environ.get("PROBE_VARIANT", "10"))
m((x, lens))
torch.cuda.synchronize()
print(f"co-tenant GEMM variant {variant}; recurrence of one layer (both directions, 501 steps), GEMM K = 2048, N = 8192")
for frac in (0.0, 0.25, 0.5, 0.75, 1.0, 1.25):
    rec, gem = [], []
    rows = int(M * frac)
    for it in range(7):
        torch.cuda.synchronize()
        lib.ms_prof_enable(1)
        lib.ms_prof_read(ms, cnt)
        go = torch.cuda.Event()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        go.record()
        if rows > 0:
            with torch.cuda.stream(side):
                side.wait_event(go)
                lib.ms_gemm_set_variant(variant)
                g0.record()
                _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), rows, K, NN, 0, 0.0, 0.0,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")
                g1.record()
                lib.ms_gemm_set_variant(0)
        m((x, lens))
        torch.cuda.synchronize()
        lib.ms_prof_read(ms, cnt)
        lib.ms_prof_enable(0)
        if it:
            rec.append(ms[1] / max(cnt[1], 1))
            if rows > 0:
                gem.append(g0.elapsed_time(g1))
    rec.sort(); gem.sort()
    print(f"GEMM rows {rows:6d} ({frac:4.2f} x 16032): recurrence {rec[len(rec) // 2]:.3f} ms"
          + (f", GEMM incl. plane split {gem[len(gem) // 2]:.3f} ms" if gem else " (alone)"))
