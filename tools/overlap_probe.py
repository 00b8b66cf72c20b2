#!/usr/bin/env python3
"""Diagnostic: does an input-projection GEMM co-run with the persistent LSTM recurrence of
another batch (two HIP streams)?  Times A = LSTM layers alone, B = GEMMs alone, C = both."""
import os
import sys
import time

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
# a stand-alone GEMM of the same shape as the projection
M, K, NN = T * N, In, 8192
xa = torch.randn(M, K, device="cuda")
w = torch.randn(NN, K, device="cuda") * 0.02
y = torch.empty(M, NN, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def lstm(k):
    with torch.cuda.stream(s1):
        for _ in range(k):
            m((x, lens))


def gemm(k):
    with torch.cuda.stream(s2):
        for _ in range(k):
            _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), M, K, NN, 0, 0.0, 0.0,
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


lstm(2); gemm(2)
k = 10
a = timed(lambda: lstm(k))
b = timed(lambda: gemm(2 * k))
c = timed(lambda: (lstm(k), gemm(2 * k)))
print(f"LSTM layer (incl. its own projection) x{k}: {a:.2f} ms; GEMM x{2*k}: {b:.2f} ms; both concurrently: {c:.2f} ms; "
      f"sum {a + b:.2f} ms")
