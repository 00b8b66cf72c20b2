#!/usr/bin/env python3
"""Diagnostic: the config-2 input-projection GEMM (M=16032, K=2048, N=8192) alone, split-bf16."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402

lib = _lib.load()
M, K, N = 16032, int(os.environ.get("PROBE_K", "2048")), 8192
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")


def run():
    _lib.check(lib.ms_linear_split_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, 0, 0.0, 0.0,
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"split GEMM {M}x{K}x{N}: {dt * 1e3:.3f} ms  = {2 * M * K * N / dt / 1e12:.1f} TF f32-equivalent, "
      f"{6 * M * K * N / dt / 1e15:.3f} PF bf16")
ref = (x[:64].double() @ w.double().T + b.double()).float()
print("max abs err on 64 rows:", float((y[:64] - ref).abs().max()))
