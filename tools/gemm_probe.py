#!/usr/bin/env python3
"""Diagnostic: the config-2 input-projection GEMM (M=16032, K=2048 | PROBE_K, N=8192) alone, split-bf16, every kernel
variant interleaved in ONE process (cdna_hip_programming.md rule 24): PROBE_VARIANTS="2,0" (2 = register-staged kernel2, 0 = shipped LDS-DMA kernel4), PROBE_ROUNDS=5.
PROBE_DATA=lstm uses operands shaped like LSTM outputs (tanh-bounded activations), default randn."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402

lib = _lib.load()
M, K, N = 16032, int(os.environ.get("PROBE_K", "2048")), 8192
variants = [int(v) for v in os.environ.get("PROBE_VARIANTS", "2,0").split(",")]
rounds = int(os.environ.get("PROBE_ROUNDS", "5"))
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda")
if os.environ.get("PROBE_DATA") == "lstm":
    x = torch.tanh(x) * torch.sigmoid(torch.randn(M, K, device="cuda"))
w = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda")
ys = {v: torch.empty(M, N, device="cuda") for v in variants}
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
# operand planes are made once (ms_linear_split_forward re-splits per call; the GEMM alone is timed through the events)


def run(v):
    lib.ms_gemm_set_variant(v)
    _lib.check(lib.ms_linear_split_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(ys[v]), M, K, N, 0, 0.0, 0.0,
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")


for v in variants:
    for _ in range(2):
        run(v)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run(v)
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) / 5 * 1e3)
ref = (x[:64].double() @ w.double().T + b.double()).float()
for v in variants:
    t = sorted(times[v])
    med = t[len(t) // 2]
    print(f"variant {v}: split + GEMM {M}x{K}x{N}: median {med:.3f} ms (min {t[0]:.3f}, max {t[-1]:.3f}) incl. ~0.1 ms of operand "
          f"splitting = {6 * M * K * N / (med * 1e-3) / 1e15:.3f} PF bf16; max abs err on 64 rows {float((ys[v][:64] - ref).abs().max()):.3e}")
if len(variants) > 1:
    a, bb = ys[variants[0]], ys[variants[1]]
    print(f"max |y{variants[0]} - y{variants[1]}| = {float((a - bb).abs().max()):.3e}")
lib.ms_gemm_set_variant(0)
