#!/usr/bin/env python3
"""Diagnostic (not product path): what a plain library fp16 GEMM (hipBLASLt through torch.matmul) reaches on this box at the
config-2 projection's shape and MFMA count, beside the library's own split GEMM, alternating in ONE process.  The f16x3 projection
executes 3 x (2 M N K) flops: the yardstick is a plain fp16 GEMM of depth 3K (same MFMA count, no split bookkeeping) and one of
depth K (x 3).  Says whether the distance of `gemm_nt_bf16x3_kernel4` to the nominal peak is the chip (power cap) or the kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402

lib = _lib.load()
M, K, N = 16032, int(os.environ.get("PROBE_K", "2048")), 8192
rounds = int(os.environ.get("PROBE_ROUNDS", "5"))
torch.manual_seed(0)
x = torch.tanh(torch.randn(M, K, device="cuda")) * torch.sigmoid(torch.randn(M, K, device="cuda"))
w = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
packed = torch.empty(lib.ms_linear_split_packed_bytes(K, N), dtype=torch.uint8, device="cuda")
_lib.check(lib.ms_linear_split_pack(_lib.ptr(w), _lib.ptr(packed), K, N, _lib.stream_ptr()), "pack")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")

xh, wh = x.half(), w.half()
x3 = torch.cat([xh, xh, xh], 1).contiguous()
w3 = torch.cat([wh, wh, wh], 1).contiguous()
yh = torch.empty(M, N, device="cuda", dtype=torch.half)
yf = torch.empty(M, N, device="cuda")


def ours():
    _lib.check(lib.ms_linear_split_forward_packed(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(y), M, K, N, 0, 0.0, 0.0,
                                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "gemm")


def lib_k():
    torch.matmul(xh, wh.T, out=yh)


def lib_3k():
    torch.matmul(x3, w3.T, out=yh)


legs = {"ours f16x3 (split of x + GEMM, 3 MFMA passes)": (ours, 3), "hipBLASLt fp16 K": (lib_k, 1), "hipBLASLt fp16 3K": (lib_3k, 3)}
for f, _ in legs.values():
    for _ in range(3):
        f()
torch.cuda.synchronize()
times = {k: [] for k in legs}
for r in range(rounds):
    for k, (f, _) in legs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        times[k].append((time.perf_counter() - t0) / 5 * 1e3)
for k, (f, passes) in legs.items():
    t = sorted(times[k])
    med = t[len(t) // 2]
    print(f"{k}: median {med:.3f} ms (min {t[0]:.3f}, max {t[-1]:.3f}) = {passes * 2 * M * K * N / (med * 1e-3) / 1e15:.3f} PF/s executed", flush=True)
