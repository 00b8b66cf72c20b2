#!/usr/bin/env python3
"""Which clock does the chip hold while the persistent recurrence runs alone, while the co-tenant projection GEMM runs alone,
and while both share the CUs?  One sampling wave (ms_clock_probe) on a stream of its own records {100 MHz wall ticks, shader
cycles} every 20 us; clock = d(cycles) / d(ticks) x 100 MHz per interval."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T = 1024, 32, 501
torch.manual_seed(0)
lib = _lib.load()
m = RNN(RNNType.LSTM, 32, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
m.check_status = False
x = torch.randn(T, N, 32, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
M, K, NN = T * N, 2048, 8192
xa = torch.randn(M, K, device="cuda")
w = torch.randn(NN, K, device="cuda") * 0.02
y = torch.empty(M, NN, device="cuda")
ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
side, probe = torch.cuda.Stream(), torch.cuda.Stream()
SAMPLES, SPACING = 400, 20          # 8 ms
buf = torch.zeros(2 * SAMPLES, dtype=torch.int64, device="cuda")
variant = int(os.environ.get("PROBE_VARIANT", "10"))
LAYERS = 3                          # back-to-back launches, so that the load lasts ~6 ms


def gemm(v):
    lib.ms_gemm_set_variant(v)
    for _ in range(LAYERS):
        _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), M, K, NN, 0, 0.0, 0.0, _lib.ptr(ws),
                                               ws.numel(), _lib.stream_ptr()), "gemm")
    lib.ms_gemm_set_variant(0)


def run(label, with_rec, gemm_variant):
    for it in range(2):
        torch.cuda.synchronize()
        go = torch.cuda.Event()
        go.record()
        with torch.cuda.stream(probe):
            probe.wait_event(go)
            _lib.check(lib.ms_clock_probe(_lib.ptr(buf), SAMPLES, SPACING, _lib.stream_ptr()), "probe")
        if gemm_variant is not None:
            with torch.cuda.stream(side):
                side.wait_event(go)
                gemm(gemm_variant)
        if with_rec:
            for _ in range(LAYERS):
                m((x, lens))
        torch.cuda.synchronize()
    b = buf.cpu().view(SAMPLES, 2).double()
    dt = (b[1:, 0] - b[:-1, 0])
    clk = (b[1:, 1] - b[:-1, 1]) / dt * 0.1        # GHz
    t_ms = (b[1:, 0] - b[0, 0]) / 1e5
    pts = [(float(t_ms[i]), float(clk[i])) for i in range(0, SAMPLES - 1, 25)]
    print(f"{label}: " + "  ".join(f"{t:4.1f}ms {c:4.2f}GHz" for t, c in pts))
    busy = clk[(t_ms > 0.5) & (t_ms < 4.5)]
    print(f"    mean clock over 0.5 .. 4.5 ms: {float(busy.mean()):.3f} GHz (min {float(busy.min()):.2f}, max {float(busy.max()):.2f})")


run("idle (the probe alone)", False, None)
run("recurrence alone", True, None)
run("8-wave projection GEMM alone (variant 0)", False, 0)
run(f"4-wave co-tenant GEMM alone (variant {variant})", False, variant)
run(f"recurrence + co-tenant GEMM (variant {variant})", True, variant)
