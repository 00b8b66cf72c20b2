#!/usr/bin/env python3
"""Diagnostic (VERDICT r1 item 5 / DESIGN 7.0): can the input-projection GEMM of one batch run in the idle half of the
persistent LSTM recurrence of another batch?

Two full config-2 recurrent stacks (5 x BiLSTM-1024, batch 32 x 501 steps) = two batches in flight:
  seq : batch A then batch B on ONE stream (what bench.py does today);
  par : batch A on stream 1, batch B on stream 2 (the library chains the persistent launches across streams, so the two
        recurrences never overlap each other; what CAN overlap is A's recurrence with B's projection and vice versa).
Each for two forms of the projection GEMM:
  variant 0 : the shipped 256 x 256 tile, 8 waves (2 x 224 VGPRs per SIMD: cannot share a CU with an LSTM workgroup);
  variant 7 : 256 x 128 tile, 4 waves (232 VGPRs, 128 KB LDS): fits on a CU BESIDE an LSTM workgroup (272 VGPRs, 20 KB).
Reports per-batch latency and throughput, and the in-library HIP-event spans of the recurrence and projection launches."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.rnn import RNN, RNNType  # noqa: E402

H, N, T, In, L = 1024, 32, 501, 640, 5
lib = _lib.load()


class Stack:
    """L single-layer modules, so that the host can issue the two batches LAYER BY LAYER (A1 B1 A2 B2 ...): the library
    chains persistent launches in host issue order, and a whole stack per call would put all of B's recurrences behind A's
    last one.  (Outputs travel as float32 between the layers here; the shipped stack hands operand planes over.)"""

    def __init__(self, seed):
        torch.manual_seed(seed)
        self.layers = [RNN(RNNType.LSTM, In if l == 0 else 2 * H, H, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
                       for l in range(L)]
        for m in self.layers:
            m.check_status = False


def stack(seed):
    return Stack(seed)


mA, mB = stack(0), stack(1)
xA = torch.randn(T, N, In, device="cuda")
xB = torch.randn(T, N, In, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "6"))


def run(mode, iters=4):
    """iters x (batch A, batch B); returns (ms per batch pair, mean per-batch latency ms, spans)."""
    ms = (ctypes.c_float * _lib.PROF_KINDS)()      # ms_prof_read writes MS_PROF_KINDS entries (4 here overran the arrays)
    cnt = (ctypes.c_int * _lib.PROF_KINDS)()
    torch.cuda.synchronize()
    lib.ms_prof_enable(1)
    lib.ms_prof_read(ms, cnt)
    lat = []
    t0 = time.perf_counter()
    for _ in range(iters):
        jobs = [[mA, xA, s1, None, None], [mB, xB, s1 if mode == "seq" else s2, None, None]]
        for j in jobs:
            with torch.cuda.stream(j[2]):
                j[3] = torch.cuda.Event(enable_timing=True)
                j[3].record()
        order = [(j, l) for j in jobs for l in range(L)] if mode == "seq" else [(j, l) for l in range(L) for j in jobs]
        for j, l in order:
            with torch.cuda.stream(j[2]):
                (j[1], _), _ = j[0].layers[l]((j[1], lens))
        for j in jobs:
            with torch.cuda.stream(j[2]):
                j[4] = torch.cuda.Event(enable_timing=True)
                j[4].record()
        lat.append([(j[3], j[4]) for j in jobs])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    lib.ms_prof_read(ms, cnt)
    lib.ms_prof_enable(0)
    lats = [a.elapsed_time(b) for evs in lat for a, b in evs]
    return wall, sum(lats) / len(lats), [ms[k] / max(cnt[k], 1) for k in range(4)]


VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "0,7").split(",")]
MODES = os.environ.get("PROBE_MODES", "seq,par").split(",")
for variant in VARIANTS:
    lib.ms_gemm_set_variant(variant)
    for mode in MODES:
        run(mode, 2)
    res = {m: [] for m in MODES}
    for _ in range(ROUNDS):                 # interleaved rounds in one process
        for mode in MODES:
            res[mode].append(run(mode))
    for mode in MODES:
        walls = sorted(r[0] for r in res[mode])
        med = res[mode][[r[0] for r in res[mode]].index(walls[len(walls) // 2])]
        print(f"GEMM variant {variant} {mode}: {med[0]:.2f} ms per PAIR of batches (min {walls[0]:.2f}, max {walls[-1]:.2f}) = "
              f"{med[0] / 2:.2f} ms per batch; per-batch latency {med[1]:.2f} ms; HIP-event spans: recurrence {med[2][1]:.3f} ms / layer, "
              f"projection {med[2][0]:.3f} ms / layer (GEMM alone at K = 2048: {med[2][2]:.3f} ms)")
lib.ms_gemm_set_variant(0)
for st in (mA, mB):
    for m in st.layers:
        _lib.check(lib.ms_rnn_status(_lib.ptr(m._workspace.buf), _lib.stream_ptr()), "persistent LSTM")
print("status ok")
