#!/usr/bin/env python3
"""What would the literal batch-32 step cost if layer l+1's input projection ran beside layer l's recurrence?  (VERDICT r5
item 2.)  A TIMING emulation on the real kernels, sizes and streams -- no data flows, nothing here is shipped.

The proposed design: cut a layer's recurrence (501 steps, 128 of the 256 CUs) into S time segments, one launch each; after
segment k the forward direction's rows of time segment k and the backward direction's rows of time segment S-1-k exist, so on
a second stream, behind an event, two half-K GEMMs (K = 1024: the forward / backward half of W_ih) of L*N = 501*32/S rows
each produce the next layer's two partial pre-activation buffers P_f, P_b on the idle CUs; the next layer starts when the
last segment's GEMMs are done.

Emulated with what exists: a segment = one call of the layer entry point over L = 501 / S steps with the projection skipped
(MS_RNN_TIMING_SKIP_PROJECTION), a half-K GEMM = ms_linear_split_forward_packed at M = L*N, K = 1024, N = 8192 with the operand
split skipped (MS_TIMING_SKIP_SPLIT=1), events between two streams exactly as the design would have them.  Each segment call
initialises its exchange (hx_init, ~5 us) and reloads W_hh (33 MB) as a real segment would reload W_hh; the real design would
not re-initialise, so the emulation is slightly pessimistic there and optimistic in nothing.

A = today's stack (5 layers: projection then recurrence, one stream);  B(S) = the overlapped schedule;  C = B's GEMMs alone and
B's segments alone (what each stream would take by itself).

    python tools/overlap_emulation.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model import rnn as R  # noqa: E402

T, N, H = 501, 32, 1024
lib = _lib.load()
torch.manual_seed(0)
stack = R.RNN(R.RNNType.LSTM, 640, H, num_layers=5, bidirectional=True, forget_gate_bias=1.0).eval()
stack.check_status = False
x0 = torch.randn(T, N, 640, device="cuda")
lens = torch.full((N,), T, dtype=torch.int64)


def timed(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


# ---- A: today's stack
ms_a = timed(lambda: stack((x0, lens)))
_lib.check(lib.ms_rnn_status(_lib.ptr(stack._workspace.buf), _lib.stream_ptr()), "stack")

# ---- pieces of B
cell = _lib.CELL_LSTM
params = stack._layer_params()
pk1 = R.PackedLayer().get(cell, 2 * H, H, params[1])          # a layer >= 1 (In = 2048)
pk0 = R.PackedLayer().get(cell, 640, H, params[0])
ws_bytes = max(lib.ms_rnn_workspace_bytes(cell, T, N, 2 * H, H, 2), lib.ms_rnn_workspace_bytes(cell, T, N, 640, H, 2))
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
hn = torch.zeros(2, N, H, device="cuda")
cn = torch.zeros(2, N, H, device="cuda")
w_half = torch.randn(8192, 1024, device="cuda") * 0.02
wp = torch.empty(lib.ms_linear_split_packed_bytes(1024, 8192), dtype=torch.uint8, device="cuda")
_lib.check(lib.ms_linear_split_pack(_lib.ptr(w_half), _lib.ptr(wp), 1024, 8192, _lib.stream_ptr()), "pack")
side = torch.cuda.Stream()
SKIP = 8192   # MS_RNN_TIMING_SKIP_PROJECTION


def segment(L, out):
    lens_dev = None
    _lib.check(lib.ms_rnn_layer_forward_ex(cell, _lib.ptr(pk1), _lib.ptr(xs[:L]), _lib.ptr(lens_dev), L, None, None, _lib.ptr(out[:L]),
                                           _lib.ptr(hn), _lib.ptr(cn), L, N, 2 * H, H, 2, SKIP, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr()), "segment")


xs = torch.randn(T, N, 2 * H, device="cuda")
out = torch.empty(T, N, 2 * H, device="cuda")


def half_gemms(L, pbuf, xrows, gws):
    m = L * N
    for _ in range(2):
        _lib.check(lib.ms_linear_split_forward_packed(_lib.ptr(xrows), _lib.ptr(wp), None, _lib.ptr(pbuf), m, 1024, 8192, 0, 0.0, 0.0,
                                                      _lib.ptr(gws), gws.numel(), _lib.stream_ptr()), "gemm")


def layer0_projection():
    # K = 640 projection of layer 0 (depends on the convolutions: never overlapped), as the layer entry point runs it
    _lib.check(lib.ms_linear_split_forward(_lib.ptr(x0.view(T * N, 640)), _lib.ptr(w0), None, _lib.ptr(p_full), T * N, 640, 8192, 0, 0.0,
                                           0.0, _lib.ptr(gws0), gws0.numel(), _lib.stream_ptr()), "proj0")


w0 = torch.randn(8192, 640, device="cuda") * 0.02
p_full = torch.empty(T * N, 8192, device="cuda")
gws0 = torch.empty(lib.ms_linear_split_workspace_bytes(T * N, 640, 8192), dtype=torch.uint8, device="cuda")

print(f"A  today's stack of 5 layers (projection, then recurrence, one stream): {ms_a:.3f} ms")
os.environ["MS_TIMING_SKIP_SPLIT"] = "1"
for S in (2, 4, 6, 8):
    L = T // S
    m = L * N
    pbuf = torch.empty(m, 8192, device="cuda")
    xrows = torch.randn(m, 1024, device="cuda")
    gws = torch.empty(lib.ms_linear_split_workspace_bytes(m, 1024, 8192), dtype=torch.uint8, device="cuda")

    def schedule(overlap=True, gemms=True, segs=True):
        main = torch.cuda.current_stream()
        layer0_projection()
        for layer in range(5):
            last = None
            for k in range(S):
                if segs:
                    segment(L, out)
                if layer < 4 and gemms:
                    if overlap:
                        e = torch.cuda.Event()
                        e.record(main)
                        with torch.cuda.stream(side):
                            side.wait_event(e)
                            half_gemms(L, pbuf, xrows, gws)
                            last = torch.cuda.Event()
                            last.record(side)
                    else:
                        half_gemms(L, pbuf, xrows, gws)
            if last is not None:
                main.wait_event(last)

    ms_b = timed(schedule)
    ms_serial = timed(lambda: schedule(overlap=False))
    ms_g = timed(lambda: schedule(overlap=False, segs=False))
    ms_s = timed(lambda: schedule(gemms=False))
    print(f"B  S = {S} segments of {L} steps: overlapped {ms_b:.3f} ms | the same launches on one stream {ms_serial:.3f} | "
          f"K-half GEMMs alone {ms_g:.3f} | segment launches alone {ms_s:.3f}   -> overlap vs today {ms_b - ms_a:+.3f} ms")
_lib.check(lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()), "segments")
