#!/usr/bin/env python3
"""The overlapped stack schedule (ms_rnn_stack_forward) against the layer-by-layer one on the config-2 recurrent stack
(5 x BiLSTM-1024, 32 x 501 frames): bit equality and time, same process, same box.
    python tools/overlap_ab.py [segments ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd.model import rnn as R  # noqa: E402

T, N = 501, 32
torch.manual_seed(0)
stack = R.RNN(R.RNNType.LSTM, 640, 1024, num_layers=5, bidirectional=True, forget_gate_bias=1.0).eval()
x = torch.randn(T, N, 640, device="cuda") * 0.3
lens = torch.full((N,), T, dtype=torch.int64)


def timed(fn, warm=5, reps=30):
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


R._OVERLAP = False
(y0, _), (h0, c0) = stack((x, lens))
stack.check_status = False
ms0 = timed(lambda: stack((x, lens)))
print(f"layer by layer: {ms0:.3f} ms per stack call")
for S in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 12, 16]:
    R._OVERLAP, R._OVERLAP_SEGMENTS = True, S
    stack.check_status = True
    (y1, _), (h1, c1) = stack((x, lens))
    eq = torch.equal(y0, y1) and torch.equal(h0, h1) and torch.equal(c0, c1)
    stack.check_status = False
    ms1 = timed(lambda: stack((x, lens)))
    print(f"overlapped, {S:2d} segments: {ms1:.3f} ms ({ms1 - ms0:+.3f})   bits equal: {eq}   max |diff| {float((y0 - y1).abs().max()):.3e}")
