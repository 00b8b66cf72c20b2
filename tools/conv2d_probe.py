#!/usr/bin/env python3
"""Diagnostic: the two config-2 convolutions alone (conv1 1 -> 32, 41 x 11 / 2 x 2; conv2 32 -> 32, 21 x 11 / 2 x 1), batch 32 x 1001
frames.  MS_CONV_FWIN=0 sends conv1 back to the exact-f32 tap kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode  # noqa: E402

torch.manual_seed(0)
c1 = MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME).eval()
c2 = MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME).eval()
x = torch.randn(32, 1, 80, 1001, device="cuda")
lens = torch.full((32,), 1001, dtype=torch.int64)
for name, conv, inp in (("conv1", c1, (x, lens)), ("conv2", c2, None)):
    if inp is None:
        inp = c1((x, lens), fused_activation=(0.0, 20.0))
    for _ in range(3):
        conv(inp, fused_activation=(0.0, 20.0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv(inp, fused_activation=(0.0, 20.0))
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms per call")
