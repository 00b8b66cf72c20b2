#!/usr/bin/env python3
"""Diagnostic: MaskConv1d shapes (the conv1d flavour of the DS2 builder) on [32, C, 1001]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myrtlespeech_amd.model.cnn import MaskConv1d, PaddingMode
torch.manual_seed(0)
for cin, cout, k, s in ((80, 512, 11, 2), (512, 512, 11, 1), (80, 4, 5, 2), (1, 32, 11, 2)):
    m = MaskConv1d(cin, cout, k, s, PaddingMode.SAME).eval()
    x = torch.randn(32, cin, 1001).cuda()
    lens = torch.full((32,), 1001, dtype=torch.int64)
    for _ in range(3): m((x, lens))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y, l = m((x, lens))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    fl = 2 * 32 * cout * y.shape[-1] * cin * k
    print(f"conv1d {cin}->{cout} k{k} s{s}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF")
