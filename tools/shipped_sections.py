#!/usr/bin/env python3
"""Diagnostic: host-side time per stage of the shipped DS2 config (sync after each stage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myrtlespeech_amd import protos as P
from myrtlespeech_amd.builders.speech_to_text import build as build_stt
from tests.test_builders_cpu import DS2_EN
torch.manual_seed(0)
stt = build_stt(P.parse(DS2_EN, P.SpeechToText)).eval()
m = stt.model
m.rnn.check_status = False
N = 32
x = torch.randn(N, 1, 80, 1001).cuda(); lens = torch.full((N,), 1001, dtype=torch.int64)
for _ in range(2): m((x, lens))
torch.cuda.synchronize()
def tick(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print(f"{label:14s} {1e3*(t1-t0):7.2f} ms"); return t1
t0 = time.perf_counter()
h, l = m.cnn((x.clone(), lens)); t0 = tick("cnn", t0)
h = m._conv_to_rnn_size(h) if hasattr(m, "_conv_to_rnn_size") else h; t0 = tick("to_rnn", t0)
(h, l), hid = m.rnn((h, l)); t0 = tick("rnn", t0)
t1 = time.perf_counter()
(y, ol), _ = m((x, lens)); tick("whole forward", t1)
t1 = time.perf_counter()
(y, ol), _ = m((x, lens))
print(f"enqueue only   {1e3*(time.perf_counter()-t1):7.2f} ms"); tick("  + drain", t1)
