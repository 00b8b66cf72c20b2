#!/usr/bin/env python3
"""Diagnostic: the reference's SHIPPED DS2 config (3 x GRU-2560 unidirectional + lookahead 80) at batch 32 x 10 s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd import protos as P  # noqa: E402
from myrtlespeech_amd.builders.speech_to_text import build as build_stt  # noqa: E402
from tests.test_builders_cpu import DS2_EN  # noqa: E402

torch.manual_seed(0)
stt = build_stt(P.parse(DS2_EN, P.SpeechToText)).eval()
stt.model.rnn.check_status = False   # like bench.py: no per-layer host sync inside the timed region
N = int(os.environ.get("PROBE_N", "32"))
x = torch.randn(N, 1, 80, 1001).cuda()
lens = torch.full((N,), 1001, dtype=torch.int64)
for _ in range(3):
    (y, ol), _ = stt.model((x, lens))
    hyp = stt.post_process(y, ol)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    (y, ol), _ = stt.model((x, lens))
    hyp = stt.post_process(y, ol)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"shipped DS2 (3xGRU-2560 + lookahead 80), batch {N}: {dt * 1e3:.1f} ms = {N * 10 / dt:.0f} audio-s/s")
