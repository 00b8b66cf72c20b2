#!/usr/bin/env python3
"""The bench network on a batch of 64 (= two batches of 32 handled as two 32-row groups of one forward): ms per forward and
per 32 utterances, greedy decode included.  MS_LSTM_WIDE=1 runs the two groups' recurrences side by side in ONE launch of the
wide-workgroup kernel; without it they run one after the other."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder  # noqa: E402

N = int(os.environ.get("PROBE_N", "64"))
K = int(os.environ.get("PROBE_STEPS", "30"))
model = bench.build_model()
model.rnn.check_status = False
dec = CTCGreedyDecoder(28)
g = torch.Generator().manual_seed(1234)
x = torch.randn(N, 1, 80, 1001, generator=g).cuda()
lens = torch.full((N,), 1001, dtype=torch.int64)
for _ in range(3):
    (y, ol), _ = model((x, lens))
    dec(y, ol)
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for _ in range(K):
        (y, ol), _ = model((x, lens))
        cur = dec.launch(y, ol)
        if prev is not None:
            prev.result()
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / K * 1e3)
ts.sort()
print(f"N={N} wide={os.environ.get('MS_LSTM_WIDE', '0')}: median {ts[len(ts) // 2]:.3f} ms per forward = {ts[len(ts) // 2] * 32 / N:.3f} ms per 32 utterances "
      f"= {N * 10.0 / ts[len(ts) // 2] * 1e3:.0f} audio-s/s")
