#!/usr/bin/env python3
"""Throughput and latency of myrtlespeech_amd.pipeline.TwoBatchesInFlight on the bench network (config 2) against the
one-batch-at-a-time path, interleaved rounds in one process, greedy decode included in both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from myrtlespeech_amd.pipeline import BatchesInFlight  # noqa: E402
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder  # noqa: E402

K = int(os.environ.get("PROBE_STEPS", "40"))
model = bench.build_model()
model.rnn.check_status = False
dec = CTCGreedyDecoder(28)
g = torch.Generator().manual_seed(1234)
x = torch.randn(32, 1, 80, 1001, generator=g).cuda()
lens = torch.full((32,), 1001, dtype=torch.int64)
pipe = BatchesInFlight(model, post=lambda out: dec.launch(out[0][0], out[0][1]), depth=int(os.environ.get("PROBE_DEPTH", "2")))


def seq():
    for _ in range(K):
        (y, ol), _ = model((x, lens))
        dec(y, ol)


def par():
    for p in pipe([(x, lens)] * K):
        p.result()


MODES = os.environ.get("PROBE_MODES", "seq,par").split(",")
fns = [(n, f) for n, f in (("seq", seq), ("par", par)) if n in MODES]
for _, fn in fns:
    fn()
res = {"seq": [], "par": []}
for _ in range(int(os.environ.get("PROBE_ROUNDS", "5"))):
    for name, fn in fns:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / K * 1e3)
pipe.check_status()
for name, _ in fns:
    t = sorted(res[name])
    print(f"{name}: median {t[len(t) // 2]:.3f} ms per batch (min {t[0]:.3f}, max {t[-1]:.3f}) = {320.0 / t[len(t) // 2] * 1e3:.0f} audio-s/s")
