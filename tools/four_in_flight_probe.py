#!/usr/bin/env python3
"""Diagnostic: FOUR batches on the chip -- two merged batches of 64 utterances (each forward = PairedBatches' merged batch: its
recurrence fills all 256 CUs as two groups of the wide kernel) in flight on two streams, the other forward's projection GEMM
beside the recurrence as the regular 8-wave kernel (variant 0) or the 4-wave co-tenant form (variant 10) -- against the same
merged batches one at a time.  ms per batch of 32, interleaved rounds in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from myrtlespeech_amd import _lib, pipeline  # noqa: E402
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder  # noqa: E402

K = int(os.environ.get("PROBE_STEPS", "20"))            # merged batches per region
VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "0,10").split(",")]
lib = _lib.load()
model = bench.build_model()
model.rnn.check_status = False
dec = CTCGreedyDecoder(28)
g = torch.Generator().manual_seed(1234)
x = torch.randn(64, 1, 80, 1001, generator=g).cuda()
lens = torch.full((64,), 1001, dtype=torch.int64)
pipe = pipeline.TwoBatchesInFlight(model, post=lambda out: dec.launch(out[0][0], out[0][1]))
legs = ["serial"] + [f"in_flight_v{v}" for v in VARIANTS]
res = {k: [] for k in legs}


def serial():
    for _ in range(K):
        (y, ol), _ = model((x, lens))
        dec.launch(y, ol)


for rnd in range(int(os.environ.get("PROBE_ROUNDS", "3")) + 1):
    for leg in legs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if leg == "serial":
            serial()
        else:
            v = int(leg.split("v")[-1])
            pipe._gemm_variant = lambda lib_, batches, v=v: v
            for p in pipe([(x, lens)] * K):
                p.result()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (2 * K) * 1e3
        if rnd:
            res[leg].append(dt)
        print(f"round {rnd} {leg}: {dt:.3f} ms per batch of 32", flush=True)
pipe.check_status()
_lib.check(lib.ms_rnn_status(_lib.ptr(model.rnn._workspace.buf), _lib.stream_ptr()), "status")
for leg in legs:
    t = sorted(res[leg])
    print(f"{leg}: median {t[len(t) // 2]:.3f} ms per batch of 32 (min {t[0]:.3f}, max {t[-1]:.3f})")
