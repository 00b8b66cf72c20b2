#!/usr/bin/env python3
"""Two batches in flight on the bench network with different co-tenant forms of the projection GEMM (ms_gemm_set_variant),
interleaved rounds in one process: ms per batch, and the in-library spans of the recurrence and of the K = 2048 GEMM."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from myrtlespeech_amd import _lib, pipeline  # noqa: E402
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder  # noqa: E402

K = int(os.environ.get("PROBE_STEPS", "40"))
VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "7,8,9").split(",")]
lib = _lib.load()
model = bench.build_model()
model.rnn.check_status = False
dec = CTCGreedyDecoder(28)
g = torch.Generator().manual_seed(1234)
x = torch.randn(32, 1, 80, 1001, generator=g).cuda()
lens = torch.full((32,), 1001, dtype=torch.int64)
pipe = pipeline.TwoBatchesInFlight(model, post=lambda out: dec.launch(out[0][0], out[0][1]))
res = {v: [] for v in VARIANTS}
spans = {v: None for v in VARIANTS}
ms = (ctypes.c_float * 9)()
cnt = (ctypes.c_int * 9)()
for rnd in range(int(os.environ.get("PROBE_ROUNDS", "4")) + 1):
    for v in VARIANTS:
        pipeline.COTENANT_GEMM_VARIANT = v
        lib.ms_prof_enable(1)
        lib.ms_prof_read(ms, cnt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pipe([(x, lens)] * K):
            p.result()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        lib.ms_prof_read(ms, cnt)
        lib.ms_prof_enable(0)
        if rnd:                                   # round 0 = warm-up
            res[v].append(dt)
            spans[v] = (ms[1] / max(cnt[1], 1), ms[2] / max(cnt[2], 1))
pipe.check_status()
for v in VARIANTS:
    t = sorted(res[v])
    print(f"variant {v}: median {t[len(t) // 2]:.3f} ms per batch (min {t[0]:.3f}, max {t[-1]:.3f}); recurrence {spans[v][0]:.3f} ms, "
          f"K=2048 GEMM {spans[v][1]:.3f} ms under co-tenancy")
