#!/usr/bin/env python3
"""configs[3] greedy decode only (tools/rnnt_cfg4_time.py without the beam leg): the command rocprofv3 profiles.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/rnnt_greedy_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import gen_rnnt_cfg4 as G  # noqa: E402
from myrtlespeech_amd.post_process.rnnt_decoder import RNNTGreedyDecoder  # noqa: E402

pred, joint = G.parts()
enc, lens = G.inputs()
enc = enc.cuda()
dec = RNNTGreedyDecoder(pred, joint, max_symbols=int(os.environ.get("PROBE_MAX_SYMBOLS", "3")))
dec(enc, lens)
ts = []
for _ in range(int(os.environ.get("PROBE_REPS", "3"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = dec(enc, lens)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"greedy: {min(ts):.1f} ms, labels per utterance {min(map(len, out))} .. {max(map(len, out))}")
