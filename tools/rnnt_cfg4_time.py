#!/usr/bin/env python3
"""RNN-T decode at BASELINE.json configs[3] (batch 16, T = 501, 2 x LSTM-1024 predictor, joint 512, beam 8) with the network
and inputs of tests/golden/gen_rnnt_cfg4.py (transcripts of 15 .. 49 labels): greedy and beam decode times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import gen_rnnt_cfg4 as G  # noqa: E402
from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder  # noqa: E402

pred, joint = G.parts()
enc, lens = G.inputs()
enc = enc.cuda()
ms = int(os.environ.get("PROBE_MAX_SYMBOLS", "3"))
for name, dec in (("greedy", RNNTGreedyDecoder(pred, joint, max_symbols=ms)), ("beam-8", RNNTBeamDecoder(pred, joint, beam_width=8, max_symbols=ms))):
    dec(enc, lens)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dec(enc, lens)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name}: {min(ts):.1f} ms (max_symbols {ms}), labels per utterance {min(map(len, out))} .. {max(map(len, out))}")

# the same network with the blank logit raised, so that the GREEDY transcripts are as sparse as a trained transducer's
# (a label every ~10 .. 30 frames): what the event-driven greedy decode is built for
blank = joint.out.bias.shape[0] - 1
for bias in (float(b) for b in os.environ.get("PROBE_BLANK_BIAS", "4,8,12").split(",")):
    with torch.no_grad():
        joint.out.bias[blank] += bias
    dec = RNNTGreedyDecoder(pred, joint, max_symbols=ms)
    dec(enc, lens)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dec(enc, lens)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"greedy, blank logit +{bias:g}: {min(ts):.1f} ms, labels per utterance {min(map(len, out))} .. {max(map(len, out))}")
    with torch.no_grad():
        joint.out.bias[blank] -= bias
