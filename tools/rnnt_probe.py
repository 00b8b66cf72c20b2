#!/usr/bin/env python3
"""Diagnostic: RNN-T beam-8 decode alone at the config-4 shape (batch 16, 501 frames, 2-layer LSTM-1024 predictor)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor  # noqa: E402
from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder  # noqa: E402

torch.manual_seed(0)
V = 28
pred = RNNTPredictor(V, 256, 1024, num_layers=2).eval()
joint = RNNTJoint(1024, 1024, 512, V).eval()
enc = torch.randn(501, 16, 1024, device="cuda")
lens = torch.full((16,), 501, dtype=torch.int64)
mode = sys.argv[1] if len(sys.argv) > 1 else "beam"
dec = RNNTBeamDecoder(pred, joint, 8, 3) if mode == "beam" else RNNTGreedyDecoder(pred, joint, 3)
dec(enc, lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = dec(enc, lens)
torch.cuda.synchronize()
print(f"{mode} decode {1e3 * (time.perf_counter() - t0):.1f} ms, mean hypothesis length {sum(map(len, out)) / len(out):.1f}")
