"""The prefix beam search at [501, 32, 29], peaked synthetic posteriors, widths 4 / 8 / 16 / 32: ms per call (HIP events).  For
tools/ab_lib.sh (AB_TAIL=4)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed
from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder

torch.manual_seed(0)
T, N, V = 501, 32, 29
lens = torch.full((N,), T, dtype=torch.int64)
probs = torch.softmax(torch.randn(T, N, V, device="cuda") * 12.0, dim=-1)
with torch.no_grad():
    for W in (4, 8, 16, 32):
        dec = CTCBeamDecoder(blank_index=V - 1, beam_width=W, prune_threshold=1e-3)
        ms = min(timed(lambda: dec(probs, lens), warm=2, it=5) for _ in range(3))
        print(f"beam V={V} W={W:3d}: {ms:7.3f} ms", flush=True)
