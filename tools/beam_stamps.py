#!/usr/bin/env python3
"""Where a frame of the CTC prefix beam search goes: MS_BEAM_STAMPS=1 makes thread 0 of utterance 0 accumulate the 100 MHz
wall clock per barrier-separated phase of the frame loop (csrc/beam.hip) into the workspace header; this prints them.
    MS_BEAM_STAMPS=1 python tools/beam_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MS_BEAM_STAMPS"] = "1"
import torch  # noqa: E402

from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder  # noqa: E402

g = torch.Generator().manual_seed(0)
probs = torch.softmax(torch.randn(501, 32, 29, generator=g) * 12, dim=2).cuda()
lens = torch.full((32,), 501, dtype=torch.int64)
dec = CTCBeamDecoder(blank_index=28, beam_width=8)
for _ in range(3):
    dec(probs, lens)
torch.cuda.synchronize()
hdr = dec._workspace.buf[:64].view(torch.int32).cpu().tolist()
names = ["top (row -> LDS)", "S1 extensions", "S2 beam entries", "S3 compaction", "S4a rank counts", "S4b select", "S5 nodes", "S6 rows + beam"]
ticks = hdr[4:4 + len(names)]
tot = sum(ticks)
print(f"utterance 0, 501 frames: {tot * 0.01:.1f} us in the frame loop = {tot * 0.01 / 501:.2f} us per frame")
for nm, t in zip(names, ticks):
    print(f"  {nm:18s} {t * 0.01 / 501 * 1e3:7.1f} ns per frame  {100.0 * t / max(tot, 1):5.1f} %")
