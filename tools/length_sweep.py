"""A net for clip lengths nobody benchmarks: the config-2 network (+ greedy decode), the CTC loss and the prefix beam search at
batch 32 for clips of 1 .. 60 s -- ms per call and audio-s/s, so that a length that falls off a fast path shows."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

import bench
from tools.op_audit import timed


def main():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    torch.manual_seed(0)
    N = 32
    frames = [int(v) for v in os.environ.get("SWEEP_T", "101,301,501,1001,2001,3001,6001").split(",")]
    model = bench.build_model()
    greedy = CTCGreedyDecoder(28)
    beam = CTCBeamDecoder(blank_index=28, beam_width=8, prune_threshold=1e-3)
    loss = CTCLoss(blank=28, reduction="sum")
    with torch.no_grad():
        for T in frames:
            x = torch.randn(N, 1, 80, T, device="cuda")
            lens = torch.full((N,), T, dtype=torch.int64)
            # (warm = 4: the first calls at a new length also grow torch's caching allocator -- 8 ms per call over the first
            # seven calls at 10 s after three shorter lengths; not the library's time)
            ms = timed(lambda: model((x, lens)), warm=4)
            (y, ol), _ = model((x, lens))
            secs = N * T / 100.0
            g = timed(lambda: greedy(y, ol))
            # (peaked synthetic posteriors, as in bench.py's beam leg: on a default-init network's near-uniform ones the
            # reference's linear-space float32 search underflows to an empty beam after ~30 frames and the kernel exits)
            probs = torch.softmax(torch.randn(y.shape, device="cuda") * 12.0, dim=-1)
            b = timed(lambda: beam(probs, ol), warm=1, it=2)
            S = max(1, min(120, y.shape[0] // 4))
            tg = torch.randint(0, 28, (N, S))
            tl = torch.full((N,), S, dtype=torch.int64)
            c = timed(lambda: loss((y, ol), (tg, tl)))
            print(f"{T / 100.0:5.1f} s clips x {N}: network {ms:8.3f} ms = {secs / (ms * 1e-3):8.0f} audio-s/s ({ms / y.shape[0] * 1e3:6.1f} us per output frame); "
                  f"greedy {g:6.3f} ms, beam-8 {b:7.3f} ms ({b / y.shape[0] * 1e3:5.2f} us per frame), CTC loss (S = {S}) {c:6.3f} ms", flush=True)


if __name__ == "__main__":
    main()
