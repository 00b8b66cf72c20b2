"""A net for CTC shapes nobody benchmarks: loss forward (+ backward), greedy and prefix beam search over alphabet sizes, target
lengths and beam widths at [501, 32, V] -- ms per call (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    torch.manual_seed(0)
    T, N = 501, 32
    lens = torch.full((N,), T, dtype=torch.int64)
    for V in (29, 64, 256, 1024, 5000):
        y = torch.randn(T, N, V, device="cuda")
        for S in (20, 120, 250, 400, 600):
            if 2 * S + 1 > T:
                continue
            tg = torch.randint(0, V - 1, (N, S))
            tl = torch.full((N,), S, dtype=torch.int64)
            loss = CTCLoss(blank=V - 1, reduction="sum")
            loss.check_status = False
            with torch.no_grad():
                f = timed(lambda: loss((y, lens), (tg, tl)))
            yg = y.clone().requires_grad_(True)

            def fb():
                yg.grad = None
                loss((yg, lens), (tg, tl)).backward()
            b = timed(fb)
            print(f"CTC loss V={V:5d} S={S:4d}: forward {f:7.3f} ms, forward + backward {b:7.3f} ms", flush=True)
        with torch.no_grad():
            g = timed(lambda: CTCGreedyDecoder(V - 1)(y, lens))
        print(f"greedy V={V:5d}: {g:7.3f} ms", flush=True)
    with torch.no_grad():
        for V in (29, 64, 128):
            probs = torch.softmax(torch.randn(T, N, V, device="cuda") * 12.0, dim=-1)
            for W in (4, 8, 16, 32, 64, 100):
                dec = CTCBeamDecoder(blank_index=V - 1, beam_width=W, prune_threshold=1e-3)
                ms = timed(lambda: dec(probs, lens), warm=1, it=3)
                print(f"beam V={V:4d} W={W:4d}: {ms:8.3f} ms ({ms / T * 1e3:6.2f} us per frame)", flush=True)


if __name__ == "__main__":
    main()
