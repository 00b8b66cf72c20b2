"""Wall / device time of the public operators at configuration-like sizes -- a net for paths nobody benchmarks (the CTC
backward took 26 ms for years of rounds because no leg timed it).  Prints ms per call (HIP events, mean of 5 after 2)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch


def timed(fn, warm=2, it=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    from myrtlespeech_amd.model.cnn import MaskConv1d, MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    from myrtlespeech_amd.model.lookahead import Lookahead
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    torch.manual_seed(0)
    T, N = 501, 32
    lens = torch.full((N,), T, dtype=torch.int64)
    rows = []
    with torch.no_grad():
        for kind, H, bi, inp in [(RNNType.LSTM, 1024, True, 2048), (RNNType.LSTM, 512, True, 1024), (RNNType.LSTM, 768, False, 768),
                                 (RNNType.LSTM, 320, True, 640), (RNNType.LSTM, 1000, True, 1000), (RNNType.LSTM, 1280, True, 1280),
                                 (RNNType.LSTM, 1536, True, 1536), (RNNType.LSTM, 2048, True, 2048), (RNNType.LSTM, 2048, False, 2048), (RNNType.GRU, 2560, False, 2560),
                                 (RNNType.GRU, 1280, True, 1280), (RNNType.GRU, 800, True, 800), (RNNType.GRU, 1024, True, 1024),
                                 (RNNType.BASIC_RNN, 1024, True, 1024)]:
            m = RNN(kind, inp, H, num_layers=1, bidirectional=bi, forget_gate_bias=1.0 if kind == RNNType.LSTM else None).eval()
            m.check_status = False
            x = torch.randn(T, N, inp, device="cuda")
            rows.append((f"RNN {kind.name} H={H} bi={bi} [{T},{N},{inp}]", timed(lambda: m((x, lens)))))
        hl = HardLSTM(2048, 1024, num_layers=1, bidirectional=True, batch_first=False).cuda().eval()
        x = torch.randn(T, N, 2048, device="cuda")
        rows.append(("HardLSTM 1024 bi", timed(lambda: hl((x, lens)))))
        la = Lookahead(2560, 80).eval()
        x = torch.randn(N, 2560, T, device="cuda")
        rows.append(("Lookahead 2560 ctx 80 [32,2560,501]", timed(lambda: la((x, lens)))))
        c1 = MaskConv1d(512, 512, 11, 1, PaddingMode.SAME).eval()
        x = torch.randn(N, 512, T, device="cuda")
        rows.append(("MaskConv1d 512->512 k11 [32,512,501]", timed(lambda: c1((x.clone(), lens)))))
        c2 = MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME).eval()
        x = torch.randn(N, 1, 80, 1001, device="cuda")
        l2 = torch.full((N,), 1001, dtype=torch.int64)
        rows.append(("MaskConv2d 1->32 41x11 s2 [32,1,80,1001]", timed(lambda: c2((x.clone(), l2)))))
        y = torch.randn(T, N, 29, device="cuda")
        rows.append(("CTCGreedyDecoder [501,32,29]", timed(lambda: CTCGreedyDecoder(28)(y, lens))))
    y = torch.randn(T, N, 29, device="cuda")
    tgt = torch.randint(0, 28, (N, 120), dtype=torch.int32)
    tl = torch.full((N,), 120, dtype=torch.int32)
    xl = torch.full((N,), T, dtype=torch.int32)
    for dim in (-1, 0, 1):
        loss = CTCLoss(blank=28, reduction="sum", dim=dim)
        rows.append((f"CTCLoss dim={dim} forward", timed(lambda: loss((y, xl), (tgt, tl)))))
        yg = y.clone().requires_grad_(True)

        def fb():
            yg.grad = None
            loss((yg, xl), (tgt, tl)).backward()
        rows.append((f"CTCLoss dim={dim} forward + backward", timed(fb)))
    big = torch.randn(T, N, 1000, device="cuda")
    tgt2 = torch.randint(0, 999, (N, 60), dtype=torch.int32)
    tl2 = torch.full((N,), 60, dtype=torch.int32)
    loss = CTCLoss(blank=999, reduction="sum")
    rows.append(("CTCLoss V=1000 forward", timed(lambda: loss((big, xl), (tgt2, tl2)))))
    bg = big.clone().requires_grad_(True)

    def fb2():
        bg.grad = None
        loss((bg, xl), (tgt2, tl2)).backward()
    rows.append(("CTCLoss V=1000 forward + backward", timed(fb2)))
    from myrtlespeech_amd.data.preprocess import MFCC, Standardize
    w = (torch.randn(32, 160000) * 0.1).cuda()
    wl = torch.full((32,), 160000)
    mf = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160})
    rows.append(("MFCC 32 x 10 s", timed(lambda: mf.batch(w, wl))))
    for name, ms in rows:
        print("%-52s %9.3f ms" % (name, ms), flush=True)


if __name__ == "__main__":
    main()
