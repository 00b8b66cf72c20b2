"""Which (cell, width, directions) still take a launch per step?  One layer at [200, 32, H] for every cell kind the reference's RNN /
HardLSTM wrappers build, widths 64 .. 3 000: us per step; anything far above ~10 us is on the per-step kernels."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(0)
    T, N = 200, 32
    lens = torch.full((N,), T, dtype=torch.int64)
    with torch.no_grad():
        for name in ("LSTM", "HardLSTM", "GRU", "BASIC_RNN"):
            for H in (64, 200, 320, 512, 800, 1024, 1280, 1536, 2048, 2560, 3000):
                row = []
                for bi in (False, True):
                    if name == "HardLSTM":
                        m = HardLSTM(H, H, num_layers=1, bidirectional=bi, batch_first=False).cuda().eval()
                    else:
                        m = RNN(getattr(RNNType, name), H, H, num_layers=1, bidirectional=bi,
                                forget_gate_bias=1.0 if name == "LSTM" else None).eval()
                    x = torch.randn(T, N, H, device="cuda")
                    ms = timed(lambda: m((x, lens)), warm=2, it=4)
                    row.append(ms * 1e3 / T)
                flag = "  <-- a launch per step?" if max(row) > 14.0 else ""
                print(f"{name:9s} H={H:5d}: {row[0]:6.1f} us per step unidirectional, {row[1]:6.1f} bidirectional{flag}", flush=True)


if __name__ == "__main__":
    main()
