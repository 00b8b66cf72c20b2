"""A net for recurrent widths x batch sizes: one bidirectional LSTM layer call at H in {256 .. 2048}, N in {16, 32, 64}: ms per
call and us per utterance, so that a (width, batch) pair that leaves CUs idle shows."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(0)
    T = 501
    kinds = {"LSTM": RNNType.LSTM, "GRU": RNNType.GRU}
    with torch.no_grad():
        for name in os.environ.get("SWEEP_KINDS", "LSTM,GRU").split(","):
            for H in [int(v) for v in os.environ.get("SWEEP_H", "256,512,768,1024,1280,1536,2048").split(",")]:
                for bi in (True, False):
                    m = RNN(kinds[name], H, H, num_layers=1, bidirectional=bi, forget_gate_bias=1.0 if name == "LSTM" else None).eval()
                    m.check_status = False
                    row = []
                    for N in (16, 32, 64):
                        lens = torch.full((N,), T, dtype=torch.int64)
                        x = torch.randn(T, N, H, device="cuda")
                        row.append(timed(lambda: m((x, lens)), warm=2, it=5))
                    print(f"{name} H={H:5d} bi={bi!s:5s}: N=16 {row[0]:7.3f}  N=32 {row[1]:7.3f}  N=64 {row[2]:7.3f} ms   (N=64 / N=32: {row[2] / row[1]:.2f})", flush=True)


if __name__ == "__main__":
    main()
