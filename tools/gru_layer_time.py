"""One persistent-GRU layer call (projection + recurrence) at the shipped width and two others: ms per call (HIP events).
For tools/ab_lib.sh (AB_TAIL=3)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(0)
    T, N = 501, 32
    lens = torch.full((N,), T, dtype=torch.int64)
    with torch.no_grad():
        for H, bi in ((2560, False), (1280, True), (1024, True)):
            m = RNN(RNNType.GRU, H, H, num_layers=1, bidirectional=bi).eval()
            m.check_status = False
            x = torch.randn(T, N, H, device="cuda")
            ms = timed(lambda: m((x, lens)), warm=3, it=10)
            print(f"GRU H={H} bi={bi} [{T},{N},{H}]: {ms:7.3f} ms per layer call", flush=True)


if __name__ == "__main__":
    main()
