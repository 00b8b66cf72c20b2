"""A net for batch sizes nobody benchmarks: one bidirectional LSTM-1024 / GRU-2560 layer and the config-2 network at batch sizes
other than 32 -- ms per call (HIP events) and ms per utterance, so that a size that falls off a fast path shows."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

import bench
from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(0)
    T = 501
    sizes = [int(v) for v in os.environ.get("SWEEP_N", "1,2,8,16,24,32,33,48,64,65,96,128,256").split(",")]
    with torch.no_grad():
        for kind, H, bi, inp in [(RNNType.LSTM, 1024, True, 2048), (RNNType.GRU, 2560, False, 2560)]:
            m = RNN(kind, inp, H, num_layers=1, bidirectional=bi, forget_gate_bias=1.0 if kind == RNNType.LSTM else None).eval()
            for N in sizes:
                lens = torch.full((N,), T, dtype=torch.int64)
                x = torch.randn(T, N, inp, device="cuda")
                ms = timed(lambda: m((x, lens)))
                print(f"RNN {kind.name} H={H} bi={bi} [{T},{N},{inp}]: {ms:8.3f} ms  {ms / N * 1e3:8.1f} us per utterance", flush=True)
        model = bench.build_model()
        for ragged in (False, True):
            for N in sizes:
                x = torch.randn(N, 1, 80, 1001, device="cuda")
                lens = torch.full((N,), 1001, dtype=torch.int64)
                if ragged:      # 5 .. 10 s, sorted, the longest a full clip
                    lens = torch.sort(torch.randint(501, 1002, (N,), generator=torch.Generator().manual_seed(N)), descending=True).values
                    lens[0] = 1001
                ms = timed(lambda: model((x, lens)))
                secs = float(lens.sum()) / 100.0
                print(f"config-2 network, batch {N}{' ragged' if ragged else ''}: {ms:8.3f} ms  {ms / N:8.3f} ms per utterance = {secs / (ms * 1e-3):9.0f} audio-s/s", flush=True)


if __name__ == "__main__":
    main()
