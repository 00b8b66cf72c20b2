"""A net for fully connected shapes nobody benchmarks: run_linear_stack on one Linear (+ clamp) over row counts x (K, N) --
ms per call and useful TFLOP/s (2 M K N), so that a shape that falls between the tile forms shows."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.fully_connected import run_linear_stack
    torch.manual_seed(0)
    with torch.no_grad():
        for K, N in ((1024, 1024), (2048, 1024), (2560, 1024), (2048, 2048), (640, 8192), (2048, 8192), (1024, 29), (2048, 29), (1024, 5000)):
            lin = torch.nn.Linear(K, N).cuda()
            row = []
            for M in (201, 512, 1024, 2048, 4096, 16032, 32064, 64128):
                x = torch.randn(M, K, device="cuda")
                ms = timed(lambda: run_linear_stack(x, [(lin, (0.0, 20.0))]), warm=2, it=5)
                row.append(f"M={M:6d} {ms:7.3f} ms {2.0 * M * K * N / (ms * 1e-3) / 1e12:6.1f} TF")
            print(f"Linear K={K:5d} N={N:5d}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
