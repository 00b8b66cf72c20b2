"""A net for convolution shapes nobody benchmarks: MaskConv2d / MaskConv1d layers of DS2-family front-ends at batch 32 x 10 s --
ms per call (HIP events, input clone included) and the executed TFLOP/s (2 x taps x outputs), so that a shape that falls off
the MFMA paths shows."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed


def main():
    from myrtlespeech_amd.model.cnn import MaskConv1d, MaskConv2d, PaddingMode
    torch.manual_seed(0)
    N, T = 32, 1001
    l = torch.full((N,), T, dtype=torch.int64)
    cases2d = [  # (cin, cout, kernel [f, t], stride [f, t], input features)
        (1, 32, [41, 11], [2, 2], 80), (1, 32, [41, 11], [2, 2], 161), (32, 32, [21, 11], [2, 1], 40), (32, 32, [21, 11], [2, 1], 81),
        (32, 96, [21, 11], [2, 1], 41), (32, 64, [21, 11], [2, 1], 40), (64, 64, [21, 11], [2, 1], 20), (1, 64, [41, 11], [2, 2], 80),
        (32, 32, [5, 5], [1, 1], 40), (32, 32, [3, 3], [1, 1], 40), (16, 32, [21, 11], [2, 1], 40), (8, 8, [5, 5], [2, 1], 40)]
    with torch.no_grad():
        for cin, cout, k, s, F in cases2d:
            t_in = T if cin == 1 else 501
            m = MaskConv2d(cin, cout, k, s, PaddingMode.SAME).eval()
            x = torch.randn(N, cin, F, t_in, device="cuda")
            ll = torch.full((N,), t_in, dtype=torch.int64)
            y, _ = m((x.clone(), ll))
            ms = timed(lambda: m((x.clone(), ll)))
            cp = timed(lambda: x.clone())
            fl = 2.0 * y.numel() * cin * k[0] * k[1]
            print(f"MaskConv2d {cin:3d} -> {cout:3d}  k {k}  s {s}  in [{N},{cin},{F},{t_in}] -> {list(y.shape)}: {ms - cp:7.3f} ms = {fl / ((ms - cp) * 1e-3) / 1e12:6.1f} TFLOP/s executed", flush=True)
        for cin, cout, k, s in [(161, 1280, 11, 2), (80, 512, 11, 2), (512, 512, 11, 1), (1280, 1280, 5, 1), (40, 256, 5, 1)]:
            m = MaskConv1d(cin, cout, k, s, PaddingMode.SAME).eval()
            x = torch.randn(N, cin, T, device="cuda")
            y, _ = m((x.clone(), l))
            ms = timed(lambda: m((x.clone(), l)))
            cp = timed(lambda: x.clone())
            fl = 2.0 * y.numel() * cin * k
            print(f"MaskConv1d {cin:4d} -> {cout:4d}  k {k}  s {s}  in [{N},{cin},{T}] -> {list(y.shape)}: {ms - cp:7.3f} ms = {fl / ((ms - cp) * 1e-3) / 1e12:6.1f} TFLOP/s executed", flush=True)


if __name__ == "__main__":
    main()
