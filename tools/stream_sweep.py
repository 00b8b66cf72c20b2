"""A net for streaming shapes nobody benchmarks: ChunkedDeepSpeech2 on the config-2 network over stream counts x chunk lengths --
ms per chunk (HIP events, median of 8 clips of 12 chunks), the real-time factor, and whether the chunks replayed as HIP graphs."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

import bench
from tools.bench_configs import LAST_SAMPLES, ev_timed


def main():
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    m = bench.build_model()
    chunks = 12
    with torch.no_grad():
        for chunk in [int(v) for v in os.environ.get("SWEEP_CHUNK", "16,32,64").split(",")]:
            for N in [int(v) for v in os.environ.get("SWEEP_N", "1,8,32,64,96,128").split(",")]:
                g = torch.Generator().manual_seed(5)
                x = torch.randn(N, 1, 80, chunk * chunks, generator=g).cuda()
                lens = torch.full((N,), chunk * chunks, dtype=torch.int64)
                stream = ChunkedDeepSpeech2(m, chunk)
                ev_timed(lambda: stream(x, lens), 3, 8)
                ms = float(np.median(LAST_SAMPLES)) / chunks
                print(f"chunk {chunk:3d} frames ({chunk * 10} ms) x {N:4d} streams: {ms:7.3f} ms per chunk = {N * chunk * 0.01 / (ms * 1e-3):8.0f} x real time; "
                      f"graph replays {stream.graph_replays}, graph error {stream.graph_error}", flush=True)


if __name__ == "__main__":
    main()
