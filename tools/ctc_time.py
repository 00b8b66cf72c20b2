"""CTC loss forward: the LDS-row kernel (MS_CTC_WAVE=0) against the four-wave pipeline, wall time per call through the Python
wrapper (run under tools/rocprof_script.sh for the kernels alone).  Arguments: T,N,V,L quadruples (default: the bench shape
and the number of active waves of the pipeline at it)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from myrtlespeech_amd.loss.ctc_loss import CTCLoss

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (501, 32, 29, 31), (501, 32, 29, 63), (501, 32, 29, 95), (501, 32, 29, 120), (1001, 32, 29, 250), (1001, 32, 29, 400)]
rng = np.random.default_rng(0)
for (Tn, N, V, S) in shapes:
    x = torch.from_numpy(rng.normal(size=(Tn, N, V)).astype(np.float32)).cuda()
    xl = torch.full((N,), Tn, dtype=torch.int32)
    yl = torch.full((N,), S, dtype=torch.int32)
    y = torch.from_numpy(rng.integers(0, V - 1, size=(N, S)).astype(np.int32))
    loss = CTCLoss(blank=V - 1, reduction="sum")
    for mode in ("0", "1"):
        os.environ["MS_CTC_WAVE"] = mode
        for _ in range(3):
            v = loss((x, xl), (y, yl))
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            v = loss((x, xl), (y, yl))
        b.record()
        torch.cuda.synchronize()
        print(Tn, N, V, S, "pipeline" if mode == "1" else "lds-row ", "%.4f ms" % (a.elapsed_time(b) / 20), float(v), flush=True)
