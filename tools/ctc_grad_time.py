"""CTC loss forward + backward through the autograd wrapper, wall time per call (run under tools/rocprof_script.sh for kernels)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from myrtlespeech_amd.loss.ctc_loss import CTCLoss

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(501, 32, 29, 120), (1001, 32, 29, 250)]
rng = np.random.default_rng(0)
for (Tn, N, V, S) in shapes:
    x = torch.from_numpy(rng.normal(size=(Tn, N, V)).astype(np.float32)).cuda().requires_grad_(True)
    xl = torch.full((N,), Tn, dtype=torch.int32)
    yl = torch.full((N,), S, dtype=torch.int32)
    y = torch.from_numpy(rng.integers(0, V - 1, size=(N, S)).astype(np.int32))
    loss = CTCLoss(blank=V - 1, reduction="sum")

    def step():
        x.grad = None
        v = loss((x, xl), (y, yl))
        v.backward()
        return v
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        v = step()
    b.record()
    torch.cuda.synchronize()
    print(Tn, N, V, S, "forward + backward %.4f ms" % (a.elapsed_time(b) / 10), float(v), float(x.grad.abs().sum()), flush=True)
