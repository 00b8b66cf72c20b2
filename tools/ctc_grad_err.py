import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from oracle import ds_oracle as O
from myrtlespeech_amd.loss.ctc_loss import CTCLoss
for (Tn, N, V, L, scale) in [(501, 8, 29, 120, 1.0), (501, 8, 29, 120, 3.0), (300, 5, 29, 128, 2.0), (64, 9, 29, 20, 2.0)]:
    rng = np.random.default_rng(Tn * 11 + L)
    x = (rng.normal(size=(Tn, N, V)) * scale).astype(np.float32)
    xl = np.full(N, Tn, np.int32); yl = np.full(N, L, np.int32)
    y = rng.integers(0, V - 1, size=(N, L)).astype(np.int32)
    want = O.ctc_grad(x, xl, y, yl, np.ones(N, np.float32), V - 1, False)
    loss = CTCLoss(blank=V - 1, reduction="sum")
    for flag in ("0", "1"):
        os.environ["MS_CTC_WAVE"] = flag
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        loss((xt, torch.from_numpy(xl)), (torch.from_numpy(y), torch.from_numpy(yl))).backward()
        g = xt.grad.cpu().numpy()
        e = np.abs(g - want)
        print(Tn, N, V, L, scale, "pipeline" if flag == "1" else "lds-row ", "max abs err %.3e  mean %.3e  99.9pct %.3e" % (e.max(), e.mean(), np.quantile(e, 0.999)), flush=True)
