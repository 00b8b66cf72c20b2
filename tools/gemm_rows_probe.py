"""The split GEMM over few rows (a 32-stream chunk: 512 rows; one utterance: 501) at the projections' widths: us per call incl. the
operand split, with 256 x 128 tiles (MS_GEMM_QUARTER_TILE=0: such outputs get 120 .. 128 workgroups, half of the chip) and with the
128 x 128 four-wave form (round 6), alternating in one process; the two results must be the same bits."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from tools.op_audit import timed
from myrtlespeech_amd.model.fully_connected import run_linear_stack

torch.manual_seed(0)
with torch.no_grad():
    for K, N in ((2560, 7680), (2048, 8192), (640, 8192), (2048, 1024)):
        lin = torch.nn.Linear(K, N).cuda()
        for M in (130, 256, 501, 512, 640, 768, 1002, 1024):
            x = torch.randn(M, K, device="cuda")
            res, ys = {}, {}
            for rep in range(2):
                for q in ("0", "1"):
                    os.environ["MS_GEMM_QUARTER_TILE"] = q
                    ys[q] = run_linear_stack(x, [(lin, (0.0, 20.0))])
                    res.setdefault(q, []).append(timed(lambda: run_linear_stack(x, [(lin, (0.0, 20.0))]), warm=2, it=10))
            a, b = min(res["0"]) * 1e3, min(res["1"]) * 1e3
            print(f"K={K} N={N} M={M:5d}: 256 x 128 tiles {a:7.1f} us | with the 128 x 128 form {b:7.1f} us | same bits: {torch.equal(ys['0'], ys['1'])}", flush=True)
os.environ.pop("MS_GEMM_QUARTER_TILE", None)
