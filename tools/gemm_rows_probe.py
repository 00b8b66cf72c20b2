"""The split GEMM over few rows (a 32-stream chunk: 512 rows; one utterance: 501) at the projections' widths: us per call incl. the
operand split.  256 x 128 tiles give such outputs 120 .. 128 workgroups -- half of the chip: 512 rows cost 100 us where 1 024 cost 126
(a 128-row tile form of the LDS-DMA kernel is the open item)."""
import sys, torch
sys.path.insert(0, '.')
from tools.op_audit import timed
from myrtlespeech_amd.model.fully_connected import run_linear_stack
torch.manual_seed(0)
with torch.no_grad():
    for K, N in ((2560, 7680), (2048, 8192), (640, 8192)):
        lin = torch.nn.Linear(K, N).cuda()
        for M in (256, 501, 512, 768, 1002, 1024, 2048):
            x = torch.randn(M, K, device="cuda")
            ms = timed(lambda: run_linear_stack(x, [(lin, None)]), warm=3, it=10)
            print(f"K={K} N={N} M={M:5d}: {ms*1e3:7.1f} us  {2.0*M*K*N/(ms*1e-3)/1e12:6.1f} TF useful", flush=True)
