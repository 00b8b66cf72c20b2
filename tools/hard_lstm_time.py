import sys, torch
sys.path.insert(0, '.')
from tools.op_audit import timed
from myrtlespeech_amd.model.hard_lstm import HardLSTM
torch.manual_seed(0)
with torch.no_grad():
    for H, bi in ((1024, True), (1280, True), (2048, False), (2048, True)):
        m = HardLSTM(H, H, num_layers=1, bidirectional=bi, batch_first=False).cuda().eval()
        x = torch.randn(501, 32, H, device="cuda")
        lens = torch.full((32,), 501, dtype=torch.int64)
        print(f"HardLSTM H={H} bi={bi} [501,32,{H}]: {timed(lambda: m((x, lens))):7.3f} ms", flush=True)
