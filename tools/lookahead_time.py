import torch, time, sys
sys.path.insert(0, '.')
from myrtlespeech_amd.model.lookahead import Lookahead
torch.manual_seed(0)
for (T, N, F, ctx) in ((501, 32, 2560, 80), (501, 32, 2048, 20), (95, 32, 2560, 80)):
    m = Lookahead(F, ctx).cuda()
    x = torch.randn(N, F, T, device='cuda')
    # the DS2 path hands [T, N, F] storage viewed as [N, F, T]
    xs = torch.randn(T, N, F, device='cuda').permute(1, 2, 0)
    for name, inp in (("tcontig", x), ("strided", xs)):
        lens = torch.full((N,), T, dtype=torch.int64)
        for _ in range(3): y = m((inp, lens))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): y = m((inp, lens))
        torch.cuda.synchronize()
        print(f"lookahead T={T} N={N} F={F} ctx={ctx} {name}: {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
