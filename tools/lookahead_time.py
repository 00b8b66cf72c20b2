import torch, time, sys
sys.path.insert(0, '.')
from myrtlespeech_amd.model.lookahead import Lookahead
torch.manual_seed(0)
for (T, N, F, ctx) in ((501, 32, 2560, 80), (501, 32, 2048, 20), (95, 32, 2560, 80)):
    m = Lookahead(F, ctx).cuda()
    x = torch.randn(N, F, T, device='cuda')
    # the DS2 path hands [T, N, F] storage viewed as [N, F, T]
    xs = torch.randn(T, N, F, device='cuda').permute(1, 2, 0)
    # the DS2 path: [T, N, F] storage in, [N, T, F] out (what the fully connected stack reads): coalesced both ways
    from myrtlespeech_amd.model.lookahead import lookahead_apply
    tnf = torch.randn(T, N, F, device='cuda')
    for _ in range(3): lookahead_apply(tnf, m.weight, (F, 1, N * F), N, F, T, out_layout="ntf")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): lookahead_apply(tnf, m.weight, (F, 1, N * F), N, F, T, out_layout="ntf")
    torch.cuda.synchronize()
    print(f"lookahead T={T} N={N} F={F} ctx={ctx} tnf->ntf: {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
    for name, inp in (("tcontig", x), ("strided", xs)):
        lens = torch.full((N,), T, dtype=torch.int64)
        for _ in range(3): y = m((inp, lens))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): y = m((inp, lens))
        torch.cuda.synchronize()
        print(f"lookahead T={T} N={N} F={F} ctx={ctx} {name}: {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
