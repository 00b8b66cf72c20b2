#!/usr/bin/env python3
"""Timeline of one TwoBatchesInFlight call on the bench network: when the host returns, when each batch's transcripts are
enqueued (HIP events), the steady-state spacing of batch pairs -- and the hipMalloc stall of a first call that is longer than
its warm-up (why bench.py warms the pipeline with a call of the timed call's length)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from myrtlespeech_amd.pipeline import TwoBatchesInFlight
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
model = bench.build_model(); model.rnn.check_status = False
dec = CTCGreedyDecoder(28)
g = torch.Generator().manual_seed(1234)
x = torch.randn(32, 1, 80, 1001, generator=g).cuda()
lens = torch.full((32,), 1001, dtype=torch.int64)
starts = {}
def pre(k):
    starts[k] = torch.cuda.Event(enable_timing=True); starts[k].record()
def post(out):
    p = dec.launch(out[0][0], out[0][1]); e = torch.cuda.Event(enable_timing=True); e.record(); return p, e
pipe = TwoBatchesInFlight(model, post=post, pre=pre)
pipe([(x, lens)] * 4)
for K in (20, 20, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base = torch.cuda.Event(enable_timing=True); base.record()
    pend = pipe([(x, lens)] * K)
    t1 = time.perf_counter()
    for pd, _ in pend: pd.result()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ends = [base.elapsed_time(e) for _, e in pend]
    sts = [base.elapsed_time(starts[k]) for k in range(K)]
    print(f"K={K}: wall {1e3*(t2-t0):.1f} ms ({1e3*(t2-t0)/K:.3f}/batch); pipe() returned after {1e3*(t1-t0):.1f}; first start {sts[0]:.2f}, last end {ends[-1]:.1f}")
    print("   batch end times (ms):", " ".join(f"{e:.1f}" for e in ends[:8]), "...", " ".join(f"{e:.1f}" for e in ends[-4:]))
    print("   deltas:", " ".join(f"{b-a:.1f}" for a, b in zip(ends[:11], ends[1:12])))
