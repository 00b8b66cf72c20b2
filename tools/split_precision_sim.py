#!/usr/bin/env python3
"""CPU simulation of the split-operand arithmetics on the trained-scale config-2 network (tools/trained_scale_probe.py for
the regime): which operand split keeps the logits within 1e-3 of the reference when the gates saturate?

Every large contraction (LSTM projections and recurrence, hidden FC layer) is run with both operands replaced by a
two-plane split  x ~ hi + lo  and the product  hi*hi + lo*hi + hi*lo  (what three MFMA passes compute), accumulated in
float64 here so that ONLY the split's error shows; state and layer outputs are rounded to float32 between operators as in
the kernels.  Planes: bf16 (the round 1-5 default, "bf16x3"), fp16 ("f16x3": 11 + 11 mantissa bits, gradual underflow as
the hardware would have to honour it, or flushed below 2^-14), against the unsplit float64 product.

Needs no reference import: plain torch CPU, seeded like bench.build_model().

    python tools/split_precision_sim.py --n 4 --t 301
"""
import argparse
import math

import torch

torch.set_grad_enabled(False)
F64 = torch.float64


def split(x, dt, flush=False, tag=False):
    hi = x.to(dt).to(torch.float32)
    lo = (x - hi).to(dt).to(torch.float32)
    if flush and dt == torch.float16:
        lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
        hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
    return hi.to(F64), lo.to(F64)


def make_mm(mode):
    if mode == "exact":
        return lambda a, w: (a.to(F64) @ w.to(F64).T)
    dt = torch.bfloat16 if mode.startswith("bf16") else torch.float16
    flush = mode.endswith("flush")

    def mm(a, w):
        ah, al = split(a, dt, flush)
        wh, wl = split(w, dt, flush)
        return ah @ wh.T + al @ wh.T + ah @ wl.T
    return mm


def lstm_stack(x, params, mm, H):
    """x [T, N, In] f32; params[layer][dir] = (w_ih, w_hh, b_ih, b_hh); returns [T, N, 2H] f32."""
    T, N, _ = x.shape
    inp = x
    for layer in params:
        outs = []
        for d, (w_ih, w_hh, b_ih, b_hh) in enumerate(layer):
            xp = (mm(inp.reshape(T * N, -1), w_ih) + (b_ih + b_hh).to(F64)).to(torch.float32).view(T, N, 4 * H)
            h = torch.zeros(N, H)
            c = torch.zeros(N, H)
            out = torch.zeros(T, N, H)
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in order:
                g = (xp[t].to(F64) + mm(h, w_hh)).to(torch.float32)
                i, f, gg, o = g.chunk(4, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                out[t] = h
            outs.append(out)
        inp = torch.cat(outs, dim=2)
    return inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--t", type=int, default=301)
    ap.add_argument("--gih", type=float, default=16.0)
    ap.add_argument("--ghh", type=float, default=2.0)
    ap.add_argument("--gf", type=float, default=6.0)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=5)
    a = ap.parse_args()
    H = a.hidden
    torch.manual_seed(0)
    conv = torch.nn.Sequential(torch.nn.Conv2d(1, 32, (41, 11), (2, 2), padding=(20, 5)), torch.nn.Hardtanh(0, 20),
                               torch.nn.Conv2d(32, 32, (21, 11), (2, 1), padding=(10, 5)), torch.nn.Hardtanh(0, 20))
    lstm = torch.nn.LSTM(640, H, a.layers, bidirectional=True)
    fc1 = torch.nn.Linear(2 * H, 1024)
    fc2 = torch.nn.Linear(1024, 29)
    params = []
    for l in range(a.layers):
        layer = []
        for sfx in ("", "_reverse"):
            layer.append((getattr(lstm, f"weight_ih_l{l}{sfx}") * a.gih, getattr(lstm, f"weight_hh_l{l}{sfx}") * a.ghh,
                          getattr(lstm, f"bias_ih_l{l}{sfx}"), getattr(lstm, f"bias_hh_l{l}{sfx}")))
        params.append(layer)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(a.n, 1, 80, a.t, generator=g)
    hc = conv(x)
    n_, c_, f_, t_ = hc.shape
    x_rnn = hc.view(n_, c_ * f_, t_).permute(2, 0, 1).contiguous()
    res = {}
    for mode in ("exact", "bf16x3", "f16x3", "f16x3_flush"):
        mm = make_mm(mode)
        top = lstm_stack(x_rnn, params, mm, H)
        T, N, _ = top.shape
        h1 = torch.clamp((mm(top.reshape(T * N, -1), fc1.weight * a.gf) + fc1.bias.to(F64)).to(torch.float32), 0, 20)
        y = (h1.to(F64) @ (fc2.weight * a.gf).to(F64).T + fc2.bias.to(F64)).to(torch.float32)
        res[mode] = (top, y)
        if mode == "exact":
            print(f"logits mean |.| {float(y.abs().mean()):.3f} max {float(y.abs().max()):.3f}")
        else:
            dy = (y - res["exact"][1]).abs()
            dt_ = (top - res["exact"][0]).abs()
            flips = int((y.argmax(-1) != res["exact"][1].argmax(-1)).sum())
            print(f"{mode:12s} max |dlogit| {float(dy.max()):.3e} mean {float(dy.mean()):.3e}; top-layer h max err "
                  f"{float(dt_.max()):.3e}; arg-max flips {flips} of {T * N}")


if __name__ == "__main__":
    main()
