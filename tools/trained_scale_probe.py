#!/usr/bin/env python3
"""Choose the gains of the "trained-scale" fixtures (tests/golden/gen_golden.py::gen_*_trained_scale_summary).

Build container only (imports the REFERENCE from /root/reference/src, CPU torch).  The reference ships no checkpoint
(run/run.py:172-185 only saves), so a default-init network lives in the linear regime: |logit| ~ 0.02, gates never
saturate.  This probe multiplies the recurrent weights by g_r (weight_ih / weight_hh, separately if asked) and the
fully-connected weights by g_f and prints, for the REFERENCE model: mean / max |logit|, the share of LSTM gate
pre-activations with |.| > 4, the number of distinct greedy symbols -- and how well conditioned the network is at those
gains: the same model in float64 against itself in float32 (a fixture whose own float32 rounding moves the logits by more
than the 1e-3 gate pins nothing).

    PYTHONDONTWRITEBYTECODE=1 python tools/trained_scale_probe.py --gih 6 --ghh 6 --gf 12 --n 4 --t 401
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True
from myrtlespeech.model.cnn import MaskConv2d, PaddingMode  # noqa: E402
from myrtlespeech.model.deep_speech_2 import DeepSpeech2  # noqa: E402
from myrtlespeech.model.fully_connected import FullyConnected  # noqa: E402
from myrtlespeech.model.rnn import RNN, RNNType  # noqa: E402
from myrtlespeech.model.seq_len_wrapper import SeqLenWrapper  # noqa: E402
from myrtlespeech.post_process.ctc_greedy_decoder import CTCGreedyDecoder  # noqa: E402

torch.set_grad_enabled(False)


def act():
    return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())


def build(seed, hidden, layers):
    torch.manual_seed(seed)
    cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.LSTM, 640, hidden, num_layers=layers, bidirectional=True, forget_gate_bias=1.0)
    fc = FullyConnected(2 * hidden, 29, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
    return DeepSpeech2(cnn, rnn, None, fc).eval()


def apply_gains(m, gih, ghh, gf, gconv=1.0):
    for k, v in m.state_dict().items():
        if "weight_ih" in k:
            v.mul_(gih)
        elif "weight_hh" in k:
            v.mul_(ghh)
        elif k.startswith("fully_connected") and k.endswith("weight"):
            v.mul_(gf)
        elif k.startswith("cnn") and k.endswith("weight"):
            v.mul_(gconv)


def gate_stats(m, x_rnn, lens):
    """Share of gate pre-activations beyond |4|, layer by layer, from single-layer torch LSTMs carrying the model's weights
    (the stack's own per-layer h sequences are not exposed); equal lengths only."""
    lstm = m.rnn.rnn
    H = lstm.hidden_size
    inp = x_rnn
    shares = []
    for l in range(lstm.num_layers):
        one = torch.nn.LSTM(inp.shape[2], H, 1, bidirectional=True).to(inp.dtype)
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(one, f"{nm}_l0{sfx}").copy_(getattr(lstm, f"{nm}_l{l}{sfx}"))
        out, _ = one(inp)
        tot = big = 0
        for d, sfx in enumerate(("", "_reverse")):
            h = out[:, :, d * H:(d + 1) * H]
            hp = torch.zeros_like(h)
            if d == 0:
                hp[1:] = h[:-1]
            else:
                hp[:-1] = h[1:]
            g = inp @ getattr(one, f"weight_ih_l0{sfx}").T + hp @ getattr(one, f"weight_hh_l0{sfx}").T \
                + getattr(one, f"bias_ih_l0{sfx}") + getattr(one, f"bias_hh_l0{sfx}")
            tot += g.numel()
            big += int((g.abs() > 4).sum())
        shares.append(big / tot)
        inp = out
    return shares, inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gih", type=float, default=1.0)
    ap.add_argument("--ghh", type=float, default=1.0)
    ap.add_argument("--gf", type=float, default=1.0)
    ap.add_argument("--gconv", type=float, default=1.0)
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--t", type=int, default=401)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-f64", action="store_true")
    a = ap.parse_args()
    m = build(a.seed, a.hidden, a.layers)
    apply_gains(m, a.gih, a.ghh, a.gf, a.gconv)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(a.n, 1, 80, a.t, generator=g)
    lens = torch.full((a.n,), a.t, dtype=torch.int64)
    t0 = time.time()
    (y, ol), hid = m((x.clone(), lens))
    print(f"forward {time.time() - t0:.1f} s; logits mean |.| {float(y.abs().mean()):.3f} max {float(y.abs().max()):.3f}")
    dec = CTCGreedyDecoder(28)(y, ol)
    syms = sorted({s for u in dec for s in u})
    print(f"greedy: {len(syms)} distinct symbols, lengths {[len(u) for u in dec]}")
    top2 = torch.topk(y, 2, dim=2).values
    margin = (top2[..., 0] - top2[..., 1])
    print(f"top1-top2 margin: median {float(margin.median()):.3e} min {float(margin.min()):.3e}; "
          f"share below 1e-3: {float((margin < 1e-3).float().mean()):.4f}")
    # the RNN's input, as DeepSpeech2.forward forms it (deep_speech_2.py:123-172)
    h, l2 = m.cnn((x.clone(), lens))
    N, C, F, T = h.shape
    x_rnn = h.view(N, C * F, T).permute(2, 0, 1).contiguous()
    shares, top = gate_stats(m, x_rnn, lens)
    print("share of gate pre-activations with |.| > 4 per layer:", [f"{s:.3f}" for s in shares],
          f"overall {sum(shares) / len(shares):.3f}")
    if not a.no_f64:
        m64 = build(a.seed, a.hidden, a.layers)
        apply_gains(m64, a.gih, a.ghh, a.gf, a.gconv)
        m64 = m64.double()
        (y64, _), hid64 = m64((x.double(), lens))
        d = (y64 - y.double()).abs()
        print(f"float64 twin vs float32 reference: max |dlogit| {float(d.max()):.3e} mean {float(d.mean()):.3e}; "
              f"h_n {float((hid64[0] - hid[0].double()).abs().max()):.3e} c_n {float((hid64[1] - hid[1].double()).abs().max()):.3e}")
        dec64 = CTCGreedyDecoder(28)(y64.float(), ol)
        print("greedy transcripts equal in both:", dec64 == dec)


if __name__ == "__main__":
    main()
