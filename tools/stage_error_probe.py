#!/usr/bin/env python3
"""Where does the split arithmetic lose accuracy on the trained-scale config-2 network?  (GPU box; no reference needed.)

Ground truth: stock torch CPU operators in FLOAT64 on the fixture's weights and input (seed-0 weights x TRAINED gains, the
config-2 batch): conv stack, every BiLSTM layer (single-layer torch LSTMs on packed sequences), the two linear layers.
Then THIS library, stage by stage, each stage fed the float64 truth of the stage below it (rounded to float32), in the
process's MS_PRECISION mode: a stage's OWN error, not the accumulated one -- and the accumulated one of the whole model
beside it.  Errors are max |.| over the frames that exist.

    MS_PRECISION=f16x3 python tools/stage_error_probe.py [--n 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

torch.set_grad_enabled(False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8, help="utterances (the first n of the fixture's batch of 32)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    from util import Golden, apply_trained_gains
    from oracle import ds_oracle as O
    from myrtlespeech_amd.model.rnn import RNN, RNNType

    g = Golden("ds2_cfg2_trained_summary")
    model = bench.build_model()
    apply_trained_gains(model, g.cfg["gains"])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    x = torch.randn(g.cfg["N"], 1, 80, g.cfg["T"], generator=gen)
    lens = torch.sort(torch.randint(501, 1002, (g.cfg["N"],), generator=gen), descending=True).values
    lens[0] = g.cfg["T"]
    x, lens = x[:a.n].contiguous(), lens[:a.n].contiguous()

    # ---- float64 truth, stage by stage
    t0 = time.time()
    h = x.double().clone()
    l = lens.clone()
    for idx, (sf, st) in ((0, (2, 2)), (2, (2, 1))):
        w, b = sd[f"cnn.{idx}.weight"].double(), sd[f"cnn.{idx}.bias"].double()
        T = h.shape[-1]
        h.masked_fill_((torch.arange(T)[None, :] >= l[:, None])[:, None, None, :], 0.0)
        pf, pt = O.pad_same(h.shape[2], w.shape[2], sf), O.pad_same(T, w.shape[3], st)
        h = F.hardtanh(F.conv2d(F.pad(h, (pt[0], pt[1], pf[0], pf[1])), w, b, stride=(sf, st)), 0.0, 20.0)
        l = (((l.float() + pt[0] + pt[1] - (w.shape[3] - 1) - 1) / st) + 1).floor().to(l.dtype)
    n, c, f, t = h.shape
    conv64 = h
    seq = h.view(n, c * f, t).permute(2, 0, 1).contiguous()
    layers64 = []
    inp = seq
    for k in range(5):
        one = torch.nn.LSTM(inp.shape[2], 1024, 1, bidirectional=True).double()
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(one, f"{nm}_l0{sfx}").copy_(sd[f"rnn.rnn.{nm}_l{k}{sfx}"].double())
        out, _ = one(torch.nn.utils.rnn.pack_padded_sequence(inp, l))
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=inp.shape[0])
        layers64.append(out)
        inp = out
    w1, b1 = sd["fully_connected.fully_connected.0.weight"].double(), sd["fully_connected.fully_connected.0.bias"].double()
    w2, b2 = sd["fully_connected.fully_connected.2.weight"].double(), sd["fully_connected.fully_connected.2.bias"].double()
    fc1_64 = F.hardtanh(F.linear(inp, w1, b1), 0.0, 20.0)
    y64 = F.linear(fc1_64, w2, b2)
    print(f"float64 truth of {a.n} utterances: {time.time() - t0:.1f} s", flush=True)
    valid = (torch.arange(t)[:, None] < l[None, :])

    def err(got, want):
        d = (got.double().cpu() - want).abs()
        return float(d[valid].max()), float(d[valid].mean())

    rec = {"mode": os.environ.get("MS_PRECISION", "f16x3"), "n": a.n}
    # ---- whole model (accumulated error)
    (y, ol), _ = model((x.clone(), lens))
    rec["whole_model_logits"] = err(y, y64)
    # ---- conv stack alone
    hc, lc = model.cnn((x.clone().cuda(), lens))
    d = (hc.double().cpu() - conv64).abs()
    rec["conv_stack"] = (float(d.max()), float(d.mean()))
    rec["conv_out_abs_max"] = float(conv64.abs().max())
    # ---- every LSTM layer alone, fed the truth of the layer below
    inp64 = seq
    for k in range(5):
        r = RNN(RNNType.LSTM, inp64.shape[2], 1024, num_layers=1, bidirectional=True)
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(r.rnn, f"{nm}_l0{sfx}").copy_(sd[f"rnn.rnn.{nm}_l{k}{sfx}"])
        (out, _), _ = r((inp64.float().cuda(), l))
        rec[f"lstm_layer_{k}_alone"] = err(out, layers64[k])
        inp64 = layers64[k]
    # ---- the stack from the true conv output
    (out, _), _ = model.rnn((seq.float().cuda(), l))
    rec["lstm_stack_from_true_conv"] = err(out, layers64[-1])
    # ---- FC from the true top layer
    (yf, _) = model.fully_connected((layers64[-1].float().transpose(0, 1).contiguous().cuda(), l))
    rec["fc_from_true_top"] = err(yf.transpose(0, 1), y64)
    # float32 torch CPU on the same stages, for scale: FC only (cheap)
    y32 = F.linear(F.hardtanh(F.linear(layers64[-1].float(), w1.float(), b1.float()), 0.0, 20.0), w2.float(), b2.float())
    rec["fc_torch_f32_from_true_top"] = err(y32, y64)
    print(json.dumps(rec, indent=1))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
