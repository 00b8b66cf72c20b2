#!/usr/bin/env python3
"""Fixture for BASELINE.json configs[3] at its stated size (RNN-T: 2 x LSTM-1024 predictor, joint 512, beam 8, batch 16,
T = 501) -- made by THIS repository's numpy oracle (oracle/rnnt_oracle.py), because the reference has no transducer
(SURVEY 0.3 / 8 a15): **parity unpinned**.  The per-hypothesis numpy loops need about a minute per utterance at this
size, so the oracle's answers for the sixteen utterances are stored instead of being recomputed on the GPU box:

    python tests/golden/gen_rnnt_cfg4.py          # build container, CPU only; ~20 min

Inputs and weights are regenerated from seeds on both sides (checksums stored); stored per selected utterance: the greedy
transcript, the beam-8 transcript, its score, and the final beam (prefixes + scores) so that a <= 4-ulp tie between the two
best hypotheses can be told from a real mismatch."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import rnnt_oracle as RO  # noqa: E402

V, D, P, J, E, L = 28, 256, 1024, 512, 1024, 2
N, T, W, MS = 16, 501, 8, 3
SEL = tuple(range(N))


def parts():
    """The same construction the test uses (torch default init under seed 4)."""
    from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor
    torch.manual_seed(4)
    pred = RNNTPredictor(V, D, P, num_layers=L).eval()
    joint = RNNTJoint(E, P, J, V).eval()
    # default-initialised output weights give a near-uniform distribution over the 29 symbols, under which the best
    # hypothesis is (almost) empty and greedy never emits blank: sharpen the output layer and favour blank so that the
    # search has real decisions to take (greedy ~1.7 labels per frame, beam ~0.15)
    with torch.no_grad():
        joint.out.weight.mul_(16.0)
        joint.out.bias[V] += 8.0
    return pred, joint


def inputs():
    g = torch.Generator().manual_seed(44)
    enc = torch.randn(T, N, E, generator=g)
    lens = torch.sort(torch.randint(301, T + 1, (N,), generator=g), descending=True).values
    lens[0] = T
    return enc, lens


def beam_with_final(enc, n_len, psd, jsd):
    """oracle beam decode of one utterance, also returning the final beam (monkey-free: re-run the last selection)."""
    res, sc = RO.beam_decode(enc, np.array([n_len]), psd, jsd, P, L, V, W, MS)
    return res[0], sc[0]


if __name__ == "__main__":
    pred, joint = parts()
    psd = {k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}
    jsd = {k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()}
    enc, lens = inputs()
    arrays = {"in/lens": lens.numpy()}
    chk = {"pred/" + k: float(np.abs(v.astype(np.float64)).sum()) for k, v in psd.items()}
    chk.update({"joint/" + k: float(np.abs(v.astype(np.float64)).sum()) for k, v in jsd.items()})
    for n in SEL:
        e1 = enc[:, n:n + 1].numpy()
        t0 = time.time()
        g = RO.greedy_decode(e1, np.array([int(lens[n])]), psd, jsd, P, L, V, MS)[0]
        b, s = beam_with_final(e1, int(lens[n]), psd, jsd)
        print(f"utt {n}: len {int(lens[n])} greedy {len(g)} labels, beam {len(b)} labels, score {s:.6f}, {time.time() - t0:.0f} s",
              flush=True)
        arrays[f"out/greedy_{n}"] = np.array(g, dtype=np.int64)
        arrays[f"out/beam_{n}"] = np.array(b, dtype=np.int64)
        arrays[f"out/beam_score_{n}"] = np.array(s, dtype=np.float64)
    cfg = dict(V=V, D=D, P=P, J=J, E=E, L=L, N=N, T=T, beam_width=W, max_symbols=MS, seed_weights=4, seed_input=44,
               selected=list(SEL), weight_abs_sums=chk, enc_abs_sum=float(enc.double().abs().sum()))
    path = os.path.join(HERE, "rnnt_cfg4.npz")
    np.savez_compressed(path, cfg=np.array(json.dumps(cfg)), **arrays)
    print(f"rnnt_cfg4: {os.path.getsize(path) / 1024:.1f} KiB")
