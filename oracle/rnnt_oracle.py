"""numpy oracle of this repository's OWN RNN-T specification (TEST INFRASTRUCTURE ONLY).

**Parity unpinned**: the reference snapshot has no transducer (no file,
no proto field), so there are no reference fixtures; this restates the specification of
``myrtlespeech_amd/model/rnnt.py`` / ``post_process/rnnt_decoder.py`` (Graves 2012
transducer: LSTM prediction network, additive tanh joint, greedy and time-synchronous beam
decoding) independently, per utterance, with plain loops.
"""
from typing import Dict, List, Tuple

import numpy as np

from oracle import ds_oracle as O

F32 = np.float32


def predictor_step(sd: Dict[str, np.ndarray], label: int, state, hidden: int, layers: int):
    """One prediction-network step for a single hypothesis: embedding -> L-layer LSTM."""
    x = sd["embedding.weight"][label][None, None, :].astype(F32)  # [1,1,D]
    rp = {k[len("rnn.rnn."):]: v for k, v in sd.items() if k.startswith("rnn.rnn.")}
    out, (hn, cn) = O.rnn_forward(O.LSTM, x, np.array([1]), rp, hidden, layers, False, state)
    return out[0, 0], (hn, cn)


def joint_logprobs(sd: Dict[str, np.ndarray], enc_vec: np.ndarray, pred_vec: np.ndarray) -> np.ndarray:
    e = O.linear(enc_vec[None], sd["enc_proj.weight"], sd["enc_proj.bias"])[0]
    p = O.linear(pred_vec[None], sd["pred_proj.weight"])[0]
    z = np.tanh(e + p).astype(F32)
    return O.log_softmax(O.linear(z[None], sd["out.weight"], sd["out.bias"])[0])


def _zero_state(layers, hidden):
    return (np.zeros((layers, 1, hidden), F32), np.zeros((layers, 1, hidden), F32))


def greedy_decode(enc, lens, pred_sd, joint_sd, hidden, layers, blank, max_symbols) -> List[List[int]]:
    T, N, _ = enc.shape
    out = []
    for n in range(N):
        hyp: List[int] = []
        pred, state = predictor_step(pred_sd, blank, _zero_state(layers, hidden), hidden, layers)
        for t in range(int(lens[n])):
            for _ in range(max_symbols):
                lp = joint_logprobs(joint_sd, enc[t, n], pred)
                k = int(np.argmax(lp))
                if k == blank:
                    break
                hyp.append(k)
                pred, state = predictor_step(pred_sd, k, state, hidden, layers)
        out.append(hyp)
    return out


def beam_decode(enc, lens, pred_sd, joint_sd, hidden, layers, blank, beam_width, max_symbols
                ) -> Tuple[List[List[int]], List[float]]:
    """Time-synchronous beam search: per frame up to ``max_symbols`` rounds; in each round every
    live hypothesis may emit blank (moves to the next frame's set, same-prefix scores merged
    with logaddexp) or a label (the ``beam_width`` best label extensions stay live)."""
    T, N, _ = enc.shape
    results, scores = [], []
    for n in range(N):
        pred0, st0 = predictor_step(pred_sd, blank, _zero_state(layers, hidden), hidden, layers)
        beam = [((), F32(0.0), st0, pred0)]
        for t in range(int(lens[n])):
            A = list(beam)
            B: "Dict[tuple, list]" = {}
            order: List[tuple] = []
            for v in range(max_symbols):
                if not A:
                    break
                lps = [joint_logprobs(joint_sd, enc[t, n], h[3]) for h in A]
                for h, lp in zip(A, lps):
                    s = F32(h[1] + lp[blank])
                    if h[0] in B:
                        B[h[0]][0] = F32(np.logaddexp(np.float64(B[h[0]][0]), np.float64(s)))  # spec: float64, rounded once
                    else:
                        B[h[0]] = [s, h[2], h[3]]
                        order.append(h[0])
                if v == max_symbols - 1:
                    break
                cand = np.stack([(h[1] + lp).astype(F32) for h, lp in zip(A, lps)])  # [len(A), V1]
                cand[:, blank] = -np.inf
                flat = cand.reshape(-1)
                idx = np.argsort(-flat, kind="stable")[:beam_width]
                A_new = []
                for i in idx:
                    if not np.isfinite(flat[i]):
                        continue
                    hi, k = divmod(int(i), cand.shape[1])
                    h = A[hi]
                    pred, st = predictor_step(pred_sd, k, h[2], hidden, layers)
                    A_new.append((h[0] + (k,), F32(flat[i]), st, pred))
                A = A_new
            items = [(p, B[p]) for p in order]
            items.sort(key=lambda it: -float(it[1][0]))  # stable: ties keep first-arrival order
            beam = [(p, v[0], v[1], v[2]) for p, v in items[:beam_width]]
        best = max(range(len(beam)), key=lambda i: (float(beam[i][1]), -i))
        results.append(list(beam[best][0]))
        scores.append(float(beam[best][1]))
    return results, scores
