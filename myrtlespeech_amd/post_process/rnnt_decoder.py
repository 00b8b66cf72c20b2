"""RNN-T greedy and time-synchronous beam decoding -- OWN specification (see model/rnnt.py;
the reference has no transducer).  The per-step arithmetic (embedding gather, predictor LSTM
step, projections, fused tanh-joint + log-softmax, top-k prune) runs on the GPU for every live
hypothesis of every utterance at once; hypothesis book-keeping (prefix tuples, same-prefix
merging of blank transitions) is host logic.
"""
import math
from typing import Dict, List, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor


class RNNTGreedyDecoder(torch.nn.Module):
    """Per frame: emit the most likely symbol until blank (at most ``max_symbols`` per frame)."""

    def __init__(self, predictor: RNNTPredictor, joint: RNNTJoint, max_symbols: int = 4):
        super().__init__()
        self.predictor, self.joint, self.max_symbols = predictor, joint, max_symbols

    def forward(self, enc: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
        _lib.require_gpu()
        t_max, n, _ = enc.shape
        blank = self.predictor.blank
        enc_p = self.joint.project_encoder(enc)
        lens_l = lens.detach().cpu().tolist()
        hyps: List[List[int]] = [[] for _ in range(n)]
        state = self.predictor.zero_state(n)
        pred, state = self.predictor.step(torch.full((n,), blank, dtype=torch.int64), state)
        for t in range(t_max):
            live = [i for i in range(n) if lens_l[i] > t]  # still emitting at this frame
            for _ in range(self.max_symbols):
                if not live:
                    break
                rows = torch.tensor(live, dtype=torch.int64, device="cuda")
                logp = self.joint.logprobs(enc_p, rows + t * n, pred[rows])
                k = logp.argmax(dim=1).cpu().tolist()  # torch.argmax: first maximum, like the oracle
                emit = [(i, ki) for i, ki in zip(live, k) if ki != blank]
                if not emit:
                    break
                er = torch.tensor([i for i, _ in emit], dtype=torch.int64, device="cuda")
                sub = (state[0][:, er].contiguous(), state[1][:, er].contiguous())
                p_new, s_new = self.predictor.step(torch.tensor([ki for _, ki in emit]), sub)
                pred = pred.clone()
                pred[er] = p_new
                state = (state[0].clone(), state[1].clone())
                state[0][:, er] = s_new[0]
                state[1][:, er] = s_new[1]
                for i, ki in emit:
                    hyps[i].append(ki)
                live = [i for i, _ in emit]
        return hyps


class RNNTBeamDecoder(torch.nn.Module):
    """Time-synchronous beam search of width ``beam_width`` (``max_symbols`` emission rounds per
    frame); see ``oracle/rnnt_oracle.py::beam_decode`` for the reference statement."""

    def __init__(self, predictor: RNNTPredictor, joint: RNNTJoint, beam_width: int = 8, max_symbols: int = 3):
        super().__init__()
        if beam_width <= 0:
            raise ValueError(f"beam_width={beam_width} must be > 0")
        self.predictor, self.joint = predictor, joint
        self.beam_width, self.max_symbols = beam_width, max_symbols

    def _topk(self, scores: torch.Tensor, k: int):
        b, c = scores.shape
        idx = torch.empty((b, k), dtype=torch.int32, device="cuda")
        val = torch.empty((b, k), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ms_rnnt_topk(_lib.ptr(scores), _lib.ptr(idx), _lib.ptr(val), b, c, k, _lib.stream_ptr()),
                   "ms_rnnt_topk")
        return idx.cpu().tolist(), val.cpu().tolist()

    def forward(self, enc: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
        _lib.require_gpu()
        t_max, n, _ = enc.shape
        w, blank, v1 = self.beam_width, self.predictor.blank, self.predictor.vocab_size + 1
        enc_p = self.joint.project_encoder(enc)
        lens_l = lens.detach().cpu().tolist()
        # a hypothesis = (prefix, score, slot); slot indexes the device-side state / predictor output pools
        st = self.predictor.zero_state(n)
        pred_pool, st = self.predictor.step(torch.full((n,), blank, dtype=torch.int64), st)
        h_pool, c_pool = st
        beams: List[List[Tuple[tuple, float, int]]] = [[((), 0.0, i)] for i in range(n)]
        for t in range(t_max):
            active = [i for i in range(n) if lens_l[i] > t]
            if not active:
                break
            A: Dict[int, List[Tuple[tuple, float, int]]] = {i: list(beams[i]) for i in active}
            B: Dict[int, Dict[tuple, List]] = {i: {} for i in active}
            order: Dict[int, List[tuple]] = {i: [] for i in active}
            for v in range(self.max_symbols):
                flat = [(i, h) for i in active for h in A[i]]
                if not flat:
                    break
                slots = torch.tensor([h[2] for _, h in flat], dtype=torch.int64, device="cuda")
                rows = torch.tensor([t * n + i for i, _ in flat], dtype=torch.int64, device="cuda")
                logp = self.joint.logprobs(enc_p, rows, pred_pool[slots])
                base = torch.tensor([h[1] for _, h in flat], dtype=torch.float32, device="cuda")
                total = base[:, None] + logp                      # [R, V1] float32
                blank_scores = total[:, blank].cpu().tolist()
                for (i, h), s in zip(flat, blank_scores):
                    if h[0] in B[i]:
                        B[i][h[0]][0] = _logaddexp32(B[i][h[0]][0], s)
                    else:
                        B[i][h[0]] = [s, h[2]]
                        order[i].append(h[0])
                if v == self.max_symbols - 1:
                    break
                total[:, blank] = float("-inf")
                # per utterance top-w over its (hypothesis, label) candidates, padded to a rectangle
                cand = torch.full((len(active), w * v1), float("-inf"), dtype=torch.float32, device="cuda")
                starts, r0 = {}, 0
                for bi, i in enumerate(active):
                    cnt = len(A[i])
                    if cnt:
                        cand[bi, :cnt * v1] = total[r0:r0 + cnt].reshape(-1)
                    starts[i] = r0
                    r0 += cnt
                idx, val = self._topk(cand, w)
                new_A: Dict[int, List] = {i: [] for i in active}
                ext_labels, ext_src = [], []
                for bi, i in enumerate(active):
                    for j in range(w):
                        if idx[bi][j] < 0 or not math.isfinite(val[bi][j]):
                            continue
                        hi, k = divmod(idx[bi][j], v1)
                        h = A[i][hi]
                        new_A[i].append([h[0] + (k,), val[bi][j], None])
                        ext_labels.append(k)
                        ext_src.append(h[2])
                if ext_labels:
                    src = torch.tensor(ext_src, dtype=torch.int64, device="cuda")
                    p_new, (h_new, c_new) = self.predictor.step(torch.tensor(ext_labels),
                                                                (h_pool[:, src].contiguous(), c_pool[:, src].contiguous()))
                    base_slot = pred_pool.shape[0]
                    pred_pool = torch.cat([pred_pool, p_new], 0)
                    h_pool = torch.cat([h_pool, h_new], 1)
                    c_pool = torch.cat([c_pool, c_new], 1)
                    q = 0
                    for i in active:
                        for hyp in new_A[i]:
                            hyp[2] = base_slot + q
                            q += 1
                A = {i: [tuple(h) for h in new_A[i]] for i in active}
            keep = set()
            for i in active:
                items = [(p, B[i][p]) for p in order[i]]
                items.sort(key=lambda it: -it[1][0])  # stable
                beams[i] = [(p, s, slot) for p, (s, slot) in items[:w]]
                keep.update(slot for _, _, slot in beams[i])
            for i in range(n):
                if i not in active:
                    keep.update(slot for _, _, slot in beams[i])
            # compact the pools to the surviving hypotheses
            keep_l = sorted(keep)
            remap = {s: j for j, s in enumerate(keep_l)}
            sel = torch.tensor(keep_l, dtype=torch.int64, device="cuda")
            pred_pool, h_pool, c_pool = pred_pool[sel], h_pool[:, sel].contiguous(), c_pool[:, sel].contiguous()
            beams = [[(p, s, remap[slot]) for p, s, slot in b] for b in beams]
        out = []
        for b in beams:
            best = max(range(len(b)), key=lambda j: (b[j][1], -j))
            out.append(list(b[best][0]))
        return out


def _logaddexp32(a: float, b: float) -> float:
    """float32 logaddexp, matching numpy.logaddexp on float32 operands."""
    import numpy as np
    return float(np.logaddexp(np.float32(a), np.float32(b)))
