"""RNN-T greedy and time-synchronous beam decoding -- OWN specification (see model/rnnt.py and
oracle/rnnt_oracle.py; the reference has no transducer).  The whole decode of a batch is ONE call into the
library (``ms_rnnt_decode``): the frame loop, the hypothesis lists, the prefix trie that gives merged blank
transitions their identity, the predictor-state pool and the top-k pruning all live on the device; the beam reads
nothing back until the label lists are fetched at the end.  The greedy decode is event-driven (the predictor only
steps after a label; 32 frames per utterance are scored against the current prediction at once), so its launch
count depends on the data: the library fetches a 4-byte counter of finished utterances every second iteration,
asynchronously and two checks behind the queue.
"""
import ctypes
from typing import List, Optional

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor


def _decode(predictor: RNNTPredictor, joint: RNNTJoint, enc: torch.Tensor, lens: torch.Tensor, beam_width: int,
            max_symbols: int, greedy: bool, workspace: _lib.Workspace, want_scores: bool = False):
    _lib.require_gpu()
    lib = _lib.load()
    t_max, n, _ = enc.shape
    lens_host = lens.detach().cpu().to(torch.int64)
    if lens_host.numel() != n:
        raise ValueError(f"lengths batch {lens_host.numel()} != encoder batch {n}")
    if n and int(lens_host.max()) > t_max:
        raise ValueError("a length exceeds the number of encoder frames")
    steps = int(lens_host.max()) if n else 0
    if n == 0 or steps == 0:
        return [[] for _ in range(n)], [0.0] * n
    enc_p = joint.project_encoder(enc)                       # [T*N, J], once per batch
    rnn = predictor.rnn.rnn                                  # torch.nn.LSTM used as the parameter container
    L, H, D, V = predictor.num_layers, predictor.hidden_size, predictor.embedding.weight.shape[1], predictor.vocab_size
    J = joint.out.weight.shape[1]
    keep = []

    def col(name: str):
        ts = [getattr(rnn, f"{name}_l{l}", None) for l in range(L)]
        ts = [None if t is None else _lib.f32c(t.detach()) for t in ts]
        keep.append(ts)
        arr = (ctypes.c_void_p * L)(*[None if t is None else t.data_ptr() for t in ts])
        keep.append(arr)
        return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))

    emb = _lib.f32c(predictor.embedding.weight.detach())
    w_pred = _lib.f32c(joint.pred_proj.weight.detach())
    w_out = _lib.f32c(joint.out.weight.detach())
    b_out = _lib.f32c(joint.out.bias.detach())
    lens_d = _lib.lens_i32(lens_host)
    stride = steps * max_symbols if greedy else steps * max(max_symbols - 1, 0) + 1
    out_idx = torch.zeros((n, stride), dtype=torch.int32, device="cuda")
    out_len = torch.zeros((n,), dtype=torch.int32, device="cuda")
    out_score = torch.zeros((n,), dtype=torch.float32, device="cuda")
    w = 1 if greedy else beam_width
    nbytes = lib.ms_rnnt_decode_workspace_bytes(steps, n, V, D, H, L, J, w, max_symbols, int(greedy))
    if nbytes == 0:
        raise ValueError("unsupported RNN-T decode shape (beam_width <= 32, at most 8 predictor layers)")
    ws = workspace.get(nbytes)
    _lib.check(lib.ms_rnnt_decode(_lib.ptr(enc_p), _lib.ptr(lens_d), _lib.ptr(emb), col("weight_ih"), col("weight_hh"),
                                  col("bias_ih"), col("bias_hh"), _lib.ptr(w_pred), _lib.ptr(w_out), _lib.ptr(b_out),
                                  _lib.ptr(out_idx), _lib.ptr(out_len), _lib.ptr(out_score), steps, n, V, D, H, L, J, w,
                                  max_symbols, int(greedy), _lib.ptr(ws), nbytes, _lib.stream_ptr()), "ms_rnnt_decode")
    lens_out = out_len.cpu().tolist()                         # the one synchronisation of the decode
    idx = out_idx.cpu()
    hyps = [idx[i, :min(lens_out[i], stride)].tolist() for i in range(n)]
    return hyps, (out_score.cpu().tolist() if want_scores else None)


class RNNTGreedyDecoder(torch.nn.Module):
    """Per frame: emit the most likely symbol until blank (at most ``max_symbols`` per frame)."""

    def __init__(self, predictor: RNNTPredictor, joint: RNNTJoint, max_symbols: int = 4):
        super().__init__()
        if max_symbols <= 0:
            raise ValueError(f"max_symbols={max_symbols} must be > 0")
        self.predictor, self.joint, self.max_symbols = predictor, joint, max_symbols
        self._ws = _lib.Workspace()

    def forward(self, enc: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
        return _decode(self.predictor, self.joint, enc, lens, 1, self.max_symbols, True, self._ws)[0]


class RNNTBeamDecoder(torch.nn.Module):
    """Time-synchronous beam search of width ``beam_width`` (``max_symbols`` emission rounds per
    frame); see ``oracle/rnnt_oracle.py::beam_decode`` for the statement of the algorithm."""

    def __init__(self, predictor: RNNTPredictor, joint: RNNTJoint, beam_width: int = 8, max_symbols: int = 3):
        super().__init__()
        if beam_width <= 0:
            raise ValueError(f"beam_width={beam_width} must be > 0")
        if max_symbols <= 0:
            raise ValueError(f"max_symbols={max_symbols} must be > 0")
        self.predictor, self.joint = predictor, joint
        self.beam_width, self.max_symbols = beam_width, max_symbols
        self._ws = _lib.Workspace()
        self.last_scores: Optional[List[float]] = None

    def forward(self, enc: torch.Tensor, lens: torch.Tensor) -> List[List[int]]:
        hyps, self.last_scores = _decode(self.predictor, self.joint, enc, lens, self.beam_width, self.max_symbols, False,
                                         self._ws, want_scores=True)
        return hyps
