"""RNN-T (transducer) modules -- this repository's OWN specification.

The reference snapshot contains no transducer, so nothing here mirrors a
reference file; BASELINE.json's configs[3] names the shape (DS2 encoder + 2-layer LSTM
predictor + joint network, beam width 8).  Specification (Graves 2012, "Sequence Transduction
with Recurrent Neural Networks"):

* encoder: any module returning ``((enc[T', N, E], lens), hid)`` -- normally ``DeepSpeech2``;
* ``RNNTPredictor``: ``Embedding(V + 1, D)`` (index ``V`` = blank = start-of-sequence) followed by
  an ``L``-layer unidirectional LSTM (``model.rnn.RNN``);
* ``RNNTJoint``: ``log_softmax(out(tanh(enc_proj(enc_t) + pred_proj(pred_u))))`` over ``V + 1``
  symbols, blank last.
"""
from typing import Optional, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.rnn import RNN, RNNType


class RNNTPredictor(torch.nn.Module):
    def __init__(self, vocab_size: int, embed_dim: int, hidden_size: int, num_layers: int = 2):
        super().__init__()
        self.vocab_size = vocab_size
        self.blank = vocab_size
        self.embedding = torch.nn.Embedding(vocab_size + 1, embed_dim)  # parameter container
        self.rnn = RNN(RNNType.LSTM, embed_dim, hidden_size, num_layers=num_layers)
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        if torch.cuda.is_available():
            self.cuda()

    def zero_state(self, rows: int):
        z = torch.zeros(self.num_layers, rows, self.hidden_size, device="cuda")
        return z, z.clone()

    def step(self, labels: torch.Tensor, state: Tuple[torch.Tensor, torch.Tensor]):
        """labels [R] int (device) -> (pred_out [R, hidden], new_state); one symbol per row."""
        _lib.require_gpu()
        r = labels.numel()
        w = _lib.f32c(self.embedding.weight.detach())
        idx = labels.to(device="cuda", dtype=torch.int32).contiguous()
        emb = torch.empty((1, r, w.shape[1]), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ms_embedding_forward(_lib.ptr(w), _lib.ptr(idx), _lib.ptr(emb), r, w.shape[1], w.shape[0],
                                                    _lib.stream_ptr()), "ms_embedding_forward")
        (out, _), new_state = self.rnn((emb, torch.ones(r, dtype=torch.int64)), state)
        return out[0], new_state


class RNNTJoint(torch.nn.Module):
    def __init__(self, enc_features: int, pred_features: int, joint_features: int, vocab_size: int):
        super().__init__()
        self.enc_proj = torch.nn.Linear(enc_features, joint_features)
        self.pred_proj = torch.nn.Linear(pred_features, joint_features, bias=False)
        self.out = torch.nn.Linear(joint_features, vocab_size + 1)
        self.vocab_size = vocab_size
        if torch.cuda.is_available():
            self.cuda()

    @staticmethod
    def _linear(x2d: torch.Tensor, lin: torch.nn.Linear) -> torch.Tensor:
        lib = _lib.load()
        m, k = x2d.shape
        y = torch.empty((m, lin.out_features), dtype=torch.float32, device="cuda")
        w = _lib.f32c(lin.weight.detach())
        b = None if lin.bias is None else _lib.f32c(lin.bias.detach())
        _lib.check(lib.ms_linear_forward(_lib.ptr(x2d), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), m, k, lin.out_features,
                                         _lib.ACT_NONE, 0.0, 0.0, _lib.stream_ptr()), "ms_linear_forward")
        return y

    def project_encoder(self, enc: torch.Tensor) -> torch.Tensor:
        """enc [T, N, E] -> [T*N, J] (done once per batch)."""
        t, n, e = enc.shape
        return self._linear(_lib.f32c(enc).reshape(t * n, e), self.enc_proj)

    def logprobs(self, enc_p: torch.Tensor, enc_rows: torch.Tensor, pred_out: torch.Tensor) -> torch.Tensor:
        """log P(symbol | frame, prefix) for R hypothesis rows: enc_rows [R] indexes rows of ``enc_p``."""
        r = pred_out.shape[0]
        pred_p = self._linear(_lib.f32c(pred_out), self.pred_proj)
        rows = enc_rows.to(device="cuda", dtype=torch.int32).contiguous()
        w = _lib.f32c(self.out.weight.detach())
        b = _lib.f32c(self.out.bias.detach())
        logp = torch.empty((r, w.shape[0]), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ms_rnnt_joint_forward(_lib.ptr(enc_p), _lib.ptr(rows), _lib.ptr(pred_p), _lib.ptr(w),
                                                     _lib.ptr(b), _lib.ptr(logp), r, w.shape[1], w.shape[0],
                                                     _lib.stream_ptr()), "ms_rnnt_joint_forward")
        return logp


class RNNT(torch.nn.Module):
    """Encoder + prediction network + joint network."""

    def __init__(self, encoder: torch.nn.Module, predictor: RNNTPredictor, joint: RNNTJoint):
        super().__init__()
        self.encoder = encoder
        self.predictor = predictor
        self.joint = joint

    def encode(self, x: Tuple[torch.Tensor, torch.Tensor], hx: Optional[object] = None):
        (enc, lens), _ = self.encoder(x, hx)
        return enc, lens
