"""Deep Speech 2 composition: mirror of myrtlespeech/model/deep_speech_2.py.

``DeepSpeech2(cnn, rnn, lookahead, fully_connected)`` takes the same sub-modules
the reference's builder assembles (builders/deep_speech_2.py:105-221) and returns
``((logits[T,N,V], lens), hid)``.  The forward pass keeps every tensor on the GPU
and fuses what the reference does as separate torch ops:

* ``MaskConv -> SeqLenWrapper(Hardtanh/ReLU)`` pairs run as one conv kernel with
  the clamp in its epilogue;
* ``(N,C,F,T) -> (T,N,C*F)`` is one tiled transpose kernel;
* the fully connected stack consumes the RNN output in its native ``[T,N,F]``
  order (rows are independent), so neither ``transpose(0, 1)`` is materialised;
* the lookahead kernel reads ``[T,N,F]`` through strides.
"""
from typing import Optional, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.cnn import Conv1dTo2d, Conv2dTo1d, MaskConv1d, MaskConv2d
from myrtlespeech_amd.model.fully_connected import FullyConnected, linear_stack_plan, run_linear_stack
from myrtlespeech_amd.model.lookahead import Lookahead, lookahead_apply
from myrtlespeech_amd.model.rnn import RNNState
from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
from myrtlespeech_amd.model.utils import activation_clamp


def _is_plain_activation(m) -> bool:
    return (isinstance(m, SeqLenWrapper) and isinstance(m.module, (torch.nn.Hardtanh, torch.nn.ReLU, torch.nn.Identity))
            and isinstance(m.seq_lens_fn, torch.nn.Identity))


class DeepSpeech2(torch.nn.Module):
    """`Deep Speech 2 <http://proceedings.mlr.press/v48/amodei16.pdf>`_
    (deep_speech_2.py:8-172); argument contract as in the reference."""

    def __init__(self, cnn: Optional[torch.nn.Module], rnn: torch.nn.Module, lookahead: Optional[torch.nn.Module],
                 fully_connected: torch.nn.Module):
        super().__init__()
        self.cnn = cnn
        self.rnn = rnn
        self.lookahead = lookahead
        self.fully_connected = fully_connected
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.cuda()

    # ------------------------------------------------------------------ stages
    def _run_cnn(self, h):
        mods = list(self.cnn) if isinstance(self.cnn, torch.nn.Sequential) else [self.cnn]
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (MaskConv1d, MaskConv2d)) and i + 1 < len(mods) and _is_plain_activation(mods[i + 1]):
                h = m(h, fused_activation=activation_clamp(mods[i + 1].module))
                i += 2
            else:
                h = m(h)
                i += 1
        return h

    @staticmethod
    def _conv_to_rnn_size(x: torch.Tensor) -> torch.Tensor:
        """(batch, chnls, feature, seq_len) -> (seq_len, batch, chnls*feature), deep_speech_2.py:114-117."""
        n, c, f, t = x.shape
        x = _lib.f32c(x)
        y = torch.empty((t, n, c * f), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ms_nct_to_tnc(_lib.ptr(x), _lib.ptr(y), n, c * f, t, _lib.stream_ptr()), "ms_nct_to_tnc")
        return y

    def _run_lookahead_ntf(self, h_tnf: torch.Tensor):
        """[T,N,F] RNN output -> lookahead (+activation) -> [N,T,F], or None if the
        lookahead module is not the builder's Sequential(Lookahead, SeqLenWrapper(act))."""
        la = self.lookahead
        mods = list(la) if isinstance(la, torch.nn.Sequential) else [la]
        if not isinstance(mods[0], Lookahead) or len(mods) > 2:
            return None
        clamp = None
        if len(mods) == 2:
            if not _is_plain_activation(mods[1]):
                return None
            clamp = activation_clamp(mods[1].module)
        t, n, f = h_tnf.shape
        return lookahead_apply(h_tnf, mods[0].weight, (f, 1, n * f), n, f, t, out_layout="ntf", clamp=clamp)

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tuple[torch.Tensor, torch.Tensor], hx: Optional[RNNState] = None
                ) -> Tuple[Tuple[torch.Tensor, torch.Tensor], RNNState]:
        return self.back(self.front(x), hx)

    def front(self, x: Tuple[torch.Tensor, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """The part of ``forward`` that depends on the input only: convolutions and the (N,C,F,T) -> (T,N,C*F) re-layout
        (deep_speech_2.py:150-157).  ``forward(x, hx) == back(front(x), hx)``; a pipeline may run ``front`` of a later batch
        on another stream while an earlier batch is in its recurrent layers (``pipeline.BatchesInFlight``)."""
        _lib.require_gpu()
        h = (x[0].cuda() if not x[0].is_cuda else x[0], x[1])
        if self.cnn is not None:
            h = self._run_cnn(h)
        return self._conv_to_rnn_size(h[0]), h[1]

    def back(self, h: Tuple[torch.Tensor, torch.Tensor], hx: Optional[RNNState] = None
             ) -> Tuple[Tuple[torch.Tensor, torch.Tensor], RNNState]:
        """Recurrent layers, lookahead and fully-connected layers on the output of ``front`` (deep_speech_2.py:158-172)."""
        return self.output(*self.recurrent(h, hx))

    def recurrent(self, h: Tuple[torch.Tensor, torch.Tensor], hx: Optional[RNNState] = None):
        """The recurrent stack on the output of ``front``: ``(seq [T, N, D*H], lens, hid)``."""
        _lib.at_issue_point()
        h, hid = self.rnn(h, hx=hx)
        _lib.at_issue_point()
        return h[0], h[1], hid

    def output(self, seq: torch.Tensor, lens: torch.Tensor, hid: RNNState
               ) -> Tuple[Tuple[torch.Tensor, torch.Tensor], RNNState]:
        """Lookahead and fully-connected layers on the recurrent stack's output (the part of ``forward`` that a pipeline may
        issue late: nothing of the NEXT batch depends on it)."""
        fc = self.fully_connected
        fused_fc = isinstance(fc, FullyConnected)
        if self.lookahead is not None:
            ntf = self._run_lookahead_ntf(seq) if seq.is_contiguous() else None
            if ntf is None:  # foreign lookahead module: follow the reference's permutes literally
                la_out, lens = self.lookahead((seq.permute(1, 2, 0), lens))
                ntf = la_out.transpose(1, 2)
            out, lens = fc((ntf, lens))
            return (out.transpose(0, 1), lens), hid
        if fused_fc:
            t, n, f = seq.shape
            y = run_linear_stack(_lib.f32c(seq).reshape(t * n, f), linear_stack_plan(fc.fully_connected, fc.training))
            return (y.reshape(t, n, -1), _lib.lens_to_device(lens)), hid
        out, lens = fc((seq.transpose(0, 1), lens))
        return (out.transpose(0, 1), lens), hid
