"""Deep Speech 1: mirror of myrtlespeech/model/deep_speech_1.py.

Same constructor and state_dict keys (``fc{1..4}.0.*``, ``bi_lstm.rnn.*``,
``out.*``); three clipped-ReLU Linear layers, one bidirectional (hard) LSTM, a
fourth clipped-ReLU layer and the output Linear, each Linear a fused MFMA GEMM.
"""
from typing import Optional, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.fully_connected import linear_stack_plan, run_linear_stack
from myrtlespeech_amd.model.hard_lstm import HardLSTM
from myrtlespeech_amd.model.rnn import RNN, RNNState, RNNType


class DeepSpeech1(torch.nn.Module):
    """`Deep Speech 1 <https://arxiv.org/abs/1412.5567>`_ with the paper's
    recurrent layer replaced by an LSTM (deep_speech_1.py:13-188)."""

    def __init__(self, input_features: int, input_channels: int, n_hidden: int, out_features: int, drop_prob: float,
                 relu_clip: float = 20.0, forget_gate_bias: float = 1.0, hard_lstm: bool = False):
        super().__init__()
        self.input_features = input_features
        self.input_channels = input_channels
        self.use_cuda = torch.cuda.is_available()
        self._relu_clip = float(relu_clip)
        self._drop_prob = drop_prob
        # single-clip serving switch (not in the reference): lets the wide exact-f32 layers take K slices
        # (MS_LINEAR_FEW_ROWS).  Off by default: with it a clip's rounding differs from the same clip inside a batch.
        self.few_rows = False
        self.fc1 = self._fully_connected(input_features * input_channels, n_hidden)
        self.fc2 = self._fully_connected(n_hidden, n_hidden)
        self.fc3 = self._fully_connected(n_hidden, 2 * n_hidden)
        lstm_kwargs = dict(input_size=2 * n_hidden, hidden_size=n_hidden, num_layers=1, bias=True, bidirectional=True,
                           forget_gate_bias=forget_gate_bias, batch_first=True)
        self.bi_lstm = HardLSTM(**lstm_kwargs) if hard_lstm else RNN(rnn_type=RNNType.LSTM, **lstm_kwargs)
        self.fc4 = self._fully_connected(2 * n_hidden, n_hidden)
        self.out = self._fully_connected(n_hidden, out_features, relu=False, dropout=False)
        if self.use_cuda:
            self.cuda()

    def _fully_connected(self, in_f: int, out_f: int, relu: bool = True, dropout: bool = True) -> torch.nn.Module:
        layers = [torch.nn.Linear(in_f, out_f)]
        if relu:
            layers.append(torch.nn.Hardtanh(0.0, self._relu_clip, inplace=True))
        if dropout:
            layers.append(torch.nn.Dropout(p=self._drop_prob))
        return layers[0] if len(layers) == 1 else torch.nn.Sequential(*layers)

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor], hx: Optional[RNNState] = None
                ) -> Tuple[Tuple[torch.Tensor, torch.Tensor], RNNState]:
        """``[batch, channels, features, seq_len] -> ([seq_len, batch, out_features], lens), hid``."""
        _lib.require_gpu()
        h, seq_lens = x
        h = _lib.f32c(h)
        n, c, f, t = h.shape
        # (N, C*F, T) -> (T, N, C*F): rows of the Linear layers are independent, so run
        # them time-major (what the LSTM kernel wants) and skip deep_speech_1.py:176's permute
        tnf = torch.empty((t, n, c * f), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ms_nct_to_tnc(_lib.ptr(h), _lib.ptr(tnf), n, c * f, t, _lib.stream_ptr()),
                   "ms_nct_to_tnc")
        plan = (linear_stack_plan(self.fc1, self.training) + linear_stack_plan(self.fc2, self.training)
                + linear_stack_plan(self.fc3, self.training))
        h = run_linear_stack(tnf.reshape(t * n, c * f), plan, few_rows=self.few_rows).reshape(t, n, -1)
        # bi_lstm is batch_first; hand it the batch-major *view* of the time-major buffer
        (h, _), hid = self.bi_lstm((h.transpose(0, 1), seq_lens), hx)
        h = h.transpose(0, 1)  # back to the time-major storage the kernel wrote
        plan = linear_stack_plan(self.fc4, self.training) + linear_stack_plan(self.out, self.training)
        h = _lib.f32c(h)
        out = run_linear_stack(h.reshape(t * n, h.shape[-1]), plan, few_rows=self.few_rows).reshape(t, n, -1)
        return (out, _lib.lens_to_device(seq_lens)), hid
