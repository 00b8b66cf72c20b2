"""Hard-sigmoid / hard-tanh LSTM: mirror of myrtlespeech/model/hard_lstm.py.

Same constructor, I/O contract and state_dict keys
(``rnn.layers.{k}[.fwd|.bwd].cell.{weight_ih,weight_hh,bias_ih,bias_hh}``) as the
reference's TorchScript implementation; the per-timestep matmul/cat loop
(hard_lstm.py:346-379) is replaced by the same HIP kernels as ``RNN`` with the
hard-activation epilogue.  Sequence lengths are ignored (hard_lstm.py:36).
"""
import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.rnn import PackedLayer, run_layers


class HardLSTMCell(torch.nn.Module):
    """Parameter holder with the reference's init (hard_lstm.py:476-511)."""

    def __init__(self, input_size: int, hidden_size: int, forget_gate_bias: Optional[float] = None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.forget_gate_bias = forget_gate_bias
        self.weight_ih = torch.nn.Parameter(torch.randn(4 * hidden_size, input_size))
        self.weight_hh = torch.nn.Parameter(torch.randn(4 * hidden_size, hidden_size))
        self.bias_ih = torch.nn.Parameter(torch.randn(4 * hidden_size))
        self.bias_hh = torch.nn.Parameter(torch.randn(4 * hidden_size))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for weight in self.parameters():
            weight.data.uniform_(-stdv, stdv)
        if self.forget_gate_bias is not None:
            h = self.hidden_size
            self.bias_ih.data[h:2 * h] = self.forget_gate_bias
            self.bias_hh.data[h:2 * h] = 0.0


class HardLSTMLayer(torch.nn.Module):
    def __init__(self, input_size: int, hidden_size: int, forget_gate_bias: Optional[float] = None):
        super().__init__()
        self.cell = HardLSTMCell(input_size, hidden_size, forget_gate_bias)


class HardLSTMBidirLayer(torch.nn.Module):
    def __init__(self, input_size: int, hidden_size: int, forget_gate_bias: Optional[float] = None):
        super().__init__()
        self.fwd = HardLSTMLayer(input_size, hidden_size, forget_gate_bias)
        self.bwd = HardLSTMLayer(input_size, hidden_size, forget_gate_bias)


class StackedLSTM(torch.nn.Module):
    """Layer container (hard_lstm.py:134-250); key prefix ``layers.{k}``."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, bidirectional: bool,
                 forget_gate_bias: Optional[float]):
        super().__init__()
        layer_type = HardLSTMBidirLayer if bidirectional else HardLSTMLayer
        d = 2 if bidirectional else 1
        self.layers = torch.nn.ModuleList(
            [layer_type(input_size, hidden_size, forget_gate_bias)]
            + [layer_type(hidden_size * d, hidden_size, forget_gate_bias) for _ in range(num_layers - 1)])


class HardLSTM(torch.nn.Module):
    """Drop-in replacement for ``RNN(rnn_type=LSTM)`` with hard activations
    (hard_lstm.py:21-129)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 batch_first: bool = False, dropout: float = 0.0, bidirectional: bool = False,
                 forget_gate_bias: Optional[float] = None):
        super().__init__()
        assert not dropout, "Dropout for HardLSTMs is not supported."
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.batch_first = batch_first
        self.num_layers = num_layers
        self.rnn = StackedLSTM(input_size, hidden_size, num_layers, bidirectional, forget_gate_bias)
        self._packed = [PackedLayer() for _ in range(num_layers)]
        self._workspace = _lib.Workspace()
        self.check_status = True
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.rnn = self.rnn.cuda()

    def _layer_params(self):
        out = []
        for layer in self.rnn.layers:
            cells = [layer.fwd.cell, layer.bwd.cell] if self.bidirectional else [layer.cell]
            out.append([(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh) for c in cells])
        return out

    def forward(self, x: Tuple[Tensor, Tensor], hx: Optional[Tuple[Tensor, Tensor]] = None
                ) -> Tuple[Tuple[Tensor, Tensor], Tuple[Tensor, Tensor]]:
        _lib.require_gpu()
        inp, lengths = x
        data = _lib.f32c(inp.transpose(0, 1) if self.batch_first else inp)  # time-major [T, N, In]
        t = data.shape[0]
        h0 = c0 = None
        if hx is not None:
            h0, c0 = _lib.f32c(hx[0]), _lib.f32c(hx[1])
        out, hn, cn = run_layers(_lib.CELL_HARD_LSTM, data, None, t, self._layer_params(), self._packed,
                                 self.hidden_size, h0, c0, self._workspace, self.check_status)
        if self.batch_first:
            out = out.transpose(0, 1)
        return (out, lengths), (hn, cn)
