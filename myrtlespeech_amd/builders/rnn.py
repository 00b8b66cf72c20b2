"""Mirror of myrtlespeech/builders/rnn.py:9-95."""
from typing import Tuple

from myrtlespeech_amd.model.rnn import RNN, RNNType

_TYPES = {0: RNNType.LSTM, 1: RNNType.GRU, 2: RNNType.BASIC_RNN}


def build(rnn_cfg, input_features: int, batch_first: bool = False) -> Tuple[RNN, int]:
    """Returns ``(RNN, output feature count)`` for an ``RNN`` config."""
    if rnn_cfg.rnn_type not in _TYPES:
        raise ValueError(f"rnn_type={rnn_cfg.rnn_type} not supported")
    fgb = rnn_cfg.forget_gate_bias.value if rnn_cfg.HasField("forget_gate_bias") else None
    rnn = RNN(rnn_type=_TYPES[rnn_cfg.rnn_type], input_size=input_features, hidden_size=rnn_cfg.hidden_size,
              num_layers=rnn_cfg.num_layers, bias=rnn_cfg.bias, bidirectional=rnn_cfg.bidirectional,
              forget_gate_bias=fgb, batch_first=batch_first)
    return rnn, rnn_cfg.hidden_size * (2 if rnn_cfg.bidirectional else 1)
