"""Mirror of myrtlespeech/builders/fully_connected.py:9-75."""
import torch

from myrtlespeech_amd.builders.activation import build as build_activation
from myrtlespeech_amd.model.fully_connected import FullyConnected


def build(fully_connected_cfg, input_features: int, output_features: int) -> FullyConnected:
    activation = build_activation(fully_connected_cfg.activation)
    if isinstance(activation, torch.nn.Identity):
        activation = None
    hidden_size = fully_connected_cfg.hidden_size if fully_connected_cfg.hidden_size > 0 else None
    dropout = fully_connected_cfg.dropout.value if fully_connected_cfg.HasField("dropout") else None
    return FullyConnected(in_features=input_features, out_features=output_features,
                          num_hidden_layers=fully_connected_cfg.num_hidden_layers, hidden_size=hidden_size,
                          hidden_activation_fn=activation, dropout=dropout)
