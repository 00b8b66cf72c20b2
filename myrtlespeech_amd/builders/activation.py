"""Mirror of myrtlespeech/builders/activation.py:5-42."""
import torch


def build(activation_cfg) -> torch.nn.Module:
    kind = activation_cfg.WhichOneof("activation")
    if kind == "identity":
        return torch.nn.Identity()
    if kind == "hardtanh":
        return torch.nn.Hardtanh(min_val=activation_cfg.hardtanh.min_val, max_val=activation_cfg.hardtanh.max_val)
    if kind == "relu":
        return torch.nn.ReLU()
    raise ValueError(f"unsupported activation_cfg {activation_cfg}")
