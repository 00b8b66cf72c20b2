"""Mirror of myrtlespeech/model/utils.py:1-26."""
from typing import Optional, Tuple


def activation_clamp(module) -> Optional[Tuple[float, float]]:
    """(lo, hi) of the clamp an activation module computes, or None for Identity.

    The reference builds exactly three activations (builders/activation.py:31-42):
    Identity, Hardtanh(min_val, max_val) and ReLU."""
    import torch

    if module is None or isinstance(module, torch.nn.Identity):
        return None
    if isinstance(module, torch.nn.Hardtanh):
        return float(module.min_val), float(module.max_val)
    if isinstance(module, torch.nn.ReLU):
        return 0.0, float("inf")
    raise NotImplementedError(f"activation {module!r} has no HIP epilogue (Identity, Hardtanh and ReLU do)")


def lookahead_to_fully_connected_size(x):
    """(batch, features, seq_len) -> (batch, seq_len, features), utils.py in the reference."""
    return x.transpose(1, 2)
