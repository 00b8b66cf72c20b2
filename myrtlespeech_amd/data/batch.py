"""Batch assembly of myrtlespeech/data/batch.py:7-107: ragged samples (sequence axis last) are ordered longest
first -- ``RNN.forward`` packs with ``enforce_sorted`` -- and right-padded into one tensor.  Pure data movement
(torch copies only), so it works on host or device tensors alike; ``shard`` hands each rank its contiguous,
still-sorted slice for the utterance-sharded path."""
from typing import List, Sequence, Tuple

import torch

Sample = Tuple[Tuple[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]


def pad_sequence(sequences: Sequence[torch.Tensor], padding_value: int = 0) -> torch.Tensor:
    """``[*, len_i]`` tensors -> ``[batch, *, max_len]`` filled with ``padding_value`` past each length
    (batch.py:7-42; dtype, device and leading dims are taken from the first tensor)."""
    first = sequences[0]
    longest = max(seq.size(-1) for seq in sequences)
    out = first.new_full((len(sequences), *first.shape[:-1], longest), padding_value)
    for row, seq in zip(out, sequences):
        row[..., :seq.size(-1)] = seq
    return out


def seq_to_seq_collate_fn(batch: List[Sample]):
    """``[((x, x_len), (y, y_len)), ...]`` -> ``((X, X_lens), (Y, Y_lens))`` sorted by ``x.size(-1)``, descending
    (stable: ties keep their order in ``batch``), inputs and targets padded (batch.py:45-107)."""
    ordered = sorted(batch, key=lambda sample: sample[0][0].size(-1), reverse=True)
    xs = pad_sequence([sample[0][0] for sample in ordered])
    x_lens = torch.tensor([sample[0][1] for sample in ordered], requires_grad=False)
    ys = pad_sequence([sample[1][0] for sample in ordered])
    y_lens = torch.tensor([sample[1][1] for sample in ordered], requires_grad=False)
    return (xs, x_lens), (ys, y_lens)
