"""CTC greedy decoder: mirror of myrtlespeech/post_process/ctc_greedy_decoder.py."""
from typing import List

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.post_process._common import check_decoder_args, ragged_to_lists


class CTCGreedyDecoder(torch.nn.Module):
    """Best-path decoding (ctc_greedy_decoder.py:6-97): per-frame argmax (ties ->
    lowest index), then drop repeats and blanks.  One kernel launch and one
    device->host copy per batch instead of a ``.item()`` per frame."""

    def __init__(self, blank_index: int):
        super().__init__()
        self.blank_index = blank_index

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
        seq_len, batch, symbols = check_decoder_args(x, lengths)
        _lib.require_gpu()
        if seq_len == 0 or batch == 0:
            return [[] for _ in range(batch)]
        xd = _lib.f32c(x)
        out_idx = torch.empty((batch, seq_len), dtype=torch.int32, device="cuda")
        out_len = torch.empty(batch, dtype=torch.int32, device="cuda")
        lens_dev = _lib.lens_i32(lengths)
        _lib.check(_lib.load().ms_ctc_greedy_decode(_lib.ptr(xd), _lib.ptr(lens_dev), _lib.ptr(out_idx),
                                                    _lib.ptr(out_len), seq_len, batch, symbols, self.blank_index,
                                                    _lib.stream_ptr()), "ms_ctc_greedy_decode")
        return ragged_to_lists(out_idx, out_len)

    def extra_repr(self) -> str:
        return f"blank_index={self.blank_index}"
