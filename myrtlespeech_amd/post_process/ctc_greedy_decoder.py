"""CTC greedy decoder: mirror of myrtlespeech/post_process/ctc_greedy_decoder.py."""
from typing import List

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.post_process._common import check_decoder_args


class PendingTranscripts:
    """A greedy decode whose result is on its way to the host."""

    def __init__(self, host, done, batch, keep=None):
        self._host, self._done, self._batch, self._keep = host, done, batch, keep

    def result(self) -> List[List[int]]:
        if self._host is None:
            return [[] for _ in range(self._batch)]
        self._done.synchronize()
        packed = self._host.numpy()
        self._keep = None
        return [packed[n, 1:1 + int(packed[n, 0])].tolist() for n in range(packed.shape[0])]


class CTCGreedyDecoder(torch.nn.Module):
    """Best-path decoding (ctc_greedy_decoder.py:6-97): per-frame argmax (ties ->
    lowest index), then drop repeats and blanks.  One kernel launch and one
    device->host copy per batch instead of a ``.item()`` per frame."""

    def __init__(self, blank_index: int):
        super().__init__()
        self.blank_index = blank_index

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
        return self.launch(x, lengths).result()

    def launch(self, x: torch.Tensor, lengths: torch.Tensor) -> "PendingTranscripts":
        """Enqueue the decode on the current stream and start the copy of its result to pinned host memory WITHOUT waiting
        for it; ``.result()`` waits and builds the lists.  (``forward`` = ``launch(...).result()``; a caller that keeps
        several batches in flight collects the results later.)"""
        seq_len, batch, symbols = check_decoder_args(x, lengths)
        _lib.require_gpu()
        if seq_len == 0 or batch == 0:
            return PendingTranscripts(None, None, batch)
        xd = _lib.f32c(x)
        # column 0 = the utterance's label count, columns 1.. = its labels: ONE read-back
        packed = torch.empty((batch, seq_len + 1), dtype=torch.int32, device="cuda")
        out_len = torch.empty(batch, dtype=torch.int32, device="cuda")
        out_idx = torch.empty((batch, seq_len), dtype=torch.int32, device="cuda")
        lens_dev = _lib.lens_i32(lengths)
        _lib.check(_lib.load().ms_ctc_greedy_decode(_lib.ptr(xd), _lib.ptr(lens_dev), _lib.ptr(out_idx),
                                                    _lib.ptr(out_len), seq_len, batch, symbols, self.blank_index,
                                                    _lib.stream_ptr()), "ms_ctc_greedy_decode")
        packed[:, 0] = out_len
        packed[:, 1:] = out_idx
        host = torch.empty(packed.shape, dtype=torch.int32, pin_memory=True)
        host.copy_(packed, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return PendingTranscripts(host, done, batch, keep=(packed, out_idx, out_len, xd, lens_dev))

    def extra_repr(self) -> str:
        return f"blank_index={self.blank_index}"
