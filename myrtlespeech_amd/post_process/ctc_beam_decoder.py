"""CTC prefix beam search: mirror of myrtlespeech/post_process/ctc_beam_decoder.py.

Same constructor checks and ``forward(x, lengths) -> List[List[int]]`` as the
reference.  The search itself (Hannun et al. 2014 prefix beam search in linear
float32 arithmetic, ctc_beam_decoder.py:175-258) runs on the GPU, one workgroup
per utterance, and reproduces the reference's beam bit for bit (visiting order,
Counter merge order, stable sort, float32 rounding).

A ``language_model`` is a host callable, so with one set the kernel is advanced a
frame at a time and the host supplies, per beam entry, the factor
``float32(lm(prefix + (separator,)) ** lm_weight)`` the reference multiplies in at
ctc_beam_decoder.py:222-228; without one the whole utterance is a single launch.
"""
from typing import Callable, List, Optional, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.post_process._common import check_decoder_args, ragged_to_lists


class CTCBeamDecoder(torch.nn.Module):
    """ctc_beam_decoder.py:10-273."""

    def __init__(self, blank_index: int, beam_width: int, prune_threshold: float = 0.001,
                 language_model: Optional[Callable[[Tuple[int, ...]], float]] = None,
                 lm_weight: Optional[float] = None, separator_index: Optional[int] = None, word_weight: float = 1.0):
        if blank_index < 0:
            raise ValueError(f"blank_index={blank_index} must be >= 0")
        if beam_width <= 0:
            raise ValueError(f"beam_width={beam_width} must be > 0")
        if prune_threshold < 0.0 or prune_threshold > 1.0:
            raise ValueError(f"prune_threshold={prune_threshold} not in [0.0, 1.0]")
        if language_model is not None and lm_weight is None:
            raise ValueError("lm_weight must be set when using language_model")
        if separator_index is not None and separator_index < 0:
            raise ValueError(f"separator_index={separator_index} must be >= 0")
        super().__init__()
        self.blank_index = blank_index
        self.beam_width = beam_width
        self.prune_threshold = prune_threshold
        self.language_model = language_model
        self.lm_weight = lm_weight
        self.separator_index = separator_index
        self.word_weight = word_weight
        self._workspace = _lib.Workspace()

    def _word_factor(self, seq_len: int) -> Optional[torch.Tensor]:
        """float32((1 + n_words) ** word_weight), n_words = 0..seq_len+1, evaluated
        on the host exactly like the sort key at ctc_beam_decoder.py:248-253."""
        if self.separator_index is None:
            return None
        vals = [float((1 + n) ** self.word_weight) for n in range(seq_len + 2)]
        return torch.tensor(vals, dtype=torch.float64).to(torch.float32).cuda()

    def forward(self, x: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
        seq_len, batch, symbols = check_decoder_args(x, lengths)
        _lib.require_gpu()
        if seq_len == 0 or batch == 0:
            return [[] for _ in range(batch)]
        lib = _lib.load()
        xd = _lib.f32c(x)
        lens_dev = _lib.lens_i32(lengths)
        w = self.beam_width
        out_idx = torch.empty((batch, seq_len), dtype=torch.int32, device="cuda")
        out_len = torch.empty(batch, dtype=torch.int32, device="cuda")
        ws = self._workspace.get(lib.ms_ctc_beam_workspace_bytes(seq_len, batch, symbols, w))
        sep = -1 if self.separator_index is None else int(self.separator_index)
        wf = self._word_factor(seq_len)
        use_lm = self.language_model is not None and self.separator_index is not None

        def call(t0, t1, lm_factor, finish, beam_len=None, beam_idx=None, beam_plen=None):
            _lib.check(lib.ms_ctc_beam_decode(_lib.ptr(xd), _lib.ptr(lens_dev), _lib.ptr(out_idx), _lib.ptr(out_len),
                                              seq_len, batch, symbols, self.blank_index, w,
                                              float(self.prune_threshold), sep, _lib.ptr(wf), t0, t1,
                                              _lib.ptr(lm_factor), finish, _lib.ptr(beam_len), _lib.ptr(beam_idx),
                                              _lib.ptr(beam_plen), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "ms_ctc_beam_decode")

        if not use_lm:
            call(0, seq_len, None, 1)
            return ragged_to_lists(out_idx, out_len)

        # host language model: one frame per launch, beam prefixes read back in between
        beam_len = torch.empty(batch, dtype=torch.int32, device="cuda")
        beam_idx = torch.empty((batch, w, seq_len), dtype=torch.int32, device="cuda")
        beam_plen = torch.empty((batch, w), dtype=torch.int32, device="cuda")
        max_len = int(lengths.max())
        call(0, 0, None, 0, beam_len, beam_idx, beam_plen)  # initialise: beam = [()]
        for t in range(max_len):
            bl, bi, bp = beam_len.cpu().tolist(), beam_idx.cpu(), beam_plen.cpu().tolist()
            fac = torch.ones((batch, w), dtype=torch.float32)
            for n in range(batch):
                for k in range(bl[n]):
                    prefix = tuple(bi[n, k, :bp[n][k]].tolist()) + (sep,)
                    fac[n, k] = float(self.language_model(prefix) ** self.lm_weight)
            call(t, t + 1, fac.cuda(), 1 if t == max_len - 1 else 0, beam_len, beam_idx, beam_plen)
        if max_len == 0:
            call(0, 0, None, 1)
        return ragged_to_lists(out_idx, out_len)

    def extra_repr(self) -> str:
        return ",\n".join([f"blank_index={self.blank_index}", f"beam_width={self.beam_width}",
                           f"prune_threshold={self.prune_threshold}", f"language_model={self.language_model}",
                           f"lm_weight={self.lm_weight}", f"separator_index={self.separator_index}",
                           f"word_weight={self.word_weight}"])
