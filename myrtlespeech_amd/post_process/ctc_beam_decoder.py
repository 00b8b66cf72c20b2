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

        # host language model.  The reference consults it for beam entry l at frame t only when the separator extension of l
        # survives the pruning test -- float32 p[t, n, sep] > prune_threshold, ctc_beam_decoder.py:198 -- and multiplies the
        # separator extension by float32(lm(l + (sep,)) ** lm_weight) (:214-230).  So the host reads the separator's column of
        # the posteriors ONCE, advances the kernel over every run of frames in which no utterance's separator survives in one
        # launch (no factor is read there), and stops only at the frames that need the model: there it reads back the live
        # beam -- prefix lengths first, then only the first max(prefix length) columns of the prefix table instead of all T
        # (round 5 copied [batch, width, T] int32 every frame: 513 KB at T = 501, batch 32, width 8) -- and calls the model for
        # the entries of the utterances whose separator survives, as the reference does.
        beam_len = torch.empty(batch, dtype=torch.int32, device="cuda")
        beam_idx = torch.empty((batch, w, seq_len), dtype=torch.int32, device="cuda")
        beam_plen = torch.empty((batch, w), dtype=torch.int32, device="cuda")
        lens_h = _lib.host_lens(lengths)
        max_len = int(lens_h.max())
        thr = torch.tensor(float(self.prune_threshold), dtype=torch.float32)
        # [T, N] bool: the separator extension of utterance n is visited at frame t (float32 compare, as in the reference)
        # (the reference SKIPS when p <= thr: a NaN probability is visited, so the test is "not (p <= thr)")
        need = ~(xd[:max_len, :, sep].cpu() <= thr) & (torch.arange(max_len)[:, None] < lens_h[None, :].to(torch.int64))
        need_t = need.any(dim=1).tolist()
        self.lm_calls = self.lm_frames = 0
        call(0, 0, None, 0, beam_len, beam_idx, beam_plen)  # initialise: beam = [()]
        t = 0
        while t < max_len:
            if not need_t[t]:
                t1 = t + 1
                while t1 < max_len and not need_t[t1]:
                    t1 += 1
                call(t, t1, None, 1 if t1 == max_len else 0, beam_len, beam_idx, beam_plen)
                t = t1
                continue
            bl, bp = beam_len.cpu().tolist(), beam_plen.cpu()
            longest = int(bp.max()) if bp.numel() else 0
            bi = beam_idx[:, :, :max(longest, 1)].cpu() if longest else None
            bp = bp.tolist()
            fac = torch.ones((batch, w), dtype=torch.float32)
            row = need[t].tolist()
            for n in range(batch):
                if not row[n]:
                    continue
                for k in range(bl[n]):
                    pre = tuple(bi[n, k, :bp[n][k]].tolist()) if bp[n][k] else ()
                    if pre and pre[-1] == sep:
                        continue        # a repeated separator takes the repeat-character branch (:210-213): no model there
                    prefix = pre + (sep,)
                    fac[n, k] = float(self.language_model(prefix) ** self.lm_weight)
                    self.lm_calls += 1
            self.lm_frames += 1
            call(t, t + 1, fac.cuda(), 1 if t == max_len - 1 else 0, beam_len, beam_idx, beam_plen)
            t += 1
        if max_len == 0:
            call(0, 0, None, 1)
        return ragged_to_lists(out_idx, out_len)

    def extra_repr(self) -> str:
        return ",\n".join([f"blank_index={self.blank_index}", f"beam_width={self.beam_width}",
                           f"prune_threshold={self.prune_threshold}", f"language_model={self.language_model}",
                           f"lm_weight={self.lm_weight}", f"separator_index={self.separator_index}",
                           f"word_weight={self.word_weight}"])
