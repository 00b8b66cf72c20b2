"""Lookahead convolution: mirror of myrtlespeech/model/lookahead.py."""
import math
from typing import Optional, Tuple

import torch

from myrtlespeech_amd import _lib


def lookahead_apply(x: torch.Tensor, weight: torch.Tensor, x_strides, n: int, f: int, t: int,
                    out_layout: str = "nft", clamp: Optional[Tuple[float, float]] = None,
                    t_out: Optional[int] = None) -> torch.Tensor:
    """Runs ``ms_lookahead_forward`` on a float32 cuda tensor addressed by element
    strides ``x_strides = (s_n, s_f, s_t)``.  ``out_layout`` "nft" -> [N,F,T]
    contiguous, "ntf" -> [N,T,F] contiguous (what the fully connected stack reads).
    ``t_out``: only the first ``t_out`` of the ``t`` frames are produced (a window of a stream, ``streaming.py``)."""
    lib = _lib.load()
    ctx = weight.shape[-1]
    w = _lib.f32c(weight.detach()).reshape(f, ctx)
    t_in, t = t, (t if t_out is None else t_out)
    if out_layout == "nft":
        y = torch.empty((n, f, t), dtype=torch.float32, device="cuda")
        ys = (f * t, t, 1)
    else:
        y = torch.empty((n, t, f), dtype=torch.float32, device="cuda")
        ys = (t * f, 1, f)
    a, lo, hi = (_lib.ACT_NONE, 0.0, 0.0) if clamp is None else (_lib.ACT_CLAMP, clamp[0], clamp[1])
    # (x may be a view: this entry point takes its element strides)
    _lib.check(lib.ms_lookahead_window_forward(_lib.ptr(x, strided=True), _lib.ptr(w), _lib.ptr(y), n, f, t_in, t, ctx, x_strides[0],
                                               x_strides[1], x_strides[2], ys[0], ys[1], ys[2], a, lo, hi, _lib.stream_ptr()),
               "ms_lookahead_forward")
    return y


class Lookahead(torch.nn.Module):
    r"""A lookahead convolution (lookahead.py:8-74): ``[batch, in_features, seq_len]``
    in and out, each output frame a per-feature linear combination of the next
    ``context`` input frames; weight ``[in_features, 1, context]``."""

    def __init__(self, in_features: int, context: int):
        super().__init__()
        self.in_features = in_features
        self.context = context
        self.weight = torch.nn.Parameter(torch.Tensor(self.in_features, 1, self.context))
        self.reset_parameters()
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.cuda()  # the reference's `self.weight.cuda()` is a no-op; move the parameter for real

    def reset_parameters(self) -> None:
        torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        _lib.require_gpu()
        acts, lens = x
        if acts.dtype != torch.float32 or not acts.is_cuda:
            acts = _lib.f32c(acts)
        n, f, t = acts.shape
        if f != self.in_features:
            raise RuntimeError(f"expected {self.in_features} features, got {f}")
        y = lookahead_apply(acts, self.weight, acts.stride(), n, f, t)
        return y, _lib.lens_to_device(lens)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, context={self.context}"
