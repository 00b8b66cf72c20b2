"""Mirror of myrtlespeech/model/seq_len_wrapper.py:8-32."""
from typing import Any, Callable, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.utils import activation_clamp


class SeqLenWrapper(torch.nn.Module):
    """Adds sequence length support to a module: ``(module(x[0]), seq_lens_fn(x[1]))``.

    Hardtanh / ReLU / Identity payloads (what builders/activation.py builds) run as
    the HIP clamp kernel; any other payload module is called as is."""

    def __init__(self, module: torch.nn.Module, seq_lens_fn: Callable[[torch.Tensor], torch.Tensor]):
        super().__init__()
        self.module = module
        self.seq_lens_fn = seq_lens_fn

    def forward(self, x: Tuple[Any, torch.Tensor], *args, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        m = self.module
        if isinstance(m, (torch.nn.Hardtanh, torch.nn.ReLU, torch.nn.Identity)) and not args and not kwargs:
            clamp = activation_clamp(m)
            if clamp is None:
                result = x[0]
            else:
                _lib.require_gpu()
                src = _lib.f32c(x[0])
                result = src if getattr(m, "inplace", False) else torch.empty_like(src)
                _lib.check(_lib.load().ms_clamp(_lib.ptr(src), _lib.ptr(result), src.numel(), clamp[0], clamp[1],
                                                _lib.stream_ptr()), "ms_clamp")
        else:
            result = m(x[0], *args, **kwargs)
        return result, self.seq_lens_fn(x[1])
