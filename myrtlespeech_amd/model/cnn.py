"""Masked convolutions over (feature, time): mirror of myrtlespeech/model/cnn.py.

Same classes, constructor arguments, ``(acts, seq_lens)`` convention and
``weight`` / ``bias`` state_dict keys as the reference; the forward pass is one
HIP implicit-GEMM kernel (``csrc/conv.hip``) instead of mask + F.pad + cuDNN.
"""
from enum import Enum
import os
from typing import List, Optional, Tuple, TypeVar, Union

import torch

from myrtlespeech_amd import _lib


class PaddingMode(Enum):
    """cnn.py:10-14."""

    NONE = 0
    SAME = 1


def pad_same(length: int, kernel_size: int, stride: int = 1, dilation: int = 1) -> Tuple[int, int]:
    """(left, right) SAME padding exactly as the reference *code* computes it
    (cnn.py:148-163; its docstring formula differs by stride-1)."""
    for name, v in (("length", length), ("kernel_size", kernel_size), ("stride", stride), ("dilation", dilation)):
        if v <= 0:
            raise ValueError(f"{name}={v} must be > 0")
    span = dilation * (kernel_size - 1) + 1
    n_out = (length + stride - 1) // stride
    total = stride * n_out - 1 + span - length
    left = total // 2
    return left, total - left


def out_lens(seq_lens: torch.Tensor, kernel_size: int, stride: int, dilation: int, padding: int) -> torch.Tensor:
    """Sequence lengths after a convolution (cnn.py:191-197): float32 arithmetic,
    floor, cast back to the input dtype.  (The reference mutates an already
    float ``seq_lens`` in place; that side effect is deliberately not kept.)"""
    v = seq_lens.to(torch.float32)
    v = (v + float(padding) - float(dilation * (kernel_size - 1) + 1)) / float(stride) + 1.0
    return v.floor().to(seq_lens.dtype)


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (list, tuple)) else (int(v), int(v))


class _PackedFilters:
    """Device-side re-layout of a conv weight, rebuilt when the parameter changes."""

    def __init__(self):
        self.key = None
        self.buf = None
        self.key_cl = None
        self.buf_cl = None
        self.workspace = _lib.Workspace()

    def get_cl(self, weight: torch.Tensor) -> torch.Tensor:
        """Split-bf16 channels-last bank for the multi-channel kernel (conv_cl.hip)."""
        key = (weight.data_ptr(), _lib.version_of(weight), tuple(weight.shape))
        if key != self.key_cl:
            lib = _lib.load()
            cout, cin, kf, kt = weight.shape
            self.buf_cl = torch.empty(lib.ms_maskconv_cl_packed_bytes(cout, cin, kf, kt), dtype=torch.uint8, device="cuda")
            w = _lib.f32c(weight.detach())
            _lib.check(lib.ms_maskconv_cl_pack(_lib.ptr(w), _lib.ptr(self.buf_cl), cout, cin, kf, kt, _lib.stream_ptr()),
                       "ms_maskconv_cl_pack")
            self.key_cl = key
        return self.buf_cl

    def get_fwin(self, weight: torch.Tensor) -> torch.Tensor:
        """Split-bf16 bank of a single-channel filter for the feature-window kernel (conv_cl.hip, ms_maskconv_fwin_*)."""
        key = (weight.data_ptr(), _lib.version_of(weight), tuple(weight.shape), "fwin")
        if key != self.key_cl:
            lib = _lib.load()
            cout, _, kf, kt = weight.shape
            self.buf_cl = torch.empty(lib.ms_maskconv_fwin_packed_bytes(cout, kf, kt), dtype=torch.uint8, device="cuda")
            w = _lib.f32c(weight.detach())
            _lib.check(lib.ms_maskconv_fwin_pack(_lib.ptr(w), _lib.ptr(self.buf_cl), cout, kf, kt, _lib.stream_ptr()),
                       "ms_maskconv_fwin_pack")
            self.key_cl = key
        return self.buf_cl

    def get_gemm1d(self, weight: torch.Tensor) -> torch.Tensor:
        """bf16 hi / lo planes [Cout, Cin*KT padded to 32] for the im2col + split-GEMM conv1d path (conv1d_gemm.hip)."""
        key = (weight.data_ptr(), _lib.version_of(weight), tuple(weight.shape), "gemm1d")
        if key != self.key_cl:
            lib = _lib.load()
            cout, cin, _, kt = weight.shape
            self.buf_cl = torch.empty(lib.ms_maskconv1d_gemm_packed_bytes(cout, cin, kt), dtype=torch.uint8, device="cuda")
            w = _lib.f32c(weight.detach())
            _lib.check(lib.ms_maskconv1d_gemm_pack(_lib.ptr(w), _lib.ptr(self.buf_cl), cout, cin, kt, _lib.stream_ptr()),
                       "ms_maskconv1d_gemm_pack")
            self.key_cl = key
        return self.buf_cl

    def get(self, weight: torch.Tensor, groups: int) -> torch.Tensor:
        key = (weight.data_ptr(), _lib.version_of(weight), tuple(weight.shape), groups)
        if key != self.key:
            lib = _lib.load()
            cout, cin_g, kf, kt = weight.shape
            nbytes = lib.ms_maskconv_packed_bytes(cout, cin_g, kf, kt, groups)
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            w = _lib.f32c(weight.detach())
            _lib.check(lib.ms_maskconv_pack(_lib.ptr(w), _lib.ptr(self.buf), cout, cin_g, kf, kt, groups,
                                            _lib.stream_ptr()), "ms_maskconv_pack")
            self.key = key
        return self.buf


def _conv_forward(x4: torch.Tensor, seq_lens: torch.Tensor, weight4: torch.Tensor, bias: Optional[torch.Tensor],
                  packed: _PackedFilters, stride, dilation, groups: int, same: bool,
                  act: Optional[Tuple[float, float]], time_pads: Optional[Tuple[int, int]] = None):
    """x4 [N,C,F,T] (float32, cuda, contiguous) -> y [N,Cout,Fout,Tout], new lens.  ``time_pads`` overrides the (left,
    right) zero padding along time (a window of a stream carries its context as real frames, ``streaming.py``)."""
    lib = _lib.load()
    n, cin, fin, tin = x4.shape
    cout, _, kf, kt = weight4.shape
    (sf, st), (df, dt) = stride, dilation
    if same:
        pt = pad_same(tin, kt, st, dt)
        pf = pad_same(fin, kf, sf, df)
    else:
        pt = pf = (0, 0)
    if time_pads is not None:
        pt = (int(time_pads[0]), int(time_pads[1]))
    fout = (fin + sum(pf) - (df * (kf - 1) + 1)) // sf + 1
    tout = (tin + sum(pt) - (dt * (kt - 1) + 1)) // st + 1
    if fout <= 0 or tout <= 0:
        raise RuntimeError("convolution output would be empty")
    # lengths arithmetic on the host values (same float32 formula, same dtype as the input), uploaded without blocking
    host_in = _lib.host_lens(seq_lens).to(seq_lens.dtype)
    new_host = out_lens(host_in, kt, st, dt, sum(pt))
    new_lens = _lib.attach_host(_lib.upload(new_host), new_host)
    y = torch.empty((n, cout, fout, tout), dtype=torch.float32, device="cuda")
    lens_dev = _lib.lens_i32(seq_lens)
    a, lo, hi = (_lib.ACT_NONE, 0.0, 0.0) if act is None else (_lib.ACT_CLAMP, act[0], act[1])
    b = None if bias is None else _lib.f32c(bias.detach())
    # below ~1 GFLOP the exact-f32 tap kernel wins (no layout pass); MS_CONV_MFMA_MIN_FLOPS moves the threshold (tests)
    big = 2.0 * y.numel() * cin * kf * kt >= float(os.environ.get("MS_CONV_MFMA_MIN_FLOPS", "1e9"))
    # (a 2-D convolution over ONE feature row only is that row's 1-D convolution if the row is not padded away: with a
    # feature stride > 1 the reference's SAME padding puts a zero row in front of it, cnn.py:148-163 -- found by tests/soak.py)
    if _lib.split_precision() and groups == 1 and fin == 1 and kf == 1 and sf == 1 and pf == (0, 0) and cin * kt >= 64 and big:
        # conv1d with many input channels: im2col (mask and padding as load predicates) + split-bf16 GEMM (conv1d_gemm.hip)
        pk = packed.get_gemm1d(weight4)
        nbytes = lib.ms_maskconv1d_gemm_workspace_bytes(n, cin, tout, cout, kt)
        ws = packed.workspace.get(nbytes)
        _lib.check(lib.ms_maskconv1d_gemm_forward(_lib.ptr(x4), _lib.ptr(lens_dev), _lib.ptr(pk), _lib.ptr(b), _lib.ptr(y), n,
                                                  cin, tin, cout, tout, kt, st, dt, pt[0], a, lo, hi, _lib.ptr(ws), nbytes,
                                                  _lib.stream_ptr()), "ms_maskconv1d_gemm_forward")
        return y, new_lens
    if _lib.split_precision() and groups == 1 and cin % 16 == 0 and big:
        # many input channels: split-bf16 implicit GEMM over channels (conv_cl.hip)
        pk = packed.get_cl(weight4)
        ws = packed.workspace.get(lib.ms_maskconv_cl_workspace_bytes(n, cin, fin, tin))
        rc = lib.ms_maskconv_cl_forward(_lib.ptr(x4), _lib.ptr(lens_dev), _lib.ptr(pk), _lib.ptr(b), _lib.ptr(y), n, cin,
                                        fin, tin, cout, fout, tout, kf, kt, sf, st, df, dt, pf[0], pt[0], a, lo, hi,
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if rc != 5:  # MS_ERR_UNSUPPORTED -> the exact-f32 kernel below
            _lib.check(rc, "ms_maskconv_cl_forward")
            return y, new_lens
    # (the feature-window form pays from ~0.2 GFLOP: a streaming window of DS2's conv1 at 32 streams is 0.59 GFLOP, 69 us on
    # the exact-f32 kernel and 28 us here, layout pass included)
    big_fwin = 2.0 * y.numel() * cin * kf * kt >= float(os.environ.get("MS_CONV_MFMA_MIN_FLOPS", "2e8"))
    if (_lib.split_precision() and groups == 1 and cin == 1 and df == 1 and sf % 2 == 0 and kf >= 16 and big_fwin
            and os.environ.get("MS_CONV_FWIN") != "0"):
        # one input channel, tall filter (DS2 conv1): the feature window takes the place of the channels (conv_cl.hip)
        pk = packed.get_fwin(weight4)
        nbytes = lib.ms_maskconv_fwin_workspace_bytes(n, tin, kf, sf, fout)
        ws = packed.workspace.get(nbytes)
        rc = lib.ms_maskconv_fwin_forward(_lib.ptr(x4), _lib.ptr(lens_dev), _lib.ptr(pk), _lib.ptr(b), _lib.ptr(y), n, fin,
                                          tin, cout, fout, tout, kf, kt, sf, st, dt, pf[0], pt[0], a, lo, hi, _lib.ptr(ws),
                                          nbytes, _lib.stream_ptr())
        if rc != 5:  # MS_ERR_UNSUPPORTED -> the exact-f32 kernel below
            _lib.check(rc, "ms_maskconv_fwin_forward")
            return y, new_lens
    pk = packed.get(weight4, groups)
    _lib.check(lib.ms_maskconv_forward(_lib.ptr(x4), _lib.ptr(lens_dev), _lib.ptr(pk), _lib.ptr(b), _lib.ptr(y), n, cin,
                                       fin, tin, cout, fout, tout, kf, kt, sf, st, df, dt, pf[0], pt[0], groups, a, lo,
                                       hi, _lib.stream_ptr()), "ms_maskconv_forward")
    return y, new_lens


def _mask_in_place(acts: torch.Tensor, seq_lens: torch.Tensor):
    """MaskConv*._mask_ (cnn.py:280-293, 425-443) on the caller's tensor."""
    if acts.is_cuda and acts.dtype == torch.float32 and acts.is_contiguous():
        n, t = acts.shape[0], acts.shape[-1]
        h = _lib.host_lens(seq_lens)
        if h.numel() == n and n and int(h.min()) >= t:
            return True          # every sequence fills the tensor: nothing to zero (a steady-state streaming chunk)
        inner = acts.numel() // (n * t)
        lens_dev = _lib.lens_i32(seq_lens)  # keep alive until the launch is enqueued
        _lib.check(_lib.load().ms_mask_time_(_lib.ptr(acts), _lib.ptr(lens_dev), n, inner, t, _lib.stream_ptr()),
                   "ms_mask_time_")
        return True
    return False


class MaskConv1d(torch.nn.Conv1d):
    """1D convolution over ``[batch, channels, seq_len]`` with per-sequence
    lengths (cnn.py:200-336).  ``torch.nn.Conv1d`` is the parameter container
    (identical init, ``weight``/``bias`` keys and repr); its forward is never used."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 padding_mode: PaddingMode = PaddingMode.NONE, dilation: int = 1, groups: int = 1, bias: bool = True):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                         dilation=dilation, groups=groups, bias=bias)
        if padding_mode not in (PaddingMode.NONE, PaddingMode.SAME):
            raise ValueError(f"unknown padding mode {padding_mode}")
        self._padding_mode = padding_mode
        self._packed = _PackedFilters()
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            super().cuda()

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor],
                fused_activation: Optional[Tuple[float, float]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        _lib.require_gpu()
        acts, seq_lens = x
        _mask_in_place(acts, seq_lens)  # visible to the caller only for device tensors, as in the reference
        x4 = _lib.f32c(acts).unsqueeze(2)
        y, new_lens = _conv_forward(x4, seq_lens, self.weight.unsqueeze(2), self.bias, self._packed,
                                    (1, self.stride[0]), (1, self.dilation[0]), self.groups,
                                    self._padding_mode == PaddingMode.SAME, fused_activation)
        return y.squeeze(2), new_lens

    def extra_repr(self) -> str:
        return super().extra_repr() + f", padding_mode={self._padding_mode}"


class MaskConv2d(torch.nn.Conv2d):
    """2D convolution over ``[batch, channels, features, seq_len]`` with
    per-sequence lengths (cnn.py:339-486); ``kernel_size``/``stride`` are
    ``[feature, time]``."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[int, List[int]],
                 stride: Union[int, List[int]] = 1, padding_mode: PaddingMode = PaddingMode.NONE, dilation: int = 1,
                 groups: int = 1, bias: bool = True):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                         dilation=dilation, groups=groups, bias=bias)
        if padding_mode not in (PaddingMode.NONE, PaddingMode.SAME):
            raise ValueError(f"unknown padding mode {padding_mode}")
        self._padding_mode = padding_mode
        self._packed = _PackedFilters()
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            super().cuda()

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor],
                fused_activation: Optional[Tuple[float, float]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        _lib.require_gpu()
        acts, seq_lens = x
        # the reference zeroes the caller's tensor in place (cnn.py:442); keep that when
        # the caller's storage is what the kernel reads, otherwise the kernel's own
        # t < len predicate does the masking on the device copy
        _mask_in_place(acts, seq_lens)
        x4 = _lib.f32c(acts)
        y, new_lens = _conv_forward(x4, seq_lens, self.weight, self.bias, self._packed, _pair(self.stride),
                                    _pair(self.dilation), self.groups, self._padding_mode == PaddingMode.SAME,
                                    fused_activation)
        return y, new_lens

    def extra_repr(self) -> str:
        return super().extra_repr() + f", padding_mode={self._padding_mode}"


SeqLenT = TypeVar("SeqLenT", torch.Tensor, Tuple[torch.Tensor, torch.Tensor])


class Conv2dTo1d(torch.nn.Module):
    """``[N, C, H, W] -> [N, C*H, W]`` (cnn.py:492-537)."""

    def __init__(self, seq_len_support: bool = True):
        super().__init__()
        self.seq_len_support = seq_len_support

    def forward(self, x: SeqLenT) -> SeqLenT:
        acts, seq_lens = x if self.seq_len_support else (x, None)
        n, c, f, t = acts.size()
        acts = acts.view(n, c * f, t)
        return (acts, seq_lens) if self.seq_len_support else acts

    def extra_repr(self) -> str:
        return f"seq_len_support={self.seq_len_support}"


class Conv1dTo2d(torch.nn.Module):
    """``[N, C, W] -> [N, 1, C, W]`` (cnn.py:540-583)."""

    def __init__(self, seq_len_support: bool = True):
        super().__init__()
        self.seq_len_support = seq_len_support

    def forward(self, x: SeqLenT) -> SeqLenT:
        acts, seq_lens = x if self.seq_len_support else (x, None)
        acts = acts.unsqueeze(1)
        return (acts, seq_lens) if self.seq_len_support else acts

    def extra_repr(self) -> str:
        return f"seq_len_support={self.seq_len_support}"
