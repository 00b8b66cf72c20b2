"""Fully connected stack: mirror of myrtlespeech/model/fully_connected.py.

``self.fully_connected`` is built exactly like the reference (``torch.nn.Linear``
[+ activation] [+ Dropout] per hidden layer, then the output ``Linear``), so the
state_dict keys (``fully_connected.{0,2,..}.weight``) and repr match; the modules
are parameter containers -- the forward pass runs each Linear (+ its clamp) as one
MFMA GEMM with a fused epilogue (``ms_linear_forward``).
"""
import os
import threading
from collections import OrderedDict
from typing import List, Optional, Tuple, Union

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.utils import activation_clamp


def linear_stack_plan(module: Union[torch.nn.Linear, torch.nn.Sequential], training: bool
                      ) -> List[Tuple[torch.nn.Linear, Optional[Tuple[float, float]]]]:
    """[(linear, fused clamp | None), ...] for a Linear/activation/Dropout chain."""
    mods = [module] if isinstance(module, torch.nn.Linear) else list(module)
    plan: List[Tuple[torch.nn.Linear, Optional[Tuple[float, float]]]] = []
    for m in mods:
        if isinstance(m, torch.nn.Linear):
            plan.append((m, None))
        elif isinstance(m, torch.nn.Dropout):
            if training and m.p > 0:
                raise RuntimeError("training-mode Dropout is outside the inference hot path; call .eval()")
        else:
            clamp = activation_clamp(m)
            if clamp is not None:
                if not plan or plan[-1][1] is not None:
                    raise NotImplementedError("activation without a preceding Linear")
                plan[-1] = (plan[-1][0], clamp)
    return plan


_SPLIT_MIN_FLOPS = 2e9  # below this the two plane-split passes cost more than they save
_OUTPUT_LAYER_MAX_COLUMNS = 64   # csrc/gemm.hip linear_splitk_slices: layers this narrow take K slices of the exact-f32 kernel
_SPLITK = os.environ.get("MS_LINEAR_SPLITK") != "0"   # K slices for exact-f32 layers with few output tiles (A/B runs: 0)
# Operand-plane scratch of the split GEMM, one per HIP stream: work on one stream is ordered, so successive layers may share
# a buffer, but two streams (pipeline.BatchesInFlight) must not -- one stream's planes would be overwritten under the other's GEMM.
_split_ws = OrderedDict()
_SPLIT_WS_STREAMS = 4    # scratch buffers kept: the most recently used streams (a pipeline uses two)


# set while a HIP graph is being sized / captured (graph_scratch): that graph's own scratch.  Per THREAD: a capture on one host
# thread must not redirect the scratch of linear layers another thread issues on another stream (a cross-stream
# race on the operand planes)
_graph_tls = threading.local()


class graph_scratch:
    """``with graph_scratch(ws):`` -- the linear layers' operand-plane scratch comes from ``ws`` (a Workspace the capturing
    graph owns and keeps alive) instead of the per-stream LRU: a graph records POINTERS, and an LRU entry may be evicted and
    freed while a graph that recorded it is still replayed."""

    def __init__(self, ws: "_lib.Workspace"):
        self.ws = ws

    def __enter__(self):
        self.prev = getattr(_graph_tls, "ws", None)
        if os.environ.get("MS_GRAPH_SCRATCH") != "0":      # (0: A/B runs -- the graphs record the per-stream LRU scratch as before)
            _graph_tls.ws = self.ws
        return self.ws

    def __exit__(self, *exc):
        _graph_tls.ws = self.prev
        return False


def _stream_workspace() -> "_lib.Workspace":
    """Least-recently-used cache keyed on the stream handle: a stream that has been destroyed (its handle may be re-issued
    later) or is no longer used loses its buffer once ``_SPLIT_WS_STREAMS`` other streams have come by.  Dropping an entry is
    safe while its stream still runs: the caching allocator hands a block allocated on stream S only to later allocations
    on S, i.e. behind the kernels that read it."""
    graph_ws = getattr(_graph_tls, "ws", None)
    if graph_ws is not None:
        return graph_ws
    key = torch.cuda.current_stream().cuda_stream
    ws = _split_ws.get(key)
    if ws is None:
        ws = _split_ws[key] = _lib.Workspace()
        while len(_split_ws) > _SPLIT_WS_STREAMS:
            _split_ws.popitem(last=False)
    else:
        _split_ws.move_to_end(key)
    return ws


def _packed_planes(lin: torch.nn.Linear, w: torch.Tensor, k: int, n: int) -> torch.Tensor:
    """[hi | lo] operand planes of ``lin.weight`` (``ms_linear_split_pack``), cached on the module and remade when the
    parameter is replaced or edited in place."""
    key = (lin.weight.data_ptr(), _lib.version_of(lin.weight), w.data_ptr())
    rec = getattr(lin, "_ms_planes", None)
    if rec is None or rec[0] != key:
        lib = _lib.load()
        buf = torch.empty(lib.ms_linear_split_packed_bytes(k, n), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ms_linear_split_pack(_lib.ptr(w), _lib.ptr(buf), k, n, _lib.stream_ptr()), "ms_linear_split_pack")
        rec = (key, buf)
        lin._ms_planes = rec
    return rec[1]


def run_linear_stack(x2d: torch.Tensor, plan, few_rows: bool = False) -> torch.Tensor:
    """x2d [M, K] float32 cuda contiguous.  Large layers with K % 32 == 0 run as the
    split-bf16 GEMM (unless MS_PRECISION=f32), the rest as the exact-f32 GEMM.

    ``few_rows`` (MS_LINEAR_FEW_ROWS): the caller states that this stack serves a handful of rows (a single clip) and lets wide
    exact-f32 layers take K slices as well.  It is a property of the call site, never derived from ``M`` here: a row's bits must
    not depend on what it is batched or chunked with."""
    flags = _lib.LINEAR_FEW_ROWS if few_rows else 0
    lib = _lib.load()
    h = x2d
    for lin, clamp in plan:
        m, k = h.shape
        n = lin.out_features
        if k != lin.in_features:
            raise RuntimeError(f"size mismatch: input has {k} features, Linear expects {lin.in_features}")
        y = torch.empty((m, n), dtype=torch.float32, device="cuda")
        a, lo, hi = (_lib.ACT_NONE, 0.0, 0.0) if clamp is None else (_lib.ACT_CLAMP, clamp[0], clamp[1])
        w = _lib.f32c(lin.weight.detach())
        b = None if lin.bias is None else _lib.f32c(lin.bias.detach())
        # (an output layer -- <= 64 columns -- is exact float32 at EVERY row count: with the flop threshold alone the 29-column
        # layer of a batch of >= 68 ten-second utterances crossed it and an utterance's logits depended on its batch: round 6)
        if _lib.split_precision() and k % 32 == 0 and n > _OUTPUT_LAYER_MAX_COLUMNS and 2.0 * m * k * n >= _SPLIT_MIN_FLOPS:
            # the weight planes are made once per (weight, version) and kept with the layer; the scratch holds the x planes
            pw = _packed_planes(lin, w, k, n)
            ws = _stream_workspace().get((m * k * 4 + 255) // 256 * 256, zero=False)
            _lib.check(lib.ms_linear_split_forward_packed(_lib.ptr(h), _lib.ptr(pw), _lib.ptr(b), _lib.ptr(y), m, k, n, a, lo, hi,
                                                          _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "ms_linear_split_forward")
        else:
            # an output layer (<= 64 columns): K slices, added in slice order (ms_linear_splitk_forward); 0 bytes = not such a
            # layer, and the call is ms_linear_forward
            nb = lib.ms_linear_splitk_workspace_bytes(m, k, n, flags) if _SPLITK else 0
            ws = _stream_workspace().get(nb, zero=False) if nb else None
            _lib.check(lib.ms_linear_splitk_forward(_lib.ptr(h), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), m, k, n, a, lo, hi,
                                                    flags, _lib.ptr(ws), nb, _lib.stream_ptr()), "ms_linear_splitk_forward")
        h = y
    return h


class FullyConnected(torch.nn.Module):
    """A fully connected neural network (fully_connected.py:9-166)."""

    def __init__(self, in_features: int, out_features: int, num_hidden_layers: int, hidden_size: Optional[int],
                 hidden_activation_fn: Optional[torch.nn.Module], dropout: Optional[float] = None):
        if num_hidden_layers < 0:
            raise ValueError("num_hidden_layers must be >= 0")
        if dropout and (dropout < 0 or dropout > 1):
            raise ValueError(f"dropout must be >= 0. and <= 1. but dropout={dropout}")
        if num_hidden_layers == 0:
            if hidden_size is not None:
                raise ValueError("num_hidden_layers==0 but hidden_size is not None")
            if hidden_activation_fn is not None:
                raise ValueError("num_hidden_layers==0 but hidden_activation_fn is not None")
            if dropout is not None:
                raise ValueError("num_hidden_layers==0 so dropout must be None.")
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.dropout = dropout
        self.fully_connected = self._build_fully_connected(in_features, out_features, num_hidden_layers, hidden_size,
                                                           hidden_activation_fn, dropout)
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.fully_connected = self.fully_connected.cuda()

    def _build_fully_connected(self, in_features, out_features, num_hidden_layers, hidden_size, hidden_activation_fn,
                               dropout) -> Union[torch.nn.Linear, torch.nn.Sequential]:
        hidden = []
        width = in_features
        for _ in range(num_hidden_layers):
            hidden.append(torch.nn.Linear(width, hidden_size))
            if hidden_activation_fn:
                hidden.append(hidden_activation_fn)
            if dropout:
                hidden.append(torch.nn.Dropout(p=dropout))
            width = hidden_size
        last = torch.nn.Linear(width, out_features)
        return torch.nn.Sequential(*hidden, last) if hidden else last

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """``[batch, max_seq_len, in_features] -> [batch, max_seq_len, out_features]``;
        lengths pass through (moved to the device like fully_connected.py:160-162)."""
        _lib.require_gpu()
        x_inp, x_len = x
        h = _lib.f32c(x_inp)
        lead = h.shape[:-1]
        y = run_linear_stack(h.reshape(-1, h.shape[-1]), linear_stack_plan(self.fully_connected, self.training))
        return y.reshape(*lead, y.shape[-1]), _lib.lens_to_device(x_len)
