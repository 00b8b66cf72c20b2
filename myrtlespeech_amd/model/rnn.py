"""Recurrent layers: mirror of myrtlespeech/model/rnn.py.

``RNN`` keeps the reference's constructor, ``(data, lengths)`` convention, hidden
state layout ``[num_layers * num_directions, batch, hidden]`` and state_dict keys
(``rnn.weight_ih_l{k}[_reverse]`` ...).  ``self.rnn`` is a ``torch.nn.LSTM`` /
``GRU`` / ``RNN`` used purely as the parameter container (identical
initialisation and key names); the forward pass is ``ms_rnn_layer_forward``
(``csrc/rnn.hip``): an MFMA input-projection GEMM plus a persistent recurrent
kernel, with ``pack_padded_sequence`` semantics folded into a per-frame predicate.
"""
import ctypes
import os
from enum import IntEnum
from typing import List, Optional, Tuple, TypeVar

import torch

from myrtlespeech_amd import _lib


class RNNType(IntEnum):
    """rnn.py:9-12."""

    LSTM = 0
    GRU = 1
    BASIC_RNN = 2


RNNState = TypeVar("RNNState", torch.Tensor, Tuple[torch.Tensor, torch.Tensor])
RNNData = TypeVar("RNNData", bound=torch.Tensor)
Lengths = TypeVar("Lengths", bound=torch.Tensor)

_OVERLAP = os.environ.get("MS_RNN_OVERLAP") != "0"              # A/B switch: 0 = the layer-by-layer schedule everywhere
_OVERLAP_SEGMENTS = int(os.environ.get("MS_RNN_OVERLAP_SEGMENTS", "8"))   # time segments per layer (tools/overlap_emulation.py)
# The half-batch pipeline (below) is OFF unless MS_RNN_HALVES=1: measured slower than the one-batch form it was meant to beat --
# 1.107 against 0.957 ms per streaming chunk of 64 streams x 16 steps, same box, inside the replayed HIP graph
# (profiles/r06_stream_half_batch_pipeline_ab.txt): ten cross-stream dependencies per chunk (7 .. 25 us of idle each on this
# chip, EXPERIMENTS.md round 5), the slicing / concatenation kernels and 512-row GEMMs cost more than the overlap returns.
_HALVES = os.environ.get("MS_RNN_HALVES") == "1"
_HALVES_MAX_ROWS = int(os.environ.get("MS_RNN_HALVES_MAX_ROWS", "4096"))   # steps x sequences up to which the half-batch pipeline is used
_HX_PREINIT = os.environ.get("MS_RNN_HX_PREINIT") != "0"      # A/B switch (tests): 0 = every layer call initialises its exchange

_CELL = {RNNType.LSTM: _lib.CELL_LSTM, RNNType.GRU: _lib.CELL_GRU, RNNType.BASIC_RNN: _lib.CELL_RNN_TANH}


class PackedLayer:
    """Kernel-layout copy of one layer's weights, rebuilt when a parameter changes."""

    def __init__(self):
        self.key = None
        self.buf = None
        self._gru_key = None
        self._gru_params = None

    def tanh_as_gru(self, params: List[Tuple[Optional[torch.Tensor], ...]]):
        """A tanh-RNN layer's parameters as those of the GRU that computes it (``tanh_rnn_as_gru``), rebuilt when a parameter
        changes (the packed copy is keyed on THESE tensors, so it survives from call to call)."""
        key = tuple((p.data_ptr(), _lib.version_of(p)) if p is not None else None for d in params for p in d)
        if key != self._gru_key:
            self._gru_params = tanh_rnn_as_gru(params)
            self._gru_key = key
        return self._gru_params

    def get(self, cell: int, in_size: int, hidden: int, params: List[Tuple[Optional[torch.Tensor], ...]],
            pad: Optional[Tuple[int, bool, int]] = None):
        """params: per direction (w_ih, w_hh, b_ih | None, b_hh | None).  ``pad`` = (padded hidden size, the layer's input is
        the padded output of the layer below, zero columns appended to the input): the weights are packed at the padded width (``pad_layer_params``);
        ``in_size`` / ``hidden`` are then the PADDED sizes."""
        key = tuple((p.data_ptr(), _lib.version_of(p)) if p is not None else None for d in params for p in d) + (pad,)
        if key != self.key:
            lib = _lib.load()
            ndir = len(params)
            if pad is not None:
                params = pad_layer_params(cell, params, pad[0], pad[1], pad[2])
            nbytes = lib.ms_rnn_packed_bytes(cell, in_size, hidden, ndir)
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            keep = [[None if p is None else _lib.f32c(p.detach()) for p in d] for d in params]
            arr = ctypes.c_void_p * ndir

            def col(i):
                if keep[0][i] is None:
                    return ctypes.cast(None, ctypes.POINTER(ctypes.c_void_p))
                return ctypes.cast(arr(*[k[i].data_ptr() for k in keep]), ctypes.POINTER(ctypes.c_void_p))

            _lib.check(lib.ms_rnn_pack(cell, in_size, hidden, ndir, col(0), col(1), col(2), col(3), _lib.ptr(self.buf),
                                       _lib.stream_ptr()), "ms_rnn_pack")
            self.key = key
        return self.buf


_GATE_SATURATED = 1.0e4     # sigmoid(+-1e4) is exactly 1 / 0 in float32 (exp2 underflows to 0 / overflows to inf; csrc/rnn.hip fast_sigmoid)
_TANH_AS_GRU = os.environ.get("MS_RNN_TANH_AS_GRU") != "0"


def tanh_rnn_as_gru(params):
    """A tanh-RNN layer (rnn.py:112-127, ``RNNType.BASIC_RNN``: h' = tanh(W_ih x + b_ih + W_hh h + b_hh)) written as a GRU:
    with the reset gate held at exactly 1 and the update gate at exactly 0 -- zero weight rows, input biases of +-1e4 --
    torch's GRU cell  n = tanh(W_in x + b_in + r (W_hn h + b_hn)),  h' = (1 - z) n + z h  IS that layer (1 * a = a,
    0 * h = 0 and n + 0 = n are exact; only the association of the sum inside the tanh differs).  The persistent GRU kernel
    then serves it (round 6: one launch per layer instead of one per step, 6.9 -> ~2.5 ms at [501, 32, 1024] bidirectional)
    at three times the gate rows -- registers and MFMA issue it has to spare.  Per direction (w_ih, w_hh, b_ih, b_hh)."""
    out = []
    for w_ih, w_hh, b_ih, b_hh in params:
        h, dev = w_hh.shape[0], w_hh.device
        f32 = lambda t_: t_.detach().float()      # noqa: E731
        wi = torch.cat([torch.zeros(2 * h, w_ih.shape[1], device=dev), f32(w_ih)]).contiguous()
        wh = torch.cat([torch.zeros(2 * h, h, device=dev), f32(w_hh)]).contiguous()
        bi = torch.cat([torch.full((h,), _GATE_SATURATED, device=dev), torch.full((h,), -_GATE_SATURATED, device=dev),
                        torch.zeros(h, device=dev) if b_ih is None else f32(b_ih)]).contiguous()
        bh = torch.cat([torch.zeros(2 * h, device=dev), torch.zeros(h, device=dev) if b_hh is None else f32(b_hh)]).contiguous()
        out.append((wi, wh, bi, bh))
    return out


def pad_layer_params(cell: int, params, hp: int, padded_input: bool, in_pad: int = 0):
    """One layer's (w_ih, w_hh, b_ih, b_hh) per direction with the hidden size padded from H to ``hp``
    (``ms_rnn_padded_hidden``): every gate block gets ``hp - H`` zero rows, ``w_hh`` as many zero columns per row, and -- for
    a layer fed by a padded layer (``padded_input``: its input is ``ndir`` blocks of ``hp``) -- ``w_ih`` zero columns at the
    padded positions of every direction block.  Zero weights and biases keep the padded units at exactly 0."""
    gates = 4 if cell in (_lib.CELL_LSTM, _lib.CELL_HARD_LSTM) else (3 if cell == _lib.CELL_GRU else 1)
    ndir = len(params)
    out = []
    for w_ih, w_hh, b_ih, b_hh in params:
        h = w_hh.shape[1]
        f32 = lambda t_: t_.detach().float().contiguous()      # noqa: E731 -- (on whatever device the parameter lives: the CPU tests use this)
        wi = f32(w_ih).view(gates, h, -1)
        if padded_input:
            wi = torch.nn.functional.pad(wi.view(gates, h, ndir, h), (0, hp - h)).reshape(gates, h, ndir * hp)
        wi = torch.nn.functional.pad(wi, (0, in_pad, 0, hp - h)).reshape(gates * hp, -1).contiguous()   # (+ in_pad zero columns)
        wh = torch.nn.functional.pad(f32(w_hh).view(gates, h, h), (0, hp - h, 0, hp - h)).reshape(gates * hp, hp).contiguous()
        bs = [None if b is None else torch.nn.functional.pad(f32(b).view(gates, h), (0, hp - h)).reshape(-1).contiguous()
              for b in (b_ih, b_hh)]
        out.append((wi, wh, bs[0], bs[1]))
    return out


def run_layers(cell: int, x: torch.Tensor, lens_dev: Optional[torch.Tensor], max_len: int,
               layer_params: List[List[Tuple[Optional[torch.Tensor], ...]]], packed: List[PackedLayer], hidden: int,
               h0: Optional[torch.Tensor], c0: Optional[torch.Tensor], workspace: _lib.Workspace, check: bool = True,
               ragged: bool = False, keep_padding: bool = False, state_inplace: bool = False):
    """x [T,N,In] float32 cuda contiguous -> (out [T,N,D*H], hn, cn|None).  ``ragged``: the lengths differ -- layers that can
    then work on the rows that exist only, like torch's packed sequences (``MS_RNN_PACKED_ROWS``; same outputs).
    ``keep_padding`` (tests): return a padded stack's outputs and states at the padded width.  ``state_inplace``: the final
    states are written over ``h0`` / ``c0`` (which are then returned): every kernel reads an element of the initial state
    and writes the same element of the final state from the same thread, so the buffers may coincide -- a replayed HIP graph
    then chains its recurrent state without copies (streaming.py)."""
    lib = _lib.load()
    t, n, _ = x.shape
    ndir = len(layer_params[0])
    nl = len(layer_params)
    if cell == _lib.CELL_RNN_TANH and _TANH_AS_GRU and _lib.split_precision():
        # a tanh-RNN stack on the persistent GRU kernel where that width has one (tanh_rnn_as_gru); else a launch per step
        hp = int(lib.ms_rnn_padded_hidden(_lib.CELL_GRU, hidden, ndir))
        if lib.ms_rnn_layer_chains_planes(_lib.CELL_GRU, hp, ndir):
            gru_params = [packed[layer].tanh_as_gru(layer_params[layer]) for layer in range(nl)]
            return run_layers(_lib.CELL_GRU, x, lens_dev, max_len, gru_params, packed, hidden, h0, None, workspace, check, ragged,
                              keep_padding, state_inplace)
    lstm_like = cell in (_lib.CELL_LSTM, _lib.CELL_HARD_LSTM)
    # a hidden size without a persistent kernel runs at the next width that has one, its extra units held at exactly 0 by zero
    # weights (ms_rnn_padded_hidden; rnn.py:112-120 accepts any hidden_size)
    true_hidden = hidden
    hidden = int(lib.ms_rnn_padded_hidden(cell, true_hidden, ndir))
    padded = hidden != true_hidden
    in_pad = 0
    if padded:
        grow = (0, hidden - true_hidden)
        h0 = None if h0 is None else torch.nn.functional.pad(h0, grow)
        c0 = None if c0 is None else torch.nn.functional.pad(c0, grow)
        # ... and its first layer's input to a multiple of 32 (zero columns on both sides of the product), which is what the
        # split-bf16 projection GEMM takes; only for stacks that are being padded anyway -- an unpadded layer keeps the GEMM
        # (and the bits) it has always had
        if _lib.split_precision() and x.shape[2] % 32:
            in_pad = -x.shape[2] % 32
            x = torch.nn.functional.pad(x, (0, in_pad))
    inplace = (state_inplace and not padded and h0 is not None and h0.is_contiguous() and tuple(h0.shape) == (nl * ndir, n, hidden)
               and (not lstm_like or (c0 is not None and c0.is_contiguous() and c0.shape == h0.shape)))
    hn = h0 if inplace else torch.empty((nl * ndir, n, hidden), dtype=torch.float32, device="cuda")
    cn = (c0 if inplace else torch.empty_like(hn)) if lstm_like else None
    inp = x
    # one workspace for the whole stack (the time-out word in it is sticky, so ONE status check -- a host sync -- after
    # the last layer covers all of them)
    in_sizes = [x.shape[2]] + [ndir * hidden] * (nl - 1)
    ws = workspace.get(max(lib.ms_rnn_workspace_bytes(cell, t, n, k, hidden, ndir) for k in in_sizes))
    # intermediate outputs of a two-stream LSTM stack travel as the next layer's GEMM operand planes inside the workspace
    chain = nl > 1 and bool(lib.ms_rnn_layer_chains_planes(cell, hidden, ndir)) and (ndir * hidden) % 32 == 0
    # (all layers or none: a chained stack hands its planes from layer to layer in one row layout)
    pack_rows = ragged and lens_dev is not None and all(lib.ms_rnn_layer_packs_rows(cell, max_len, n, k, hidden, ndir)
                                                        for k in in_sizes)
    # one exchange initialisation for the whole stack where the layer kind allows it (ms_rnn_hx_preinit): the layers then use
    # exchange regions 0 .. nl - 1 of the workspace instead of re-initialising region 0 before every layer
    # ---- the overlapped schedule (round 6): the whole stack in ONE call, layer l+1's K-half projections computed on a second
    # stream beside layer l's recurrence (ms_rnn_stack_forward: a bidirectional wide-workgroup LSTM stack of <= 32 sequences
    # uses half of the CUs per recurrence).  Same bits as the layer loop below.  Not while two batches are in flight (the other
    # batch already fills the idle CUs, and its issue points are per layer), not inside a stream capture, not for packed rows.
    state_ok = all(s_ is None or (s_.is_contiguous() and tuple(s_.shape) == (nl * ndir, n, hidden)) for s_ in (h0, c0))
    if (nl > 1 and chain and not padded and not pack_rows and _OVERLAP and _lib.issue_point is None and state_ok
            and not torch.cuda.is_current_stream_capturing()
            and lib.ms_rnn_stack_overlap_ok(cell, t, n, x.shape[2], hidden, ndir, nl)):
        pks = [packed[layer].get(cell, in_sizes[layer], hidden, layer_params[layer], None) for layer in range(nl)]
        arr = (ctypes.c_void_p * nl)(*[pk.data_ptr() for pk in pks])
        out = torch.empty((t, n, ndir * hidden), dtype=torch.float32, device="cuda")
        _lib.check(lib.ms_rnn_stack_forward(cell, ctypes.cast(arr, ctypes.c_void_p), _lib.ptr(x), _lib.ptr(lens_dev), max_len,
                                            _lib.ptr(h0), _lib.ptr(c0), _lib.ptr(out), _lib.ptr(hn), _lib.ptr(cn), t, n, x.shape[2],
                                            hidden, ndir, nl, _OVERLAP_SEGMENTS, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                   "ms_rnn_stack_forward")
        if check:
            _lib.check(lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()), "ms_rnn_stack_forward")
        return out, hn, cn
    # ---- two half-batches in a pipeline (round 6; an experiment that did not pay, see _HALVES): 33 .. 64 SHORT sequences (a streaming chunk: 64 streams x 16 steps) ran as
    # one batch whose recurrence holds every CU for a launch that is latency-bound anyway, its projections in between.  As
    # two halves on two streams -- half A's recurrence (128 CUs) beside half B's projection, then the roles swapped -- a layer
    # takes two phases of max(recurrence, projection) instead of recurrence + projection.  Every utterance goes through the
    # same arithmetic whichever group or batch it is in (section 5 of DESIGN.md), so the outputs are the same bits.
    if (_HALVES and 32 < n <= 64 and t * n <= _HALVES_MAX_ROWS and chain and not padded and not pack_rows and nl <= 8
            and _lib.issue_point is None and bool(lib.ms_rnn_layer_is_wide(cell, hidden, ndir, n))
            and all(s_ is None or tuple(s_.shape) == (nl * ndir, n, hidden) for s_ in (h0, c0))):
        return _run_layers_halves(lib, cell, x, lens_dev, max_len, layer_params, packed, hidden, h0, c0, hn, cn, workspace, check,
                                  in_sizes, lstm_like)
    preinit = 0
    if 1 < nl <= 8 and _HX_PREINIT:
        rc = lib.ms_rnn_hx_preinit(cell, t, n, max(in_sizes), hidden, ndir, max_len, nl, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if rc == 0:
            preinit = 4096          # MS_RNN_HX_PREINIT
        elif rc != 5:               # MS_ERR_UNSUPPORTED: this layer kind initialises per call
            _lib.check(rc, "ms_rnn_hx_preinit")
    for layer in range(nl):
        in_size = in_sizes[layer]
        pk = packed[layer].get(cell, in_size, hidden, layer_params[layer],
                               (hidden, layer > 0, in_pad if layer == 0 else 0) if padded else None)
        flags = (1 if (chain and layer > 0) else 0) | (2 if (chain and layer < nl - 1) else 0) | (4 if pack_rows else 0)
        flags |= preinit | ((layer << 8) if preinit else 0)
        out = None if flags & 2 else torch.empty((t, n, ndir * hidden), dtype=torch.float32, device="cuda")
        sl = slice(layer * ndir, (layer + 1) * ndir)
        h0l = None if h0 is None else h0[sl].contiguous()
        c0l = None if (c0 is None or not lstm_like) else c0[sl].contiguous()
        hnl = hn[sl]
        cnl = cn[sl] if lstm_like else None
        _lib.check(lib.ms_rnn_layer_forward_ex(cell, _lib.ptr(pk), _lib.ptr(None if flags & 1 else inp), _lib.ptr(lens_dev),
                                               max_len, _lib.ptr(h0l), _lib.ptr(c0l), _lib.ptr(out), _lib.ptr(hnl),
                                               _lib.ptr(cnl), t, n, in_size, hidden, ndir, flags, _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr()),
                   "ms_rnn_layer_forward")
        inp = out
        if layer + 1 < nl:
            _lib.at_issue_point()   # two batches in flight: the other batch's next layer is issued here
    if check:
        _lib.check(lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()), "ms_rnn_layer_forward")
    if padded and not keep_padding:      # drop the padded units (exact zeros) of every direction
        inp = inp.view(t, n, ndir, hidden)[..., :true_hidden].reshape(t, n, ndir * true_hidden)
        hn = hn[..., :true_hidden].contiguous()
        cn = None if cn is None else cn[..., :true_hidden].contiguous()
    return inp, hn, cn


def _run_layers_halves(lib, cell, x, lens_dev, max_len, layer_params, packed, hidden, h0, c0, hn, cn, workspace, check, in_sizes,
                       lstm_like):
    """``run_layers`` for 33 .. 64 short sequences as two half-batches ([0, 32) and [32, n)) interleaved on two streams: a layer
    call is issued in its two parts (MS_RNN_PROJECTION_ONLY, MS_RNN_RECURRENCE_ONLY); a half's recurrence waits for the other
    half's previous one (two persistent launches are never resident together), its next projection follows on its own stream
    and runs beside the other half's recurrence.  Works inside a stream capture (the second stream is forked from and joined
    to the current one)."""
    t, n, _ = x.shape
    ndir, nl = len(layer_params[0]), len(layer_params)
    cur = torch.cuda.current_stream()
    extra = getattr(workspace, "_halves", None)
    if extra is None:
        extra = workspace._halves = (_lib.Workspace(), _lib.Workspace(), torch.cuda.Stream())
    streams = (cur, extra[2])
    cuts = ((0, 32), (32, n))
    xs = [x[:, a:b].contiguous() for a, b in cuts]
    lens_h = [None if lens_dev is None else lens_dev[a:b].contiguous() for a, b in cuts]
    # a half's longest sequence (lengths are sorted in decreasing order: the second half's is not the batch's)
    outs = [torch.empty((t, b - a, ndir * hidden), dtype=torch.float32, device="cuda") for a, b in cuts]
    hns = [torch.empty((nl * ndir, b - a, hidden), dtype=torch.float32, device="cuda") for a, b in cuts]
    cns = [torch.empty_like(h_) if lstm_like else None for h_ in hns]
    h0s = [None if h0 is None else h0[:, a:b].contiguous() for a, b in cuts]
    c0s = [None if (c0 is None or not lstm_like) else c0[:, a:b].contiguous() for a, b in cuts]
    wss = [extra[i].get(max(lib.ms_rnn_workspace_bytes(cell, t, b - a, k, hidden, ndir) for k in in_sizes)) for i, (a, b) in enumerate(cuts)]
    pks = [packed[layer].get(cell, in_sizes[layer], hidden, layer_params[layer], None) for layer in range(nl)]
    PROJ, REC = 16384, 8192          # MS_RNN_PROJECTION_ONLY, MS_RNN_RECURRENCE_ONLY
    streams[1].wait_stream(cur)
    pre = [0, 0]
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            if 1 < nl and _HX_PREINIT:
                rc = lib.ms_rnn_hx_preinit(cell, t, cuts[i][1] - cuts[i][0], max(in_sizes), hidden, ndir, max_len, nl, _lib.ptr(wss[i]),
                                           wss[i].numel(), _lib.stream_ptr())
                if rc == 0:
                    pre[i] = 4096
                elif rc != 5:
                    _lib.check(rc, "ms_rnn_hx_preinit")

    def part(i, layer, which):
        nh = cuts[i][1] - cuts[i][0]
        flags = (1 if layer > 0 else 0) | (2 if layer < nl - 1 else 0) | pre[i] | ((layer << 8) if pre[i] else 0) | which
        sl = slice(layer * ndir, (layer + 1) * ndir)
        h0l = None if h0s[i] is None else h0s[i][sl].contiguous()
        c0l = None if c0s[i] is None else c0s[i][sl].contiguous()
        out = outs[i] if layer == nl - 1 else None
        _lib.check(lib.ms_rnn_layer_forward_ex(cell, _lib.ptr(pks[layer]), _lib.ptr(xs[i] if layer == 0 else None), _lib.ptr(lens_h[i]),
                                               max_len, _lib.ptr(h0l), _lib.ptr(c0l), _lib.ptr(out), _lib.ptr(hns[i][sl]),
                                               _lib.ptr(cns[i][sl] if lstm_like else None), t, nh, in_sizes[layer], hidden, ndir, flags,
                                               _lib.ptr(wss[i]), wss[i].numel(), _lib.stream_ptr()), "ms_rnn_layer_forward")
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            part(i, 0, PROJ)
    rec_done = [None, None]
    for layer in range(nl):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                if rec_done[1 - i] is not None:
                    streams[i].wait_event(rec_done[1 - i])
                part(i, layer, REC)
                rec_done[i] = torch.cuda.Event()
                rec_done[i].record(streams[i])
                if layer + 1 < nl:
                    part(i, layer + 1, PROJ)
    cur.wait_stream(streams[1])
    if check:
        for w_ in wss:
            _lib.check(lib.ms_rnn_status(_lib.ptr(w_), _lib.stream_ptr()), "ms_rnn_layer_forward")
    out = torch.cat(outs, dim=1)
    hn.copy_(torch.cat(hns, dim=1))
    if lstm_like:
        cn.copy_(torch.cat(cns, dim=1))
    return out, hn, cn


class RNN(torch.nn.Module):
    """A recurrent neural network (rnn.py:41-205); see the module docstring."""

    def __init__(self, rnn_type: RNNType, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 dropout: float = 0.0, bidirectional: bool = False, forget_gate_bias: Optional[float] = None,
                 batch_first: bool = False):
        super().__init__()
        if rnn_type == RNNType.LSTM:
            rnn_cls = torch.nn.LSTM
        elif rnn_type == RNNType.GRU:
            rnn_cls = torch.nn.GRU
        elif rnn_type == RNNType.BASIC_RNN:
            rnn_cls = torch.nn.RNN
        else:
            raise ValueError(f"unknown rnn_type {rnn_type}")
        self.batch_first = batch_first
        self.bidirectional = bidirectional
        self.rnn_type = rnn_type
        # parameter container only: never called
        self.rnn = rnn_cls(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, bias=bias,
                           batch_first=batch_first, dropout=dropout, bidirectional=bidirectional)
        if rnn_type == RNNType.LSTM and bias and forget_gate_bias is not None:
            # forward-direction biases only, like rnn.py:122-127
            for layer in range(num_layers):
                getattr(self.rnn, f"bias_ih_l{layer}").data[hidden_size:2 * hidden_size] = forget_gate_bias
                getattr(self.rnn, f"bias_hh_l{layer}").data[hidden_size:2 * hidden_size] = 0.0
        self._packed = [PackedLayer() for _ in range(num_layers)]
        self._workspace = _lib.Workspace()
        self.check_status = True
        self.inplace_state = False      # streaming graphs: write h_n / c_n over the hx tensors they were given (run_layers)
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.rnn = self.rnn.cuda()

    def _layer_params(self):
        r = self.rnn
        out = []
        for layer in range(r.num_layers):
            dirs = []
            for sfx in ([""] + (["_reverse"] if self.bidirectional else [])):
                g = lambda n: getattr(r, f"{n}_l{layer}{sfx}")  # noqa: E731
                dirs.append((g("weight_ih"), g("weight_hh"), g("bias_ih") if r.bias else None,
                             g("bias_hh") if r.bias else None))
            out.append(dirs)
        return out

    def forward(self, x: Tuple[RNNData, Lengths], hx: Optional[RNNState] = None
                ) -> Tuple[Tuple[RNNData, Lengths], RNNState]:
        """``((out, lengths), hid)`` exactly as rnn.py:133-185: rows past each
        sequence's length are 0, ``hid`` holds each sequence's last valid state."""
        _lib.require_gpu()
        inp, lengths = x
        if self.training and self.rnn.dropout > 0 and self.rnn.num_layers > 1:
            raise RuntimeError("inter-layer dropout in training mode is outside the inference hot path")
        lens_cpu = _lib.host_lens(lengths)
        if lens_cpu.numel() > 1 and bool((lens_cpu[:-1] < lens_cpu[1:]).any()):
            # pack_padded_sequence(enforce_sorted=True), rnn.py:170-175
            raise RuntimeError("`lengths` array must be sorted in decreasing order when `enforce_sorted` is True")
        max_len = int(lens_cpu[0]) if lens_cpu.numel() else 0
        data = _lib.f32c(inp.transpose(0, 1) if self.batch_first else inp)  # time-major [T, N, In]
        t, n, _ = data.shape
        if lens_cpu.numel() != n:
            raise RuntimeError(f"expected {n} lengths, got {lens_cpu.numel()}")
        if max_len < 1 or max_len > t or int(lens_cpu.min()) < 1:
            raise RuntimeError("lengths must be in [1, seq_len]")
        h0 = c0 = None
        if hx is not None:
            if self.rnn_type == RNNType.LSTM:
                h0, c0 = _lib.f32c(hx[0]), _lib.f32c(hx[1])
            else:
                h0 = _lib.f32c(hx)
        out, hn, cn = run_layers(_CELL[self.rnn_type], data, _lib.lens_i32(lens_cpu), max_len, self._layer_params(),
                                 self._packed, self.rnn.hidden_size, h0, c0, self._workspace, self.check_status,
                                 ragged=bool(int(lens_cpu[-1]) < max_len), state_inplace=self.inplace_state)
        if self.batch_first:
            out = out.transpose(0, 1)
        hid = (hn, cn) if self.rnn_type == RNNType.LSTM else hn
        return (out, lengths), hid

    def _init_hidden(self, batch: int, dtype: torch.dtype) -> RNNState:
        """rnn.py:187-205 (zeros; one tensor object shared by h0 and c0)."""
        zeros = torch.zeros(self.rnn.num_layers * (2 if self.bidirectional else 1), batch, self.rnn.hidden_size,
                            dtype=dtype)
        return (zeros, zeros) if self.rnn_type == RNNType.LSTM else zeros
