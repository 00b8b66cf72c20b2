"""ctypes binding of ``libms_hotpath.so`` (see ``include/ms_hotpath.h``).

The product path has no CPU fallback: if the library is missing, loading fails
loudly; if there is no HIP device, every op raises ``RuntimeError``.
"""
import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MS_HOTPATH_LIB: another build of the same library (same-box A/B runs of two kernel versions, tools/ab_lib.sh)
LIB_PATH = os.environ.get("MS_HOTPATH_LIB") or os.path.join(_HERE, "libms_hotpath.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ms_hotpath.h")

MS_OK = 0
ABI_VERSION = 4    # include/ms_hotpath.h MS_ABI_VERSION this binding was written against (checked in load())
ERR_NAMES = {1: "MS_ERR_INVALID", 2: "MS_ERR_HIP", 3: "MS_ERR_WORKSPACE", 4: "MS_ERR_TIMEOUT", 5: "MS_ERR_UNSUPPORTED"}

CELL_LSTM, CELL_GRU, CELL_RNN_TANH, CELL_HARD_LSTM = 0, 1, 2, 3
ACT_NONE, ACT_CLAMP = 0, 1
LINEAR_FEW_ROWS = 1   # MS_LINEAR_FEW_ROWS
PROF_KINDS = 9        # MS_PROF_KINDS: entries ms_prof_read writes into each of its two arrays

_P = c_void_p
_PP = POINTER(c_void_p)

# name -> (restype, argtypes); mirrors include/ms_hotpath.h one to one
SIGNATURES = {
    "ms_abi_version": (c_int, []),
    "ms_last_error": (c_char_p, []),
    "ms_mask_time_": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ms_maskconv_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "ms_maskconv_pack": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ms_maskconv_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 17 + [c_float, c_float, _P]),
    "ms_maskconv_cl_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ms_maskconv_cl_pack": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ms_maskconv_cl_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ms_maskconv_cl_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 16 + [c_float, c_float, _P, c_size_t, _P]),
    "ms_maskconv_fwin_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ms_maskconv_fwin_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ms_maskconv_fwin_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "ms_maskconv_fwin_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 14 + [c_float, c_float, _P, c_size_t, _P]),
    "ms_maskconv1d_gemm_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ms_maskconv1d_gemm_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ms_maskconv1d_gemm_workspace_bytes": (c_size_t, [c_int] * 5),
    "ms_maskconv1d_gemm_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 10 + [c_float, c_float, _P, c_size_t, _P]),
    "ms_nct_to_tnc": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ms_clamp": (c_int, [_P, _P, c_size_t, c_float, c_float, _P]),
    "ms_linear_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "ms_linear_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ms_linear_splitk_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, _P, c_size_t, _P]),
    "ms_linear_split_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ms_linear_split_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, _P]),
    "ms_linear_split_packed_bytes": (c_size_t, [c_int, c_int]),
    "ms_linear_split_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "ms_linear_split_forward_packed": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, _P]),
    "ms_lookahead_forward": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int] + [c_long] * 6 + [c_int, c_float, c_float, _P]),
    "ms_lookahead_window_forward": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int] + [c_long] * 6 + [c_int, c_float, c_float, _P]),
    "ms_rnn_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ms_ctc_status": (c_int, [_P, _P]),
    "ms_rnn_padded_hidden": (c_int, [c_int, c_int, c_int]),
    "ms_rnn_hx_preinit": (c_int, [c_int] * 8 + [_P, c_size_t, _P]),
    "ms_rnn_pack": (c_int, [c_int, c_int, c_int, c_int, _PP, _PP, _PP, _PP, _P, _P]),
    "ms_rnn_workspace_bytes": (c_size_t, [c_int] * 6),
    "ms_rnn_layer_forward": (c_int, [c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    "ms_rnn_layer_forward_ex": (c_int, [c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P] + [c_int] * 6 + [_P, c_size_t, _P]),
    "ms_rnn_layer_chains_planes": (c_int, [c_int, c_int, c_int]),
    "ms_rnn_layer_is_wide": (c_int, [c_int] * 4),
    "ms_rnn_stack_overlap_ok": (c_int, [c_int] * 7),
    "ms_rnn_stack_forward": (c_int, [c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P] + [c_int] * 7 + [_P, c_size_t, _P]),
    "ms_rnn_status": (c_int, [_P, _P]),
    "ms_rnn_debug_offset": (c_size_t, [c_int] * 6),
    "ms_rnn_layer_packs_rows": (c_int, [c_int] * 6),
    "ms_gemm_set_variant": (c_int, [c_int]),
    "ms_prof_enable": (c_int, [c_int]),
    "ms_prof_read": (c_int, [POINTER(c_float), POINTER(c_int)]),
    "ms_clock_probe": (c_int, [_P, c_int, c_int, _P]),
    "ms_barrier_chain_probe": (c_int, [_P, c_int, c_int, _P]),
    "ms_ctc_loss_workspace_bytes": (c_size_t, [c_int] * 4),
    "ms_ctc_loss_forward": (c_int, [_P] * 7 + [c_int] * 7 + [_P, c_size_t, _P]),
    "ms_log_softmax_axis": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ms_log_softmax_axis_backward": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "ms_ctc_loss_backward_workspace_bytes": (c_size_t, [c_int] * 4),
    "ms_ctc_loss_backward": (c_int, [_P] * 7 + [c_int] * 6 + [_P, c_size_t, _P]),
    "ms_ctc_greedy_decode": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ms_embedding_forward": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "ms_rnnt_joint_forward": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ms_rnnt_topk": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "ms_ctc_beam_workspace_bytes": (c_size_t, [c_int] * 4),
    "ms_ctc_beam_decode": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_int, c_int,
                                   _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "ms_rnnt_decode_workspace_bytes": (c_size_t, [c_int] * 10),
    "ms_rnnt_decode": (c_int, [_P, _P, _P, _PP, _PP, _PP, _PP, _P, _P, _P, _P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P]),
    "ms_mfcc_workspace_bytes": (c_size_t, [c_int] * 5),
    "ms_mfcc_forward": (c_int, [_P] * 7 + [c_int] * 7 + [c_float, _P, c_size_t, _P]),
    "ms_mfcc_legacy_forward": (c_int, [_P] * 7 + [c_int] * 8 + [c_double, _P]),
    "ms_standardize_workspace_bytes": (c_size_t, [c_int]),
    "ms_standardize_forward": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "ms_context_frames_forward": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ms_spec_augment_": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
}

_lib = None


def header_symbols():
    """Every function name declared in include/ms_hotpath.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C myrtlespeech_amd/csrc`). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.ms_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports MS_ABI_VERSION {got}, this binding needs {ABI_VERSION}: rebuild the "
                               "library (`make -C myrtlespeech_amd/csrc`)")
        _lib = lib
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("myrtlespeech_amd: a HIP device (MI355X) is required; there is no CPU fallback")


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, strided=False):
    """DEVICE pointer of a tensor (None -> NULL) for a C-ABI argument (``strided=True``: only for the entry points that take the
    tensor's element strides as arguments -- ms_lookahead_* -- and so may be handed a view).  Every pointer of include/ms_hotpath.h is a device
    pointer unless its name ends in ``_host`` and every kernel assumes dense storage, so anything else is refused HERE with an
    exception: a host tensor's ``data_ptr()`` handed to a kernel is a GPU page fault that takes the whole process down (round
    5: a test passed CPU tensors to ``run_layers``; the runtime aborted inside ``ms_rnn_status``), and a strided view would be
    read as if it were dense."""
    if t is None:
        return c_void_p(0)
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"_lib.ptr: expected a torch.Tensor or None, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"_lib.ptr: a {t.device.type} tensor (shape {tuple(t.shape)}) cannot be a device-pointer argument; "
                         "move it to the GPU first (host buffers go through _lib.host_ptr)")
    if not strided and not t.is_contiguous():
        raise ValueError(f"_lib.ptr: tensor of shape {tuple(t.shape)} and strides {tuple(t.stride())} is not contiguous; the "
                         "kernels read dense storage (call .contiguous())")
    return c_void_p(t.data_ptr())


def host_ptr(t):
    """HOST pointer of a CPU tensor (None -> NULL): only for arguments whose name ends in ``_host`` in include/ms_hotpath.h."""
    if t is None:
        return c_void_p(0)
    if not isinstance(t, torch.Tensor) or t.is_cuda or not t.is_contiguous():
        raise ValueError("_lib.host_ptr: expected a contiguous CPU tensor")
    return c_void_p(t.data_ptr())


def check(rc, what):
    if rc != MS_OK:
        msg = load().ms_last_error().decode()
        raise RuntimeError(f"{what} failed with {ERR_NAMES.get(rc, rc)}: {msg}")


def f32c(t):
    """float32, contiguous, on the GPU (torch only moves bytes here)."""
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_cuda:
        t = t.cuda()
    return t.contiguous()


# MS_ASYNC_LENS=0: blocking uploads and read-backs of the lengths (the behaviour before the side channel; for A/B runs)
_ASYNC_LENS = os.environ.get("MS_ASYNC_LENS") != "0"


# Constant length tensors for HIP-graph captures.  A captured forward (streaming.py) sees only lengths that are all equal (a
# steady-state chunk); made by ``torch.full`` inside the capture they become fill NODES that run at every replay -- about
# seven ~4.6 us launches per chunk (1 055 us).  The eager warm-up pass that precedes every capture leaves the constants here
# (made outside any capture, so they belong to no graph's pool), and the capture picks them up instead of recording fills.
# Entries are keyed on (shape, dtype, value) and dropped if somebody edited the tensor in place.
_const_lens = {}


def _const_tensor(shape, dtype, value, make: bool):
    key = (tuple(shape), dtype, int(value))
    rec = _const_lens.get(key)
    if rec is not None:
        t, version = rec
        if version_of(t) == version:
            return t
        # edited in place by somebody: graphs that hold it read the edited values (as they would with a tensor of their own);
        # the entry is kept alive but no longer handed out
        _const_lens[("stale", len(_const_lens)) + key] = rec
        del _const_lens[key]
    if not make or len(_const_lens) >= 4096:
        # (full: no new entries -- a captured graph holds the ADDRESS of every constant it picked up, so entries are never
        # evicted; a capture that finds none records a fill node as before)
        return None
    t = torch.full(tuple(shape), int(value), dtype=dtype, device="cuda")
    _const_lens[key] = (t, version_of(t))
    return t


def _all_equal_value(h: torch.Tensor):
    """The common value of a small integer host tensor whose elements are all equal, else None."""
    if h.numel() == 0 or h.numel() > 4096 or h.dtype not in (torch.int32, torch.int64):
        return None
    flat = h.reshape(-1)
    v = int(flat[0])
    return v if bool((flat == v).all()) else None


def upload(host: torch.Tensor, dtype=None) -> torch.Tensor:
    """Small host tensor -> device without stalling the host: a pageable ``.cuda()`` is stream-ordered AND blocks the host
    until the device gets there, i.e. it drains the launch queue in the middle of a forward.  Pinned + non_blocking does
    neither (torch's caching host allocator keeps the staging block alive until the copy has run)."""
    h = host.detach()
    if dtype is not None and h.dtype != dtype:
        h = h.to(dtype)
    h = h.contiguous()
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # inside a HIP graph capture (streaming.py) a host -> device copy would bake a host address into the graph; the
        # lengths of a steady-state chunk are all equal, and a fill kernel makes those on the device
        flat = h.reshape(-1)
        if flat.numel() and bool((flat == flat[0]).all()):
            c = _const_tensor(h.shape, h.dtype, flat[0].item(), make=False) if h.dtype in (torch.int32, torch.int64) else None
            return c if c is not None else torch.full(h.shape, flat[0].item(), dtype=h.dtype, device="cuda")
        raise RuntimeError("host values that differ cannot be uploaded inside a HIP graph capture")
    if torch.cuda.is_available():
        v = _all_equal_value(h)
        if v is not None:
            _const_tensor(h.shape, h.dtype, v, make=True)      # for a capture of the same forward that may follow
    if not _ASYNC_LENS:
        return h.cuda()
    try:
        h = h.pin_memory()
    except RuntimeError:
        return h.cuda()
    return h.to("cuda", non_blocking=True)


def attach_host(dev: torch.Tensor, host: torch.Tensor) -> torch.Tensor:
    """Lengths travel between modules on the device (the reference's convention) but drive host control flow (step
    counts, validation): the device tensor a module returns remembers the host values it was made from, so the next
    module does not read them back (a blocking copy = another drain of the launch queue).  The host copy is keyed on the
    device tensor's version counter and storage address (like the weight caches in ``model/rnn.py``): an in-place edit of
    the lengths between two modules (``lens -= k``, a masked update, the reference's float in-place ``out_lens``,
    ``cnn.py:191-197``) bumps ``_version`` and the stale host values are dropped in ``host_lens``."""
    dev._ms_host = (host.detach().clone(), version_of(dev), dev.data_ptr())
    return dev


def version_of(t: torch.Tensor) -> int:
    """Version counter of ``t``; tensors made under ``torch.inference_mode()`` do not track one (reading it raises) and
    cannot be edited in place outside inference mode, so they are keyed on their address alone."""
    return -1 if t.is_inference() else t._version


def host_lens(lens: torch.Tensor) -> torch.Tensor:
    """Host int64 values of a lengths tensor (from the side channel of ``attach_host`` when there is one)."""
    if not lens.is_cuda:
        return lens.detach().to(torch.int64)
    h = cached_host(lens) if _ASYNC_LENS else None
    if h is not None:
        return h.to(torch.int64)
    return lens.detach().to("cpu", torch.int64)


def cached_host(lens: torch.Tensor):
    """The host values attached to ``lens`` if they still describe it, else None (and the stale record is dropped)."""
    rec = getattr(lens, "_ms_host", None)
    if rec is None:
        return None
    h, version, address = rec
    if h.shape == lens.shape and version == version_of(lens) and address == lens.data_ptr():
        return h
    lens._ms_host = None  # edited in place (or re-pointed) since it was attached: the host values are stale
    return None


def lens_to_device(lens: torch.Tensor) -> torch.Tensor:
    """The reference's ``seq_lens.cuda()`` at the end of every module, without the blocking upload."""
    if lens.is_cuda:
        return lens
    return attach_host(upload(lens), lens.detach())


def lens_i32(lens):
    """int32 device copy of a lengths tensor."""
    if lens.is_cuda:
        if lens.dtype == torch.int32 and lens.is_contiguous():
            return lens
        h = cached_host(lens) if _ASYNC_LENS else None
        if h is not None:                      # the conversion kernel of a captured forward is a node of every replay
            v = _all_equal_value(h)
            if v is not None:
                c = _const_tensor(lens.shape, torch.int32, v, make=not torch.cuda.is_current_stream_capturing())
                if c is not None and torch.cuda.is_current_stream_capturing():
                    return c
        return lens.to(torch.int32).contiguous()
    return upload(lens, torch.int32)


def split_precision():
    """True unless MS_PRECISION=f32 (exact float32 MFMA everywhere)."""
    return os.environ.get("MS_PRECISION") != "f32"


# Issue-point hook: called by the recurrent stack after every layer's launches have been enqueued (and by DeepSpeech2 after
# its convolutions).  None except while ``pipeline.TwoBatchesInFlight`` runs, which uses it to alternate the host-side issue
# of two batches layer by layer.
issue_point = None


def at_issue_point():
    hook = issue_point
    if hook is not None:
        hook()


class Workspace:
    """Grow-only device scratch buffer owned by a module."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, zero: bool = True):
        if self.buf is None or self.buf.numel() < nbytes:
            # zero-filled: the recurrent layers keep a sticky time-out word in the first bytes (ms_rnn_status); pure scratch
            # (``zero=False``: the linear layers' operand planes) is not -- allocated inside a HIP-graph capture the fill
            # would be a node of every replay
            make = torch.zeros if zero else torch.empty
            self.buf = make(max(int(nbytes), 256), dtype=torch.uint8, device="cuda")
        return self.buf
