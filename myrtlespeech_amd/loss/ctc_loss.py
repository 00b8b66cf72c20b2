"""CTC loss: mirror of myrtlespeech/loss/ctc_loss.py.

``CTCLoss(blank, reduction, zero_infinity, dim)`` applied to
``((x[T,N,V], x_lens), (y, y_lens))`` = log-softmax over symbols followed by the
log-space CTC forward recursion (``ms_ctc_loss_forward``, one workgroup per
utterance).  ``log_softmax`` / ``ctc_loss`` attributes exist for repr/API parity
with the reference; they hold configuration only.  When the logits require grad the loss is
an autograd node whose backward is the alpha-beta posterior kernel (``ms_ctc_loss_backward``):
``loss.backward()`` fills ``x.grad`` as it does through the reference's LogSoftmax + CTCLoss.
"""
from typing import Tuple

import torch

from myrtlespeech_amd import _lib

_REDUCTION = {"none": 0, "mean": 1, "sum": 2}


class _CTCLossFunction(torch.autograd.Function):
    """Forward = ``ms_ctc_loss_forward``; backward = ``ms_ctc_loss_backward`` (logits gradient only), followed by
    ``ms_log_softmax_axis_backward`` when the constructor's ``dim`` is not the symbol axis."""

    @staticmethod
    def forward(ctx, x, run_forward, meta):
        out = run_forward(x)
        ctx.save_for_backward(x)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        m = ctx.meta
        lib = _lib.load()
        t, n, v = x.shape
        g = _lib.f32c(grad_out.detach())
        if m["red"] == 0:
            grad_nll = g.reshape(n)
        elif m["red"] == 2:
            grad_nll = g.reshape(1).expand(n)
        else:  # mean: loss = mean_n(nll_n / max(len_n, 1))
            grad_nll = g.reshape(1) / (m["yl_dev"].clamp(min=1).to(torch.float32) * n)
        grad_nll = grad_nll.contiguous()
        grad = torch.empty_like(x)
        nbytes = lib.ms_ctc_loss_backward_workspace_bytes(t, n, v, m["s_max"])
        ws = m["workspace"].get(nbytes)
        # `log_probs` (CTCLoss(dim != -1), ctc_loss.py:37-45): the values LogSoftmax(dim) produced over time or over the batch,
        # which torch.nn.CTCLoss took as log-probabilities; the gradient with respect to THEM comes out of the alpha-beta
        # kernel, LogSoftmax's own backward over that axis follows
        lp = m.get("log_probs")
        src = x if lp is None else lp
        _lib.check(lib.ms_ctc_loss_backward(_lib.ptr(src), _lib.ptr(m["xl_dev"]), _lib.ptr(m["y_dev"]), _lib.ptr(m["off_dev"]),
                                            _lib.ptr(m["yl_dev"]), _lib.ptr(grad_nll), _lib.ptr(grad), t, n, v, m["s_max"],
                                            m["blank"], m["zero_infinity"] | (0 if lp is None else 2), _lib.ptr(ws), nbytes,
                                            _lib.stream_ptr()),
                   "ms_ctc_loss_backward")
        if m.get("check_status"):     # the alpha / beta pipeline of the backward has the same bounded mailbox spin as the forward
            _lib.check(lib.ms_ctc_status(_lib.ptr(ws), _lib.stream_ptr()), "ms_ctc_loss_backward")
        if lp is not None:
            outer, axis, inner = m["axis_view"]
            _lib.check(lib.ms_log_softmax_axis_backward(_lib.ptr(lp), _lib.ptr(grad), _lib.ptr(grad), outer, axis, inner,
                                                        _lib.stream_ptr()), "ms_log_softmax_axis_backward")
        return grad, None, None


class CTCLoss(torch.nn.Module):
    """ctc_loss.py:7-101."""

    def __init__(self, blank: int = 0, reduction: str = "mean", zero_infinity: bool = False, dim: int = -1):
        super().__init__()
        if reduction not in _REDUCTION:
            raise ValueError(f"{reduction} is not a valid value for reduction")
        self.log_softmax = torch.nn.LogSoftmax(dim=dim)
        self.ctc_loss = torch.nn.CTCLoss(blank=blank, reduction=reduction, zero_infinity=zero_infinity)
        self.use_cuda = torch.cuda.is_available()
        self._workspace = _lib.Workspace()
        self._bwd_workspace = _lib.Workspace()
        # like RNN.check_status: after every forward ask ms_ctc_status (a 4-byte read-back and a stream synchronisation)
        # whether a wave of the alpha pipeline timed out, and raise instead of handing back a NaN loss.  A caller that must
        # not synchronise sets it to False and calls ``status()`` when it reads the loss.
        # COST in a training loop: one 4-byte device-to-host copy and a stream synchronisation per forward and
        # another per backward -- the host cannot run ahead of the device across the loss.  A loop that reads ``loss.item()``
        # every step synchronises there anyway; one that does not should set ``check_status = False`` and call ``status()``
        # wherever it does read the loss (the time-out word is sticky: nothing is lost by asking later).
        self.check_status = True

    def status(self) -> None:
        """Synchronises the current stream; raises RuntimeError if a loss kernel of this module timed out (reported once)."""
        lib = _lib.load()
        for ws in (self._workspace, self._bwd_workspace):
            if ws.buf is not None:
                _lib.check(lib.ms_ctc_status(_lib.ptr(ws.buf), _lib.stream_ptr()), "ms_ctc_loss_forward")

    def forward(self, inputs: Tuple[torch.Tensor, torch.Tensor], targets: Tuple[torch.Tensor, torch.Tensor]
                ) -> torch.Tensor:
        _lib.require_gpu()
        lib = _lib.load()
        x, x_lens = inputs
        y, y_lens = targets
        if x.dim() != 3:
            raise RuntimeError("inputs must be [max_seq_len, batch, features]")
        dim = self.log_softmax.dim
        if not -3 <= dim <= 2:
            raise IndexError(f"Dimension out of range (expected to be in range of [-3, 2], but got {dim})")
        dim %= 3
        x = _lib.f32c(x)
        t, n, v = x.shape
        log_probs_in = 0
        x_in, log_probs, axis_view = x, None, None
        if dim != 2:
            # ctc_loss.py:37-45 forwards ANY dim to LogSoftmax: the values normalised over time (0) or over the batch (1) are
            # what torch.nn.CTCLoss then takes as log-probabilities.  One extra pass; the kernel skips its own normalisation.
            xn = torch.empty_like(x)
            axis_view = (1, t, n * v) if dim == 0 else (t, n, v)
            _lib.check(lib.ms_log_softmax_axis(_lib.ptr(x), _lib.ptr(xn), *axis_view, _lib.stream_ptr()),
                       "ms_log_softmax_axis")
            log_probs, log_probs_in = xn.detach(), 2       # MS_CTC_LOG_PROBS_IN
        blank = self.ctc_loss.blank
        if not 0 <= blank < v:
            raise RuntimeError("blank must be in label range")
        xl = x_lens.detach().to("cpu", torch.int64)
        yl = y_lens.detach().to("cpu", torch.int64)
        if xl.numel() != n or yl.numel() != n:
            raise RuntimeError("input_lengths and target_lengths must be of size batch_size")
        xl_list, yl_list = xl.tolist(), yl.tolist()
        if n and (max(xl_list) > t or min(xl_list) < 0):
            raise RuntimeError("input lengths must be in [0, max_seq_len]")
        yl_max = max(yl_list) if n else 0
        if y.dim() == 2:  # padded [N, S]
            s_pad = y.shape[1]
            if yl_max > s_pad:
                raise RuntimeError("target length exceeds the padded target width")
            offsets = torch.arange(n, dtype=torch.int64) * s_pad
        elif y.dim() == 1:  # concatenated
            offsets = torch.cumsum(yl, 0) - yl
            if sum(yl_list) > y.numel():
                raise RuntimeError("sum(target_lengths) exceeds the number of targets")
        else:
            raise RuntimeError("targets must be [batch, max_target_len] or 1-D")
        s_max = 2 * yl_max + 1
        # ONE staged upload for everything that starts on the host (lengths, offsets and -- when the caller keeps them
        # there -- the targets): four separate small copies cost more host time than the loss kernel takes
        host_parts = [xl.to(torch.int32), offsets.to(torch.int32), yl.to(torch.int32)]
        y_on_host = not y.is_cuda
        if y_on_host and y.numel():
            host_parts.append(y.detach().to(torch.int32).reshape(-1))
        packed = _lib.upload(torch.cat(host_parts))
        xl_dev, off_dev, yl_dev = packed[:n], packed[n:2 * n], packed[2 * n:3 * n]
        if y_on_host and y.numel():
            y_dev = packed[3 * n:]
        elif y.numel():
            y_dev = y.detach().to(dtype=torch.int32).contiguous().reshape(-1)
        else:
            y_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
        red = _REDUCTION[self.ctc_loss.reduction]
        zero_inf = int(bool(self.ctc_loss.zero_infinity))
        fwd_flags = zero_inf | log_probs_in

        def run_forward(logits: torch.Tensor) -> torch.Tensor:
            if log_probs is not None:
                logits = log_probs
            nll = torch.empty(n, dtype=torch.float32, device="cuda")
            reduced = torch.empty(1, dtype=torch.float32, device="cuda")
            ws = self._workspace.get(lib.ms_ctc_loss_workspace_bytes(t, n, v, s_max))
            _lib.check(lib.ms_ctc_loss_forward(_lib.ptr(logits), _lib.ptr(xl_dev), _lib.ptr(y_dev), _lib.ptr(off_dev),
                                               _lib.ptr(yl_dev), _lib.ptr(nll), _lib.ptr(reduced), t, n, v, s_max, blank,
                                               red, fwd_flags, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "ms_ctc_loss_forward")
            if self.check_status:
                _lib.check(lib.ms_ctc_status(_lib.ptr(ws), _lib.stream_ptr()), "ms_ctc_loss_forward")
            return nll if red == 0 else reduced[0]

        if torch.is_grad_enabled() and x.requires_grad:
            meta = dict(red=red, s_max=s_max, blank=blank, zero_infinity=zero_inf, xl_dev=xl_dev, off_dev=off_dev,
                        yl_dev=yl_dev, y_dev=y_dev, workspace=self._bwd_workspace, log_probs=log_probs, axis_view=axis_view,
                        check_status=self.check_status)
            return _CTCLossFunction.apply(x_in, run_forward, meta)
        return run_forward(x)
