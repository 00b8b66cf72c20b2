"""Chunked (streaming) Deep Speech 2 inference (new -- the reference only has the
``hx`` in / ``hid`` out plumbing of rnn.py:133-185 and deep_speech_2.py:123-172, no chunker).

Two contracts, both pinned by fixtures made with the reference itself:

``ChunkedDeepSpeech2(model, chunk_frames)`` -- *the reference's plumbing, chunk by chunk*
(``tests/golden/stream_*.npz``, ``cfg5_stream*_summary.npz``): the input is cut into consecutive
``chunk_frames``-frame slices; every slice goes through the ordinary ``DeepSpeech2.forward`` with the
recurrent state returned by the previous slice.  Convolutions therefore see each slice on its own (SAME
padding at the slice edges) and the backward direction of a bidirectional RNN restarts per slice -- by
construction, exactly as a caller of the reference would get.  Utterances that have ended leave the batch
(lengths are sorted in decreasing order, so they are a suffix); their final state is kept.

``ChunkedDeepSpeech2(model, chunk_frames, carry_context=True)`` -- *the full-utterance result, delivered in
chunks* (unidirectional stacks; pinned by the reference's FULL-utterance outputs, ``ds2_tiny_gru_lookahead.npz``
and ``ds2_shipped_summary.npz``): every convolution keeps the input frames its next output frame still needs
(``kernel - 1 - left pad`` of them at the stride phase it has reached) and emits an output frame only once all
of its taps have arrived; SAME padding exists only at the true start and end of the utterance, with the
left / right split the reference computes from the batch's padded length (cnn.py:148-163); the recurrent state
is carried; the lookahead (lookahead.py:36-70) holds back ``context - 1`` frames.  The concatenated chunk
outputs are the logits ``DeepSpeech2.forward`` gives on the whole clip.  Algorithmic latency (frames of input
that must have arrived before the first logit row can be emitted) is ``ChunkedDeepSpeech2.latency_frames()``: for
the shipped config (conv 11 x stride 2, conv 11 x stride 1, lookahead 80) 174 input frames -- frame 0 itself plus
5 (conv1's right context) + 2 * 5 (conv2's, through conv1's stride) + 2 * 79 (the lookahead's) = 1.73 s of future
audio, 1.58 s of which are the lookahead's.
"""
from typing import List, Optional, Tuple

import torch

from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.cnn import (Conv1dTo2d, Conv2dTo1d, MaskConv1d, MaskConv2d, PaddingMode, _conv_forward, _pair,
                                        out_lens, pad_same)
from myrtlespeech_amd.model.lookahead import Lookahead, lookahead_apply
from myrtlespeech_amd.model.utils import activation_clamp


class _ConvStage:
    """One masked convolution of the stack as a stream operator over time.

    Global frame indices: the layer's input is ``total_in`` frames long (the batch's padded length at this depth);
    output frame ``u`` reads input frames ``u * stride - left + tap * dilation``; frames below 0, at or past an
    utterance's own length (the reference's in-place mask, cnn.py:425-443) and at or past ``total_in`` read as zero.
    ``cache`` holds the received input frames from index ``next_out * stride - left`` on."""

    def __init__(self, conv, clamp, total_in: int, lens_in: torch.Tensor):
        self.conv, self.clamp = conv, clamp
        self.is1d = isinstance(conv, MaskConv1d)
        kt = conv.kernel_size[0] if self.is1d else conv.kernel_size[1]
        self.stride = conv.stride[0] if self.is1d else _pair(conv.stride)[1]
        self.dilation = conv.dilation[0] if self.is1d else _pair(conv.dilation)[1]
        self.span = self.dilation * (kt - 1) + 1
        self.same = conv._padding_mode == PaddingMode.SAME
        self.left, self.right = pad_same(total_in, kt, self.stride, self.dilation) if self.same else (0, 0)
        self.total_in = total_in
        self.total_out = (total_in + self.left + self.right - self.span) // self.stride + 1
        self.lens_in = lens_in
        self.lens_out = out_lens(lens_in, kt, self.stride, self.dilation, self.left + self.right)
        self.next_out = 0
        self.cache: Optional[torch.Tensor] = None     # [N, C, F, frames] (conv2d) or [N, C, frames] (conv1d)
        self.cache_start = -self.left
        self.skip = 0        # frames still to arrive that no output reads (stride > span: the gaps between the windows)

    def push(self, new: Optional[torch.Tensor], final: bool) -> Optional[torch.Tensor]:
        """``new``: the next input frames (time last) or None; returns the output frames that became computable."""
        parts = []
        skip_in = self.skip
        if new is not None and self.skip:
            drop = min(self.skip, new.shape[-1])
            new, self.skip = new[..., drop:], self.skip - drop
        if self.cache is None:
            ref = new
            if ref is None:
                return None
            if self.left:
                parts.append(ref.new_zeros(ref.shape[:-1] + (self.left,)))
        else:
            parts.append(self.cache)
        if new is not None and (new.shape[-1] or not parts):
            parts.append(new)
        window = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
        received = self.cache_start + window.shape[-1]
        if final:
            if received > self.total_in:                      # frames past the batch's padded length: not part of the clip
                window = window[..., :window.shape[-1] - (received - self.total_in)]
                received = self.total_in
            tail = self.total_in + self.right - received
            if tail > 0:
                window = torch.cat([window, window.new_zeros(window.shape[:-1] + (tail,))], dim=-1)
        lw = window.shape[-1]
        cnt = 0 if lw < self.span else (lw - self.span) // self.stride + 1
        cnt = min(cnt, self.total_out - self.next_out)
        out = None
        if cnt > 0:
            used = (cnt - 1) * self.stride + self.span          # the window's frames these outputs read
            win = window[..., :used].contiguous()
            lens_w = (self.lens_in.to(torch.int64) - self.cache_start).clamp(min=0, max=used)
            x4 = win.unsqueeze(2) if self.is1d else win
            w4 = self.conv.weight.unsqueeze(2) if self.is1d else self.conv.weight
            stride = (1, self.stride) if self.is1d else _pair(self.conv.stride)
            dil = (1, self.dilation) if self.is1d else _pair(self.conv.dilation)
            y, _ = _conv_forward(x4, lens_w, w4, self.conv.bias, self.conv._packed, stride, dil, self.conv.groups, self.same,
                                 self.clamp, time_pads=(0, 0))
            out = y.squeeze(2) if self.is1d else y
            self.next_out += cnt
            self.cache_start += cnt * self.stride
            if cnt * self.stride > lw:                # the next window starts at a frame that has not arrived yet
                self.skip = cnt * self.stride - lw
                window = window[..., lw:]
            else:
                window = window[..., cnt * self.stride:]
        self.cache = window
        # what this push did, for the steady-state test of the graph path: (cache frames before, after, outputs, frames read)
        # (frames skipped at either end of the push count: with stride > span a push that STARTED inside a gap consumed fewer
        # frames than the next one of the same size will -- found by tests/soak.py, kernel 1 / stride 3 / chunks of 5)
        self.last = (lw - (0 if new is None else new.shape[-1]), window.shape[-1], cnt,
                     (cnt - 1) * self.stride + self.span if cnt > 0 else 0, self.skip + skip_in, final)
        return out


class _ContextStream:
    """State of one batch streaming through ``model`` with carried context (see the module docstring)."""

    def __init__(self, model, n: int, total_frames: int, lens: torch.Tensor, hx=None):
        from myrtlespeech_amd.model.deep_speech_2 import _is_plain_activation
        if getattr(model.rnn, "bidirectional", False):
            raise ValueError("carry_context=True reproduces the full-utterance result, which a bidirectional recurrent layer "
                             "cannot give chunk by chunk (its backward direction starts at the utterance's last frame)")
        self.model, self.n = model, n
        self.ops: List = []                # ("conv", _ConvStage) | ("reshape", module)
        mods = [] if model.cnn is None else (list(model.cnn) if isinstance(model.cnn, torch.nn.Sequential) else [model.cnn])
        total, cur_lens = total_frames, lens.detach().to("cpu", torch.int64)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (MaskConv1d, MaskConv2d)):
                clamp = None
                if i + 1 < len(mods) and _is_plain_activation(mods[i + 1]):
                    clamp = activation_clamp(mods[i + 1].module)
                    i += 1
                st = _ConvStage(m, clamp, total, cur_lens)
                self.ops.append(("conv", st))
                total, cur_lens = st.total_out, st.lens_out
            elif isinstance(m, (Conv2dTo1d, Conv1dTo2d)):
                self.ops.append(("reshape", m))
            elif _is_plain_activation(m) and isinstance(m.module, torch.nn.Identity):
                pass
            else:
                raise ValueError(f"carry_context=True does not know how to stream {type(m).__name__}")
            i += 1
        self.total_out = total                       # frames the recurrent stack sees (= rows of the logits)
        self.out_lens = cur_lens
        self.rnn_pos = 0
        self.full_state = None
        if hx is not None:                           # the caller's initial state (deep_speech_2.py:123-127), one row per utterance
            self.full_state = tuple(_lib.f32c(s).clone() for s in hx) if isinstance(hx, tuple) else _lib.f32c(hx).clone()
        self.state = self.full_state
        la = model.lookahead
        self.la = self.la_clamp = None
        if la is not None:
            lm = list(la) if isinstance(la, torch.nn.Sequential) else [la]
            if not isinstance(lm[0], Lookahead) or len(lm) > 2 or (len(lm) == 2 and not _is_plain_activation(lm[1])):
                raise ValueError("carry_context=True needs the builder's Sequential(Lookahead, SeqLenWrapper(act))")
            self.la = lm[0]
            self.la_clamp = activation_clamp(lm[1].module) if len(lm) == 2 else None
        self.la_cache: Optional[torch.Tensor] = None      # [frames, N, F] recurrent outputs not yet fully consumed
        self.emitted = 0
        self.last_push = None        # (chunk shape, rows into the recurrent stack, rows emitted) of the last eager push
        self.graph = None            # the _ContextGraph this stream is attached to
        self.graph_ok = True         # False once the stream has left steady state (or a capture failed)

    def latency_frames(self) -> int:
        """Input frames that must have arrived before logit row 0 can be emitted."""
        need = 1 + (0 if self.la is None else self.la.context - 1)     # rows of the recurrent output
        for kind, st in reversed(self.ops):
            if kind == "conv":
                need = (need - 1) * st.stride + st.span - st.left
        return max(need, 0)

    def push(self, chunk: Optional[torch.Tensor], final: bool) -> Optional[torch.Tensor]:
        """``chunk [N, C, F, frames]`` (or None with ``final``): returns the logit rows ``[rows, N, V]`` that became
        computable (None if there are none yet)."""
        h = None if chunk is None else _lib.f32c(chunk)
        for kind, op in self.ops:
            if kind == "conv":
                h = op.push(h, final)
            elif h is not None:
                h = op(h) if not getattr(op, "seq_len_support", False) else op((h, None))[0]
        rows = None
        if h is not None:
            rows = self._recurrent(self.model._conv_to_rnn_size(h))
        out = self._output(rows, final)
        self.last_push = None if (chunk is None or final) else \
            (tuple(chunk.shape), 0 if rows is None else rows.shape[0], 0 if out is None else out.shape[0])
        return out

    def conv_stages(self):
        return [op for kind, op in self.ops if kind == "conv"]

    def steady_signature(self, chunk_shape):
        """Key of the steady state this stream is in, or None: the last (eager) push took a chunk of this shape, every
        convolution gave outputs and was left with as many cached frames as it started with (nothing skipped), the recurrent
        stack got rows and the lookahead passed on as many rows as it received -- the next push of the same shape then does the
        same launches on the same shapes, provided no utterance ends inside it (``_ContextGraph.can_replay``)."""
        lp = self.last_push
        if lp is None or lp[0] != tuple(chunk_shape) or lp[1] == 0 or lp[1] != lp[2] or self.full_state is None:
            return None
        sig = [tuple(chunk_shape), self.n, lp[1]]
        for st in self.conv_stages():
            before, after, cnt, used, skip, final = st.last
            if final or skip or cnt == 0 or before != after or st.cache is None:
                return None
            sig.append((after, cnt, used))
        if self.la is not None:
            if self.la_cache is None or self.la_cache.shape[0] != self.la.context - 1:
                return None
        return tuple(sig)

    def _recurrent(self, seq: torch.Tensor) -> torch.Tensor:
        """``seq [cnt, N, CF]``: rows ``rnn_pos ..`` of the recurrent stack's input; returns ``[cnt, N, H]`` with zero rows
        for utterances that have ended (pad_packed_sequence, rnn.py:179-183)."""
        cnt = seq.shape[0]
        lens_c = (self.out_lens - self.rnn_pos).clamp(min=0, max=cnt)
        alive = int((lens_c > 0).sum())
        self.rnn_pos += cnt
        hidden = self.model.rnn.rnn.hidden_size
        if alive == 0:
            return torch.zeros((cnt, self.n, hidden), dtype=torch.float32, device="cuda")
        hx = None
        if self.state is not None:
            hx = tuple(s[:, :alive].contiguous() for s in self.state) if isinstance(self.state, tuple) else \
                self.state[:, :alive].contiguous()
        (y, _), st = self.model.rnn((seq[:, :alive].contiguous(), lens_c[:alive]), hx=hx)
        if self.full_state is None:
            mk = lambda s: torch.zeros(s.shape[0], self.n, s.shape[2], device=s.device)  # noqa: E731
            self.full_state = tuple(mk(s) for s in st) if isinstance(st, tuple) else mk(st)
        for fs, s in zip(self.full_state if isinstance(st, tuple) else (self.full_state,), st if isinstance(st, tuple) else (st,)):
            if not (alive == self.n and fs.data_ptr() == s.data_ptr()):
                fs[:, :alive] = s
        self.state = self.full_state
        if alive == self.n:            # every utterance alive: the stack's output IS the block (no zero fill, no copy)
            return y
        block = torch.zeros((cnt, self.n, hidden), dtype=torch.float32, device="cuda")
        block[:, :alive] = y
        return block

    def _output(self, rows: Optional[torch.Tensor], final: bool) -> Optional[torch.Tensor]:
        from myrtlespeech_amd.model.fully_connected import FullyConnected, linear_stack_plan, run_linear_stack
        fc = self.model.fully_connected
        if self.la is None:
            if rows is None:
                return None
            ready = rows
        else:
            parts = [p for p in (self.la_cache, rows) if p is not None and p.shape[0]]
            if not parts:
                return None
            window = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
            frames = window.shape[0]
            emit = frames if final else max(frames - (self.la.context - 1), 0)
            self.la_cache = window[emit:]
            if emit == 0:
                return None
            window = window.contiguous()
            f = window.shape[2]
            ntf = lookahead_apply(window, self.la.weight, (f, 1, self.n * f), self.n, f, frames, out_layout="ntf",
                                  clamp=self.la_clamp, t_out=emit)
            ready = ntf.transpose(0, 1)                          # [emit, N, F] view
        t = ready.shape[0]
        self.emitted += t
        if isinstance(fc, FullyConnected):
            flat = _lib.f32c(ready).reshape(t * self.n, ready.shape[2])
            y = run_linear_stack(flat, linear_stack_plan(fc.fully_connected, fc.training))
            return y.reshape(t, self.n, -1)
        out, _ = fc((ready.transpose(0, 1), self.out_lens))
        return out.transpose(0, 1)


def _pipeline_running() -> bool:
    """A ``pipeline.BatchesInFlight`` call is in progress in this process.  The library serialises persistent recurrent launches
    of different streams by chaining them with events at ENQUEUE time (``PersistentTurn``, csrc/rnn.hip) -- which a captured
    graph skips (nothing runs at capture time, and the replay stream is not the capture stream).  A graph holding persistent
    launches replayed beside another stream's persistent launch could lose co-residency and run into the 2 s time-out
    (ADVICE r4), so while a pipeline is running no NEW graph is started: those calls take the eager path (same bits)."""
    return _lib.issue_point is not None


def _workspace_slots(model):
    """(owner, attribute name) of every grow-only scratch buffer a forward pass of ``model`` uses."""
    slots = []
    for m in model.modules():
        if isinstance(getattr(m, "_workspace", None), _lib.Workspace):
            slots.append((m, "_workspace"))
        pk = getattr(m, "_packed", None)
        if pk is not None and isinstance(getattr(pk, "workspace", None), _lib.Workspace):
            slots.append((pk, "workspace"))
    return slots


class _ContextGraph:
    """One steady-state ``push`` of a carried-context stream -- every utterance alive, a whole chunk, every convolution's
    cache and the lookahead's held-back rows at their steady sizes -- captured once as a HIP graph and replayed per chunk.
    The eager push is ~40 launches plus the Python of the stream operators (concatenations, slices, length uploads): about a
    millisecond of host time per chunk around ~0.5 ms of kernels; a replay costs the host one input copy, one
    ``hipGraphLaunch`` and one clone of the logit rows.

    What a push carries from chunk to chunk lives in STATIC tensors the graph owns: the convolutions' cached input frames,
    the lookahead's held-back rows, the recurrent state.  The capture records ``push`` itself (same kernels, same order, same
    shapes as the eager call: ``torch.equal`` results) followed by copies of the new caches back into the static tensors; a
    capture executes nothing, so the stream's host-side counters are put back afterwards and advanced per replay by the
    amounts the captured push advanced them.  A stream attaches to a graph of its signature by copying its caches / state in;
    it leaves the graph (for good) at the first push that is not steady -- an utterance ending inside it, the final flush."""

    def __init__(self, stream: "_ContextStream", chunk_shape):
        import os
        model = stream.model
        self.chunk_shape = tuple(chunk_shape)
        self.x = torch.zeros(self.chunk_shape, dtype=torch.float32, device="cuda")
        stages = stream.conv_stages()
        self.caches = [st.cache.contiguous().clone() for st in stages]
        self.la_cache = None if stream.la is None else stream.la_cache.contiguous().clone()
        fs = stream.full_state
        self.state = tuple(s_.clone() for s_ in fs) if isinstance(fs, tuple) else fs.clone()
        self.attach(stream, copy=False)
        snap = self._counters(stream)
        slots = _workspace_slots(model) if os.environ.get("MS_STREAM_GRAPH_OWN_WS") != "0" else []
        self._own = [_lib.Workspace() for _ in slots]
        saved = [getattr(o, a) for o, a in slots]
        for w_, old in zip(self._own, saved):          # sized like the model's (grown by the eager pushes before this one):
            if old.buf is not None:                    # no allocation, hence no zero-fill node, inside the capture
                w_.get(old.buf.numel())
        check_was = getattr(model.rnn, "check_status", None)
        inplace_was = getattr(model.rnn, "inplace_state", None)
        self.graph = torch.cuda.CUDAGraph()
        from myrtlespeech_amd.model import fully_connected as _fc
        self._fc_ws = _lib.Workspace()          # the linear layers' plane scratch of THIS graph (not the per-stream LRU)
        old_fc = _fc._stream_workspace()
        if old_fc.buf is not None:
            self._fc_ws.get(old_fc.buf.numel(), zero=False)      # sized by the eager pushes before this one: no allocation inside the capture
        scratch = _fc.graph_scratch(self._fc_ws)
        scratch.__enter__()
        try:
            for (o, a), w_ in zip(slots, self._own):
                setattr(o, a, w_)
            if check_was is not None:
                model.rnn.check_status = False          # ms_rnn_status synchronises: not inside a capture
            if inplace_was is not None:
                model.rnn.inplace_state = True          # the new state lands in the static state tensors: no copies in the graph
            torch.cuda.synchronize()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                y = stream.push(self.x, False)
                for st, c in zip(stages, self.caches):
                    c.copy_(st.cache)
                if self.la_cache is not None:
                    self.la_cache.copy_(stream.la_cache)
            if y is None or any(st.cache.shape != c.shape for st, c in zip(stages, self.caches)):
                raise RuntimeError("the captured push was not a steady-state push")
        except BaseException:
            # nothing ran on the device: the static clones still hold what the stream held, the counters go back
            self._restore(stream, snap)
            self.attach(stream, copy=False)
            stream.graph = None
            raise
        finally:
            for (o, a), w_ in zip(slots, saved):
                setattr(o, a, w_)
            if check_was is not None:
                model.rnn.check_status = check_was
            if inplace_was is not None:
                model.rnn.inplace_state = inplace_was
            scratch.__exit__(None, None, None)
        self.y = y
        after = self._counters(stream)
        self.delta = [b - a for a, b in zip(snap, after)]
        self.used = [st.last[3] for st in stages]
        self.cnt = [st.last[2] for st in stages]
        self.rows = stream.last_push[1]
        self._restore(stream, snap)
        self.attach(stream, copy=False)
        rnn_ws = [w_ for (o, a), w_ in zip(slots, self._own) if o is getattr(model, "rnn", None)]
        self.rnn_ws = rnn_ws[0] if rnn_ws else None

    @staticmethod
    def _counters(stream):
        v = []
        for st in stream.conv_stages():
            v += [st.next_out, st.cache_start]
        return v + [stream.rnn_pos, stream.emitted]

    @staticmethod
    def _restore(stream, v):
        it = iter(v)
        for st in stream.conv_stages():
            st.next_out, st.cache_start = next(it), next(it)
        stream.rnn_pos, stream.emitted = next(it), next(it)

    def attach(self, stream, copy=True):
        """Point ``stream`` at this graph's static tensors (``copy``: after moving the stream's current contents in)."""
        stages = stream.conv_stages()
        if copy:
            for st, c in zip(stages, self.caches):
                c.copy_(st.cache)
            if self.la_cache is not None:
                self.la_cache.copy_(stream.la_cache)
            fs = stream.full_state
            for a_, b_ in zip(self.state if isinstance(fs, tuple) else (self.state,), fs if isinstance(fs, tuple) else (fs,)):
                a_.copy_(b_)
        for st, c in zip(stages, self.caches):
            st.cache = c
        if self.la_cache is not None:
            stream.la_cache = self.la_cache
        stream.full_state = stream.state = self.state
        stream.graph = self

    def detach(self, stream):
        """The stream goes on eagerly with private copies (another stream may attach to this graph and overwrite its tensors)."""
        for st, c in zip(stream.conv_stages(), self.caches):
            st.cache = c.clone()
        if self.la_cache is not None:
            stream.la_cache = self.la_cache.clone()
        stream.full_state = stream.state = tuple(s_.clone() for s_ in self.state) if isinstance(self.state, tuple) \
            else self.state.clone()
        stream.graph, stream.graph_ok = None, False

    @staticmethod
    def fits(stream, used, cnt, rows) -> bool:
        """No utterance ends inside what a steady push reads (every length clamp of the eager push is inactive) and no
        layer reaches the end of the clip."""
        for st, u, c in zip(stream.conv_stages(), used, cnt):
            if int(st.lens_in.min()) - st.cache_start < u or st.total_out - st.next_out < c:
                return False
        return int(stream.out_lens.min()) - stream.rnn_pos >= rows

    def can_replay(self, stream, chunk) -> bool:
        """The next push is the captured one."""
        return tuple(chunk.shape) == self.chunk_shape and self.fits(stream, self.used, self.cnt, self.rows)

    def replay(self, stream, chunk):
        self.x.copy_(chunk)
        self.graph.replay()
        self._restore(stream, [a + d for a, d in zip(self._counters(stream), self.delta)])
        return self.y.clone()


class _ChunkGraph:
    """One steady-state slice -- ``alive`` utterances, all with a full ``frames``-frame slice, state carried -- captured once as
    a HIP graph (``torch.cuda.CUDAGraph``) and replayed per chunk: the ~45 launches of a slice then cost the host one
    ``hipGraphLaunch`` instead of ~1 ms of Python / ctypes work per chunk, and the device does exactly
    the launches the eager call enqueues (same kernels, same order, same buffers: bit-identical results).  The recurrent
    state lives in the graph's own static tensors and is chained inside the graph (state_in <- state_out)."""

    def __init__(self, model, alive: int, shape, state):
        self.x = torch.zeros((alive,) + tuple(shape[1:]), dtype=torch.float32, device="cuda")
        self.state = tuple(torch.zeros_like(s[:, :alive]) for s in state) if isinstance(state, tuple) else \
            torch.zeros_like(state[:, :alive])
        frames = shape[-1]
        host = torch.full((alive,), frames, dtype=torch.int64)
        self.lens = _lib.attach_host(torch.full((alive,), frames, dtype=torch.int64, device="cuda"), host)
        self.graph = torch.cuda.CUDAGraph()
        # The graph records POINTERS.  The model's grow-only scratch buffers (recurrent workspace, convolution planes) may be
        # re-allocated by a later, larger call of the same model -- the graph would then write into freed memory -- so the
        # graph gets scratch buffers of its own: fresh Workspace objects are swapped in, one eager forward sizes them (and
        # does every first-use set-up of the library outside the capture), the capture records them, and the model's own
        # objects are put back.  (Packed weights are keyed on the parameters' versions: ChunkedDeepSpeech2 drops its graphs
        # when a parameter changes.)
        import os
        slots = _workspace_slots(model) if os.environ.get("MS_STREAM_GRAPH_OWN_WS") != "0" else []     # (0: A/B runs only)
        self._own = [_lib.Workspace() for _ in slots]
        saved = [getattr(o, a) for o, a in slots]
        rnn = getattr(model, "rnn", None)
        inplace_was = getattr(rnn, "inplace_state", None)
        from myrtlespeech_amd.model import fully_connected as _fc
        self._fc_ws = _lib.Workspace()          # the linear layers' plane scratch of THIS graph (not the per-stream LRU)
        scratch = _fc.graph_scratch(self._fc_ws)
        scratch.__enter__()
        try:
            for (o, a), w in zip(slots, self._own):
                setattr(o, a, w)
            if inplace_was is not None:
                rnn.inplace_state = True       # h_n / c_n land in the static state tensors themselves: no copies in the graph
            model((self.x, self.lens), self.state)
            torch.cuda.synchronize()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                (y, ol), st = model((self.x, self.lens), self.state)
                for a_, b_ in zip(self.state if isinstance(st, tuple) else (self.state,), st if isinstance(st, tuple) else (st,)):
                    if a_.data_ptr() != b_.data_ptr():
                        a_.copy_(b_)
        finally:
            for (o, a), w in zip(slots, saved):
                setattr(o, a, w)
            if inplace_was is not None:
                rnn.inplace_state = inplace_was
            scratch.__exit__(None, None, None)
        self.y, self.out_lens_host = y, _lib.host_lens(ol).clone()
        rnn_ws = [w for (o, a), w in zip(slots, self._own) if o is getattr(model, "rnn", None)]
        self.rnn_ws = rnn_ws[0] if rnn_ws else None      # holds the sticky time-out word of this graph's recurrent launches

    def load_state(self, state, alive: int):
        if isinstance(state, tuple):
            for a, b in zip(self.state, state):
                a.copy_(b[:, :alive])
        else:
            self.state.copy_(state[:, :alive])

    def run(self, xc: torch.Tensor):
        self.x.copy_(xc)
        self.graph.replay()
        return self.y


class GraphedForward:
    """``model((x, lens), hx)`` for inputs of ONE shape whose lengths are all equal (a single clip, a full batch) as a captured
    HIP graph: the forward's ~20 launches cost the host one input copy and one ``hipGraphLaunch`` (DeepSpeech1 on one 4 s clip
    is ~0.75 ms of kernels inside ~1.0 ms of eager Python: the device idles a quarter of the call).  Same launches on the
    same buffers as the eager call, so the results are its bits (``tests/test_gpu_configs.py``).  Any other input -- ragged
    lengths, a shape it has not seen while ``max_graphs`` are held, a capture that fails -- runs eagerly.  The graph owns its
    scratch buffers (see ``_ChunkGraph``) and is dropped when a parameter changes.  Outputs are copies: they stay valid after
    the next call."""

    def __init__(self, model, max_graphs: int = 4):
        self.model, self.max_graphs = model, max_graphs
        self._graphs = {}
        self._sig = None
        self.graph_error: Optional[str] = None
        self.replays = 0

    class _Graph:
        def __init__(self, model, shape, length: int, hx):
            import os
            n = shape[0]
            self.x = torch.zeros(tuple(shape), dtype=torch.float32, device="cuda")
            host = torch.full((n,), length, dtype=torch.int64)
            self.lens = _lib.attach_host(torch.full((n,), length, dtype=torch.int64, device="cuda"), host)
            self.hx = None if hx is None else (tuple(torch.zeros_like(_lib.f32c(s_)) for s_ in hx) if isinstance(hx, tuple)
                                               else torch.zeros_like(_lib.f32c(hx)))
            slots = _workspace_slots(model) if os.environ.get("MS_STREAM_GRAPH_OWN_WS") != "0" else []
            self._own = [_lib.Workspace() for _ in slots]
            saved = [getattr(o, a) for o, a in slots]
            checked = [m_ for m_ in model.modules() if getattr(m_, "check_status", False) is True]
            self.graph = torch.cuda.CUDAGraph()
            from myrtlespeech_amd.model import fully_connected as _fc
            self._fc_ws = _lib.Workspace()      # the linear layers' plane scratch of THIS graph (not the per-stream LRU)
            scratch = _fc.graph_scratch(self._fc_ws)
            scratch.__enter__()
            try:
                for (o, a), w_ in zip(slots, self._own):
                    setattr(o, a, w_)
                for m_ in checked:
                    m_.check_status = False              # ms_rnn_status synchronises: not inside a capture
                model((self.x, self.lens), self.hx)      # sizes the scratch, does the library's first-use set-up, fills the constants
                torch.cuda.synchronize()
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    (self.y, self.out_lens), self.hid = model((self.x, self.lens), self.hx)
            finally:
                for (o, a), w_ in zip(slots, saved):
                    setattr(o, a, w_)
                for m_ in checked:
                    m_.check_status = True
                scratch.__exit__(None, None, None)
            self.status_ws = [w_ for (o, a), w_ in zip(slots, self._own) if o in checked]

        def run(self, x, hx):
            self.x.copy_(x)
            if hx is not None:
                for a_, b_ in zip(self.hx if isinstance(hx, tuple) else (self.hx,), hx if isinstance(hx, tuple) else (hx,)):
                    a_.copy_(b_)
            self.graph.replay()
            hid = tuple(h_.clone() for h_ in self.hid) if isinstance(self.hid, tuple) else self.hid.clone()
            return (self.y.clone(), _lib.attach_host(self.out_lens.clone(), _lib.host_lens(self.out_lens))), hid

    def __call__(self, x: torch.Tensor, lens: torch.Tensor, hx=None):
        _lib.require_gpu()
        lens_h = _lib.host_lens(lens)
        same = lens_h.numel() > 0 and bool((lens_h == lens_h[0]).all())
        if not same or self.graph_error is not None or _pipeline_running():
            return self.model((x, lens), hx)
        # what a captured graph froze besides the shapes: the parameters (pointer + version) and every module's call-site
        # switches that select kernels (DeepSpeech1.few_rows, RNN.inplace_state / check_status: ADVICE r5 -- flipping
        # few_rows after a capture kept replaying the old choice)
        sig = (tuple((p_.data_ptr(), _lib.version_of(p_)) for p_ in self.model.parameters()),
               tuple((n_, getattr(m_, a_)) for n_, m_ in self.model.named_modules() for a_ in ("few_rows", "inplace_state")
                     if hasattr(m_, a_)))
        if sig != self._sig:
            self._graphs.clear()
            self._sig = sig
        key = (tuple(x.shape), int(lens_h[0]), None if hx is None else isinstance(hx, tuple))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= self.max_graphs:
                return self.model((x, lens), hx)
            try:
                with torch.no_grad():
                    g = GraphedForward._Graph(self.model, x.shape, int(lens_h[0]), hx)
                self._graphs[key] = g
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager forward is the definition
                self.graph_error = f"{type(e).__name__}: {e}"[:300]
                torch.cuda.synchronize()
                return self.model((x, lens), hx)
        self.replays += 1
        return g.run(x if x.is_cuda else x.cuda(), hx)

    def check_status(self):
        """Raise if a persistent recurrence inside a replayed graph timed out (the eager call checks per forward)."""
        for g in self._graphs.values():
            for w_ in g.status_ws:
                if w_.buf is not None:
                    _lib.check(_lib.load().ms_rnn_status(_lib.ptr(w_.buf), _lib.stream_ptr()), "ms_rnn_layer_forward")


class ChunkedDeepSpeech2:
    def __init__(self, model, chunk_frames: int, carry_context: bool = False, use_graph: Optional[bool] = None):
        """``use_graph`` (default: on unless ``MS_STREAM_GRAPH=0``): replay steady-state slices / pushes as a captured HIP
        graph (both modes; any slice that is not steady state -- ragged lengths inside the slice, a short last slice, the
        held-back context filling up or being flushed -- runs eagerly)."""
        if chunk_frames <= 0:
            raise ValueError(f"chunk_frames={chunk_frames} must be > 0")
        self.model = model
        self.chunk_frames = chunk_frames
        self.carry_context = carry_context
        self._stream: Optional[_ContextStream] = None
        import os
        self.use_graph = (os.environ.get("MS_STREAM_GRAPH") != "0") if use_graph is None else bool(use_graph)
        self._graphs = {}
        self._ctx_graphs = {}
        self._graph_sig = None
        self.graph_error: Optional[str] = None
        self.graph_replays = 0

    # ------------------------------------------------------------------ carried context: explicit stream interface
    def begin(self, lens: torch.Tensor, total_frames: Optional[int] = None, hx=None) -> None:
        """Start a batch in carried-context mode: ``lens [N]`` (sorted in decreasing order) are the utterances' frame
        counts, ``total_frames`` the batch's padded length (default ``lens[0]``) -- the reference's SAME padding splits
        left / right by that length (cnn.py:148-163), so it has to be known up front to reproduce its output.  ``hx``: the
        recurrent stack's initial state, as ``DeepSpeech2.forward`` takes it."""
        _lib.require_gpu()
        lens_cpu = lens.detach().to("cpu", torch.int64)
        if lens_cpu.numel() > 1 and bool((lens_cpu[:-1] < lens_cpu[1:]).any()):
            raise RuntimeError("lengths must be sorted in decreasing order")
        total = int(lens_cpu[0]) if total_frames is None else int(total_frames)
        if self.use_graph:           # a graph holds the packed weights' addresses: a parameter that changed drops every graph
            psig = tuple((p_.data_ptr(), _lib.version_of(p_)) for p_ in self.model.parameters())
            if psig != self._graph_sig:
                self._graphs.clear()
                self._ctx_graphs.clear()
                self._graph_sig = psig
        self._stream = _ContextStream(self.model, int(lens_cpu.numel()), total, lens_cpu, hx)

    def push(self, chunk: Optional[torch.Tensor], final: bool = False) -> Optional[torch.Tensor]:
        """Feed the next input frames of every utterance of the batch (``[N, C, F, frames]``; utterances that have ended
        get whatever padding the caller has -- it is masked like cnn.py:425-443) and receive the logit rows that became
        computable, ``[rows, N, V]`` or None.  ``final=True`` flushes the held-back context (chunk may be None)."""
        if self._stream is None:
            raise RuntimeError("call begin(lens) first")
        st = self._stream
        with torch.no_grad():
            if self.use_graph and st.graph_ok and self.graph_error is None and not (_pipeline_running() and st.graph is None):
                y = self._push_graph(st, chunk, final)
                if y is not None:
                    return y
            return st.push(chunk, final)

    def _push_graph(self, st: "_ContextStream", chunk, final: bool):
        """Steady-state pushes replay a captured HIP graph (``_ContextGraph``); returns None when this push has to run
        eagerly.  A stream that has left its graph stays eager: what is left then is the clip's tail."""
        g = st.graph
        steady = chunk is not None and not final
        if g is None and steady:
            sig = st.steady_signature(chunk.shape)
            if sig is None:
                return None
            g = self._ctx_graphs.get(sig)        # (begin() dropped the graphs of an older parameter set)
            if g is not None and not g.can_replay(st, chunk):
                return None
            if g is None:
                stages = st.conv_stages()      # the capture must not record a clamped push: test before building
                if not _ContextGraph.fits(st, [s_.last[3] for s_ in stages], [s_.last[2] for s_ in stages], st.last_push[1]):
                    return None
                try:
                    g = _ContextGraph(st, chunk.shape)
                    self._ctx_graphs[sig] = g
                except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager push is the definition
                    self.graph_error = f"{type(e).__name__}: {e}"[:300]
                    st.graph_ok = False
                    torch.cuda.synchronize()
                    return None
            else:
                g.attach(st)
        if g is None:
            return None
        if steady and g.can_replay(st, chunk):
            self.graph_replays += 1
            return g.replay(st, chunk if chunk.is_cuda else chunk.cuda())
        # leaving the graph: its static tensors hold the current caches / state; the eager pushes go on from copies of them
        g.detach(st)
        if getattr(self.model.rnn, "check_status", False) and g.rnn_ws is not None and g.rnn_ws.buf is not None:
            _lib.check(_lib.load().ms_rnn_status(_lib.ptr(g.rnn_ws.buf), _lib.stream_ptr()), "ms_rnn_layer_forward")
        return None

    def latency_frames(self, total_frames: int = 1 << 20) -> int:
        """Algorithmic latency of the carried-context mode in input frames (see the module docstring)."""
        s = _ContextStream(self.model, 1, total_frames, torch.tensor([total_frames]))
        return s.latency_frames()

    def _call_with_context(self, x: torch.Tensor, lens: torch.Tensor, hx=None):
        n, t_total = x.shape[0], x.shape[-1]
        self.begin(lens, t_total, hx)
        st = self._stream
        x = x if x.is_cuda else x.cuda()
        outs = []
        t0 = 0
        while t0 < t_total:
            t1 = min(t0 + self.chunk_frames, t_total)
            y = self.push(x[..., t0:t1], final=(t1 == t_total))
            if y is not None:
                outs.append(y)
            t0 = t1
        logits = torch.cat(outs, 0)
        assert logits.shape[0] == st.total_out, (logits.shape, st.total_out)
        return (logits, st.out_lens.to(lens.dtype)), st.full_state

    # ------------------------------------------------------------------ the reference's plumbing, slice by slice
    def step(self, chunk: torch.Tensor, chunk_lens: torch.Tensor, state=None):
        """One slice: ``chunk [n_alive, C, F, <=chunk_frames]``; returns ``((y, out_lens), new_state)``."""
        return self.model((chunk, chunk_lens), state)

    def __call__(self, x: torch.Tensor, lens: torch.Tensor, hx=None):
        """Whole (padded, length-sorted) batch, processed slice by slice.  Returns
        ``((logits[T_out, N, V], out_lens[N]), (h_n, c_n) | h_n)`` with T_out the sum of the
        slices' output frames and the state of every utterance at its own last slice."""
        _lib.require_gpu()
        if self.carry_context:
            return self._call_with_context(x, lens, hx)
        n, t_total = x.shape[0], x.shape[-1]
        lens_cpu = lens.detach().to("cpu", torch.int64)
        if n > 1 and bool((lens_cpu[:-1] < lens_cpu[1:]).any()):
            raise RuntimeError("lengths must be sorted in decreasing order")
        x = x if x.is_cuda else x.cuda()
        outs, out_lens = [], torch.zeros(n, dtype=torch.int64)
        state = hx
        full_state = None
        active = None            # the _ChunkGraph whose static tensors hold the current state, if any
        if self.use_graph:       # a graph holds the packed weights' addresses: a parameter that changed invalidates every graph
            sig = tuple((p_.data_ptr(), _lib.version_of(p_)) for p_ in self.model.parameters())
            if sig != self._graph_sig:
                self._graphs.clear()
                self._graph_sig = sig
        used = set()
        check_was = getattr(self.model.rnn, "check_status", None)
        t0 = 0
        while t0 < t_total:
            alive = int((lens_cpu > t0).sum())
            if alive == 0:
                break
            xc = x[:alive, :, :, t0:t0 + self.chunk_frames]
            lc = (lens_cpu[:alive] - t0).clamp(max=xc.shape[-1])
            steady = (self.use_graph and self.graph_error is None and state is not None and xc.shape[-1] == self.chunk_frames
                      and int(lc.min()) == self.chunk_frames and not _pipeline_running())
            y = ol_host = None
            cg_used = False
            if steady:
                key = (alive, tuple(xc.shape[1:]))
                cg = self._graphs.get(key)
                if cg is None:
                    try:
                        if check_was is not None:
                            self.model.rnn.check_status = False      # ms_rnn_status synchronises: not inside a capture
                        cur = active.state if active is not None else state
                        cg = _ChunkGraph(self.model, alive, xc.shape, cur)
                        self._graphs[key] = cg
                    except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager path below is the definition
                        self.graph_error = f"{type(e).__name__}: {e}"[:300]
                        cg = None
                        torch.cuda.synchronize()
                    finally:
                        if check_was is not None:
                            self.model.rnn.check_status = check_was
                if cg is not None:
                    if active is not cg:
                        cg.load_state(active.state if active is not None else state, alive)
                        active = cg
                    y, ol_host = cg.run(xc), cg.out_lens_host
                    state = cg.state
                    cg_used = True
                    used.add(cg)
                    self.graph_replays += 1
            if y is None:
                if active is not None:       # leave the graph: its static state is the current state
                    state, active = active.state, None
                hx_c = None
                if state is not None:
                    hx_c = tuple(s_[:, :alive].contiguous() for s_ in state) if isinstance(state, tuple) else \
                        state[:, :alive].contiguous()
                (y, ol), state = self.step(xc.contiguous(), lc, hx_c)
                ol_host = _lib.host_lens(ol)   # host values ride along with the device tensor: no read-back
            # the returned state holds every utterance's state at ITS last slice: rows are copied out of the running state only
            # when utterances are about to leave the batch (and once at the end), not per slice
            t_next = t0 + self.chunk_frames
            alive_next = int((lens_cpu > t_next).sum()) if t_next < t_total else 0
            if alive_next < alive:
                if full_state is None:
                    full_state = tuple(torch.zeros(s_.shape[0], n, s_.shape[2], device=s_.device) for s_ in state) \
                        if isinstance(state, tuple) else torch.zeros(state.shape[0], n, state.shape[2], device=state.device)
                if isinstance(state, tuple):
                    for fs, s_ in zip(full_state, state):
                        fs[:, alive_next:alive] = s_[:, alive_next:alive]
                else:
                    full_state[:, alive_next:alive] = state[:, alive_next:alive]
            if alive == n:
                block = y.clone() if cg_used else y
            else:
                block = torch.zeros((y.shape[0], n, y.shape[2]), dtype=y.dtype, device=y.device)
                block[:, :alive] = y
            outs.append(block)
            out_lens[:alive] += ol_host
            t0 = t_next
        if check_was:            # the graphs' recurrent launches report through the graphs' own workspaces
            for cg in used:
                if cg.rnn_ws is not None and cg.rnn_ws.buf is not None:
                    _lib.check(_lib.load().ms_rnn_status(_lib.ptr(cg.rnn_ws.buf), _lib.stream_ptr()), "ms_rnn_layer_forward")
        return (torch.cat(outs, 0), out_lens.to(lens.dtype)), full_state
