"""Two batches in flight on one MI355X (new design; the reference runs one batch at a time on one stream).

Why: the persistent LSTM recurrence is a latency chain -- its CUs idle about half the time waiting for the hand-over of
``h`` -- while the input-projection GEMM of the next layer is MFMA-bound, and within ONE batch the two cannot overlap
(layer k+1's projection needs both directions of layer k at every frame).  With a SECOND batch in flight, batch B's
projection runs in the shadow of batch A's recurrence and vice versa:

* two HIP streams, one per batch; the host issues the two batches LAYER BY LAYER in turn (A1 B1 A2 B2 ...), because the
  library chains persistent launches across streams in host issue order (two recurrences must never be resident together:
  each fills every CU with workgroups that wait for their own peers, ``csrc/rnn.hip`` PersistentTurn);
* what runs beside a recurrence depends on its kernel (``_gemm_variant``): the wide-workgroup LSTM kernel (H = 1024, round 3)
  occupies half of the CUs per batch, so the other batch's projection is the regular 8-wave GEMM on the free half (11.5 ms per
  batch on the config-2 network); the 8-unit persistent kernels fill every CU with 304 registers per SIMD and get the 4-wave
  256 x 128 form of the LDS-DMA GEMM (176 VGPRs, 128 KB of LDS, ``ms_gemm_set_variant(COTENANT_GEMM_VARIANT)``), whose
  workgroups fit on a CU BESIDE a workgroup of the recurrence; the 8-wave form cannot share a CU with those and would only
  time-slice (``tools/overlap_probe.py``);
* results are bit-identical to the one-batch path (same kernels' arithmetic; ``tests/test_gpu_pipeline.py``).

Measured and dropped: more than two batches in flight (depth 3: 14.2, depth 4: 15.1 ms per batch against 13.7 for depth 2), and
issuing the NEXT batch's convolutions early on a third stream so that they leave the critical slot between a batch's output
layers and the next one's first projection (14.5 ms): a third resident kernel takes LDS and CU share from the co-tenant GEMM,
which the recurrences then wait for; issuing a batch's output layers late (behind the first recurrent layer of the slot's next
batch, so that the next batch's convolutions and first projection come first: 14.08 against 13.74 .. 13.95 ms).

Measured (``profiles/r02*_overlap*``): the recurrence slows by ~45 % under its co-tenant (its exchange shares the CU's
memory queue with the GEMM's operand stream), the GEMM hides completely, net ~12 % more batches per second on the recurrent
stack; per-batch latency is that of two batches.  A throughput mode, therefore: a caller that needs the lowest latency
per batch keeps calling the model directly.

The alternation is done with two host threads that pass a baton at the issue points of ``_lib.at_issue_point`` (after the
convolutions, after every recurrent layer): only one thread runs at a time, so nothing in the library or the modules has to
be thread-safe beyond that.
"""
import copy
import threading
from typing import Callable, List, Optional, Sequence

import torch

from myrtlespeech_amd import _lib

# ms_gemm_set_variant value of the co-tenant projection GEMM: the 4-wave 256 x 128 LDS-DMA kernel with a wave's 12 DMA pieces of
# the next K-block issued ONE in front of each MFMA group of the block's first four steps (variant 10).  Round 2 issued them
# three at a time at the head of those steps (variant 7): same-process A/B on the bench network, 40 batches x 4 rounds
# (tools/cotenant_variants.py, profiles/r03g_*): 14.42 -> 13.52 ms per batch; the GEMM's own time beside the recurrence
# 2.79 -> 2.46 ms, the recurrence's 2.68 -> 2.58 ms.  Bit-identical outputs (tests/cfg_checks.py::gemm_variants_equal).
COTENANT_GEMM_VARIANT = 10


class _Baton:
    """Round-robin alternation between the worker threads; a thread that has finished is skipped from then on."""

    def __init__(self, n: int):
        self.cv = threading.Condition()
        self.n = n
        self.turn = 0
        self.active = [True] * n

    def _next_active(self, after: int) -> int:
        for d in range(1, self.n + 1):
            k = (after + d) % self.n
            if self.active[k]:
                return k
        return after

    def wait_turn(self, me: int):
        with self.cv:
            while self.turn != me:
                self.cv.wait()

    def pass_on(self, me: int):
        with self.cv:
            self.turn = self._next_active(me)
            self.cv.notify_all()
        self.wait_turn(me)

    def leave(self, me: int):
        with self.cv:
            self.active[me] = False
            if self.turn == me:
                self.turn = self._next_active(me)
            self.cv.notify_all()


def replica(model: torch.nn.Module) -> torch.nn.Module:
    """A copy of ``model`` for another stream that SHARES its Parameters and buffers (the same objects: a later
    ``load_state_dict`` / ``checkpoint.load`` / in-place edit of the caller's model is seen by every replica, whose
    packed-weight caches are keyed on the parameters' version counters and addresses and repack by themselves) and owns
    everything else: workspaces, packed-weight caches, convolution scratch.  The reference's contract is that checkpoints
    load into ``stt.model`` (scripts/export_ds1_onnx.py:49-50) -- there is one set of weights."""
    memo = {id(t): t for t in list(model.parameters()) + list(model.buffers())}
    return copy.deepcopy(model, memo)


class BatchesInFlight:
    """``BatchesInFlight(model, depth=2)(batches)``: run ``model`` over a list of ``(x, lens)`` batches, ``depth`` at a time
    (``TwoBatchesInFlight`` = depth 2).  With depth 3 there is always a batch whose projection has finished when a
    recurrence ends, so the recurrences follow each other without a gap (with depth 2 the slot that carries a batch's
    output layers, the next batch's convolutions and its first projection is longer than the recurrence it hides under).

    ``model`` is any module of this package whose forward enqueues work on the current stream (normally ``DeepSpeech2``); a
    ``replica`` serves the second stream (its own workspaces and packed-weight caches, the SAME Parameter objects: weights
    loaded or edited after the pipe was built reach both streams).  ``post`` (optional) is called on each batch's output on that batch's stream right after
    its forward was issued -- e.g. ``decoder.launch`` of ``CTCGreedyDecoder``, whose ``.result()`` the caller collects
    afterwards -- so that no host read-back interrupts the alternation.  Results come back in the order of ``batches``."""

    def __init__(self, model: torch.nn.Module, post: Optional[Callable] = None, pre: Optional[Callable] = None, depth: int = 2):
        _lib.require_gpu()
        if depth < 2 or depth > 4:
            raise ValueError(f"depth={depth} must be in [2, 4]")
        self.depth = depth
        self.models = tuple([model] + [replica(model) for _ in range(depth - 1)])
        self.streams = tuple(torch.cuda.Stream() for _ in range(depth))
        self.post = post
        self.pre = pre      # called with the batch index on the batch's stream before its forward is issued (e.g. to record an event)
        # the recurrent stacks (their workspaces carry ms_rnn_status's sticky word) -- by TYPE: a wrapped model that contains its
        # loss would otherwise hand the CTC workspace to ms_rnn_status (CTCLoss has check_status and _workspace too)
        from myrtlespeech_amd.model.hard_lstm import HardLSTM
        from myrtlespeech_amd.model.rnn import RNN
        self._stacks = [m for mod in self.models for m in mod.modules() if isinstance(m, (RNN, HardLSTM))]
        self._n_caller_stacks = len(self._stacks) // depth

    def __call__(self, batches: Sequence) -> List:
        lib = _lib.load()
        results: List = [None] * len(batches)
        errors: List = []
        baton = _Baton(self.depth)
        slot_of = {}
        device = torch.cuda.current_device()
        caller = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(caller)
        inference = torch.is_inference_mode_enabled()

        def hook():
            me = slot_of.get(threading.get_ident())
            if me is not None:
                baton.pass_on(me)

        def worker(me: int):
            slot_of[threading.get_ident()] = me
            try:
                torch.cuda.set_device(device)                   # the current device is a per-thread setting
                baton.wait_turn(me)
                # grad / inference mode are per-thread settings: the workers take the caller's inference mode
                with torch.cuda.stream(self.streams[me]), torch.no_grad(), torch.inference_mode(inference):
                    self.streams[me].wait_event(ready)          # inputs made on the caller's stream
                    for k in range(me, len(batches), self.depth):
                        if self.pre is not None:
                            self.pre(k)
                        out = self.models[me](batches[k])
                        results[k] = out if self.post is None else self.post(out)
                        baton.pass_on(me)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
                errors.append(e)
            finally:
                baton.leave(me)

        prev_hook = _lib.issue_point
        _lib.issue_point = hook
        lib.ms_gemm_set_variant(self._gemm_variant(lib, batches))
        # a per-call status check of a recurrent stack synchronises its stream in the middle of the alternation: switched
        # off while the pipe runs (and put back: models[0] is the CALLER's model), the sticky time-out word is read once
        # at the end of this call instead
        prev_check = [m.check_status for m in self._stacks]
        for m in self._stacks:
            m.check_status = False
        try:
            threads = [threading.Thread(target=worker, args=(m,), daemon=True) for m in range(self.depth)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            for m, c in zip(self._stacks, prev_check):
                m.check_status = c
            lib.ms_gemm_set_variant(0)
            _lib.issue_point = prev_hook
        for s in self.streams:                                  # later work on the caller's stream sees the results
            done = torch.cuda.Event()
            done.record(s)
            caller.wait_event(done)
        if errors:
            raise errors[0]

        def hand_over(obj):
            """Result tensors were allocated on a worker stream and will be used (and freed) on the caller's: tell the
            caching allocator, or a freed block could be re-used by the worker stream under the caller's kernels."""
            if isinstance(obj, torch.Tensor):
                if obj.is_cuda:
                    obj.record_stream(caller)
            elif isinstance(obj, (tuple, list)):
                for o in obj:
                    hand_over(o)

        hand_over(results)
        if any(prev_check[:self._n_caller_stacks]):       # self._stacks lists the caller's model first
            # one host sync per call, after the streams have been joined to the caller's: a persistent launch that gave up
            # (MS_ERR_TIMEOUT, outputs are garbage) must not pass silently.  A caller that switched the per-call check of
            # its model off (``model.rnn.check_status = False``: it checks by itself, like bench.py's timed region) is not
            # synchronised here either.
            self.check_status()
        return results

    def _gemm_variant(self, lib, batches) -> int:
        """Which projection GEMM runs beside the other batch's recurrence.  Recurrent stacks that take the wide-workgroup
        kernel (H = 1024 LSTM, split-bf16 operands: ``ms_rnn_layer_is_wide``) occupy HALF of the CUs per batch, so the regular
        8-wave GEMM simply runs on the other half and spreads over all of them when the recurrence ends (round 3, measured
        11.65 ms per batch against 12.86 with the co-tenant form and 13.4 with the 8-unit recurrence + co-tenant,
        ``profiles/r03ad_*``); every other persistent recurrence fills all CUs with 304 registers per SIMD and gets the
        4-wave co-tenant form that fits beside it."""
        if not batches:
            return 0
        from myrtlespeech_amd.model.hard_lstm import HardLSTM
        from myrtlespeech_amd.model.rnn import _CELL, RNN
        n = int(batches[0][1].numel())
        wide = []
        for m in self._stacks[:self._n_caller_stacks]:
            if isinstance(m, RNN):
                cell, hidden = _CELL[m.rnn_type], m.rnn.hidden_size
            elif isinstance(m, HardLSTM):
                cell, hidden = _lib.CELL_HARD_LSTM, m.hidden_size
            else:
                return COTENANT_GEMM_VARIANT
            wide.append(bool(lib.ms_rnn_layer_is_wide(int(cell), int(hidden), 2 if m.bidirectional else 1, n)))
        return 0 if wide and all(wide) else COTENANT_GEMM_VARIANT

    def check_status(self):
        """Raise if a persistent recurrent launch of either replica timed out since the last check (synchronises)."""
        lib = _lib.load()
        for m in self._stacks:
            if m._workspace.buf is not None:
                _lib.check(lib.ms_rnn_status(_lib.ptr(m._workspace.buf), _lib.stream_ptr()), "persistent recurrent layer")


class TwoBatchesInFlight(BatchesInFlight):
    def __init__(self, model: torch.nn.Module, post: Optional[Callable] = None, pre: Optional[Callable] = None):
        super().__init__(model, post=post, pre=pre, depth=2)


class PairedBatches:
    """``PairedBatches(model)(batches)``: the other throughput mode (round 3) -- consecutive batches go through ``model`` two at
    a time as ONE batch.  For the config-2 network (5 x BiLSTM-1024, batches of 32) the library then runs the two batches'
    recurrences side by side in one launch of the wide-workgroup kernel (``csrc/rnn.hip``, ``lstm_persistent_wide2_kernel``:
    16 hidden units per workgroup, so the 64 KB of ``h`` a workgroup pulls out of L2 every step feed twice the arithmetic and
    a batch needs half the chip): 1.96 ms per layer for two batches against 1.70 ms for one, and every other kernel of the
    step (convolutions, projection GEMMs, output layers, greedy decode) runs once on 64 utterances.  No threads, one stream.

    What a caller gets per batch is what ``model`` returns for the merged batch, cut back to the batch's utterances in
    their own order: ``((logits[T', N_b, V], lens_b), hid_b)``.  Utterances do not interact in any module and no
    KERNEL's arithmetic for an utterance depends on what it is batched with -- but which kernel a layer runs on can depend
    on the merged batch's row count (``csrc/gemm_split.hip``: the GEMM's tile variant follows ``M = T * N``; ``csrc/rnn.hip``:
    the recurrent kernel follows ``N``).  So the guarantee is: **bit-identical to the one-batch result wherever both sizes
    select the same kernels** -- the config-2 network at batches of 32 does, ``tests/test_gpu_pipeline.py`` asserts
    ``torch.equal`` there at full size -- **and within float32 rounding of it (a different order of the same sums, same
    transcripts) otherwise**, e.g. the small network of ``__graft_entry__.smoke()``, which checks ``< 1e-5``.  Per-batch
    latency is that of the PAIR's forward (about twice the one-batch figure); ``bench.py`` states which mode its ``value``
    came from and that latency (``config.pipeline``, ``latency_ms_per_batch``).

    Pairs are merged in decreasing length order (``enforce_sorted``, rnn.py:174) and need the same frame count and at most
    64 utterances together; a batch that cannot be paired (odd one out, different frame counts) runs alone.  Like the
    reference's ``MaskConv*`` the convolutions zero the input past each length: the callers' tensors receive that masking
    back (a copy, only when some length is shorter than the frame count).  ``post`` / ``pre`` as in ``BatchesInFlight``."""

    def __init__(self, model: torch.nn.Module, post: Optional[Callable] = None, pre: Optional[Callable] = None, max_rows: int = 64):
        _lib.require_gpu()
        self.model, self.post, self.pre, self.max_rows = model, post, pre, max_rows

    def _finish(self, out):
        return out if self.post is None else self.post(out)

    def __call__(self, batches: Sequence) -> List:
        results: List = [None] * len(batches)
        k = 0
        with torch.no_grad():
            while k < len(batches):
                (xa, la) = batches[k]
                pair = None
                if k + 1 < len(batches):
                    (xb, lb) = batches[k + 1]
                    if xa.shape[1:] == xb.shape[1:] and xa.shape[0] + xb.shape[0] <= self.max_rows and xa.dtype == xb.dtype:
                        pair = (xb, lb)
                if pair is None:
                    if self.pre is not None:
                        self.pre(k)
                    results[k] = self._finish(self.model((xa, la)))
                    k += 1
                    continue
                xb, lb = pair
                ha, hb = _lib.host_lens(la), _lib.host_lens(lb)
                na = int(ha.numel())
                both = torch.cat([ha, hb])
                order = torch.sort(both, descending=True, stable=True).indices          # merged position -> source row
                where = torch.empty_like(order)
                where[order] = torch.arange(order.numel())                               # source row -> merged position
                dev = xa.device if xa.is_cuda else torch.device("cuda")
                order_d = order.to(dev)
                x = torch.cat([xa.to(dev), xb.to(dev)]).index_select(0, order_d)
                lens = both[order].to(la.dtype)
                if self.pre is not None:
                    self.pre(k)
                    self.pre(k + 1)
                (y, out_lens), hid = self.model((x, lens))
                if bool((both < xa.shape[-1]).any()):                                    # the reference masks its input in place
                    back = x.index_select(0, where.to(dev))
                    if xa.is_cuda:
                        xa.copy_(back[:na])
                    if xb.is_cuda:
                        xb.copy_(back[na:])
                out_host = _lib.host_lens(out_lens)
                for b, (lo, hi) in enumerate(((0, na), (na, both.numel()))):
                    pos = where[lo:hi]
                    pos_d = pos.to(dev)
                    lens_b = out_host[pos].to(out_lens.dtype)
                    lens_b = _lib.attach_host(_lib.upload(lens_b), lens_b) if out_lens.is_cuda else lens_b
                    if isinstance(hid, tuple):
                        hid_b = tuple(h.index_select(1, pos_d) for h in hid)
                    else:
                        hid_b = None if hid is None else hid.index_select(1, pos_d)
                    results[k + b] = self._finish(((y.index_select(1, pos_d), lens_b), hid_b))
                k += 2
        return results
