"""Two batches in flight (myrtlespeech_amd/pipeline.py): bit-identical to running the batches one after the other.
Needs a real MI355X: -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_ds2(H):
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
    torch.manual_seed(H)

    def act():
        return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
    cnn = torch.nn.Sequential(MaskConv2d(1, 8, [11, 5], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(8, 8, [5, 5], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.LSTM, 8 * 10, H, num_layers=3, bidirectional=True, forget_gate_bias=1.0)
    fc = FullyConnected(2 * H, 29, 1, 96, torch.nn.Hardtanh(0.0, 20.0))
    return DeepSpeech2(cnn, rnn, None, fc).eval()


def _batches(n, N, T_, F, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.randn(N, 1, F, T_, generator=g).cuda()
        lens = torch.sort(torch.randint(T_ // 2, T_ + 1, (N,), generator=g), descending=True).values
        lens[0] = T_
        out.append((x, lens))
    return out


@pytest.mark.parametrize("H,n_batches", [(256, 5), (128, 2), (64, 3)])   # persistent two-stream, persistent one-stream, step kernels
def test_two_batches_in_flight_equal_the_sequential_run(H, n_batches):
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model = _small_ds2(H)
    dec = CTCGreedyDecoder(28)
    batches = _batches(n_batches, 12, 90, 40, H)
    want = []
    for x, lens in batches:
        (y, ol), (hn, cn) = model((x.clone(), lens))
        want.append((y, ol, hn, cn, dec(y, ol)))
    pipe = TwoBatchesInFlight(model, post=lambda out: (out, dec.launch(out[0][0], out[0][1])))
    got = pipe([(x.clone(), lens) for x, lens in batches])
    pipe.check_status()
    assert len(got) == n_batches
    for (((y, ol), (hn, cn)), pending), (wy, wol, whn, wcn, wdec) in zip(got, want):
        assert torch.equal(y, wy) and torch.equal(ol.cpu(), wol.cpu())
        assert torch.equal(hn, whn) and torch.equal(cn, wcn)
        assert pending.result() == wdec
    # the one-batch path is back to its own kernels afterwards
    (y, _), _ = model((batches[0][0].clone(), batches[0][1]))
    assert torch.equal(y, want[0][0])


def test_three_batches_in_flight_equal_the_sequential_run():
    """The depth parameter (measured slower than 2 on config 2; kept for other shapes): same outputs."""
    from myrtlespeech_amd.pipeline import BatchesInFlight
    model = _small_ds2(256)
    batches = _batches(7, 8, 70, 40, 3)
    want = [model((x.clone(), lens))[0][0] for x, lens in batches]
    pipe = BatchesInFlight(model, depth=3)
    got = pipe([(x.clone(), lens) for x, lens in batches])
    pipe.check_status()
    for ((y, _), _), w in zip(got, want):
        assert torch.equal(y, w)
    with pytest.raises(ValueError):
        BatchesInFlight(model, depth=1)


def test_two_batches_in_flight_full_size_network():
    """The config-2 network (5 x BiLSTM-1024, batch 32 x 1001 frames): the co-tenant projection GEMM (4 waves, 256 x 128
    tiles) and the cross-stream chain of the persistent launches at the size they were built for."""
    import bench
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    model = bench.build_model()
    batches = _batches(4, 32, 1001, 80, 7)
    want = [model((x.clone(), lens))[0][0] for x, lens in batches]
    pipe = TwoBatchesInFlight(model)
    got = pipe([(x.clone(), lens) for x, lens in batches])
    pipe.check_status()
    for ((y, _), _), w in zip(got, want):
        assert torch.equal(y, w)


def test_pipeline_propagates_errors_and_restores_the_hooks():
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    model = _small_ds2(64)
    pipe = TwoBatchesInFlight(model)
    good = _batches(1, 4, 60, 40, 1)[0]
    bad = (good[0], torch.tensor([10, 20, 30, 40]))          # not sorted in decreasing order
    with pytest.raises(RuntimeError):
        pipe([good, bad, good])
    assert _lib.issue_point is None
    assert len(pipe([good, good, good])) == 3


def test_split_gemm_scratch_is_per_stream():
    """The operand planes of a large Linear are made in a scratch buffer: one per HIP stream, or the two streams of the
    pipeline would overwrite each other's planes under a running GEMM (found by review in round 2; the alternation happened
    to keep the two output stacks apart in time)."""
    from myrtlespeech_amd.model import fully_connected as F
    lin = torch.nn.Linear(2048, 1024).cuda()
    x = torch.randn(16032, 2048, device="cuda")
    want = F.run_linear_stack(x, [(lin, None)])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(F.run_linear_stack(x, [(lin, None)]))
    torch.cuda.synchronize()
    keys = {s1.cuda_stream, s2.cuda_stream}
    assert keys <= set(F._split_ws) and F._split_ws[s1.cuda_stream].buf.data_ptr() != F._split_ws[s2.cuda_stream].buf.data_ptr()
    assert all(torch.equal(o, want) for o in outs)


def test_weights_loaded_after_the_pipe_was_built_reach_both_streams(tmp_path):
    """The replicas share the caller's Parameters: a checkpoint loaded into ``model`` AFTER the pipe was
    constructed (the reference loads into ``stt.model``, scripts/export_ds1_onnx.py:49-50), and an in-place edit after
    that, are seen by every stream -- every batch equals the sequential run on the new weights."""
    from myrtlespeech_amd import checkpoint
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    model = _small_ds2(256)
    batches = _batches(4, 12, 90, 40, 11)
    pipe = TwoBatchesInFlight(model)
    before = [o[0][0].clone() for o in pipe([(x.clone(), lens) for x, lens in batches])]
    assert all(p is q for p, q in zip(pipe.models[0].parameters(), pipe.models[1].parameters()))
    torch.manual_seed(99)
    other = _small_ds2(256)
    with torch.no_grad():
        for p in other.parameters():
            p.mul_(1.25)
    path = tmp_path / "state_dict_1.pt"
    torch.save({k: v.detach().cpu() for k, v in other.state_dict().items()}, str(path))
    checkpoint.load(model, path)
    want = [model((x.clone(), lens))[0][0] for x, lens in batches]
    assert not torch.equal(want[0], before[0])
    got = pipe([(x.clone(), lens) for x, lens in batches])
    for ((y, _), _), w in zip(got, want):
        assert torch.equal(y, w)
    with torch.no_grad():                      # in-place edit: version counters move, both packed caches rebuild
        model.rnn.rnn.weight_hh_l1.mul_(0.5)
        model.fully_connected.fully_connected[0].bias.add_(0.125)
    want2 = [model((x.clone(), lens))[0][0] for x, lens in batches]
    assert not torch.equal(want2[0], want[0])
    got2 = pipe([(x.clone(), lens) for x, lens in batches])
    for ((y, _), _), w in zip(got2, want2):
        assert torch.equal(y, w)


def test_pipe_checks_the_status_word_and_restores_the_models_own_setting():
    """the per-call status check is switched off only WHILE the pipe runs and the sticky time-out word is read
    at the end of every call (one sync)."""
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    model = _small_ds2(256)
    assert model.rnn.check_status is True
    pipe = TwoBatchesInFlight(model)
    assert model.rnn.check_status is True and pipe.models[1].rnn.check_status is True
    calls = []
    orig = pipe.check_status
    pipe.check_status = lambda: (calls.append(1), orig())[1]
    pipe(_batches(3, 8, 70, 40, 5))
    assert calls == [1] and model.rnn.check_status is True
    model.rnn.check_status = False             # a caller that checks by itself (bench.py) is not synchronised by the pipe
    pipe(_batches(2, 8, 70, 40, 6))
    assert calls == [1] and model.rnn.check_status is False


def test_split_gemm_scratch_is_pruned_when_streams_come_and_go():
    """VERDICT r2 small close: the per-stream scratch of the split GEMM is a small LRU, not one buffer per stream handle for
    the life of the process."""
    from myrtlespeech_amd.model import fully_connected as F
    lin = torch.nn.Linear(2048, 1024).cuda()
    x = torch.randn(4096, 2048, device="cuda")
    want = F.run_linear_stack(x, [(lin, None)])
    torch.cuda.synchronize()
    outs = []
    for _ in range(12):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            outs.append(F.run_linear_stack(x, [(lin, None)]))
    torch.cuda.synchronize()
    assert len(F._split_ws) <= F._SPLIT_WS_STREAMS
    assert all(torch.equal(o, want) for o in outs)


@pytest.mark.parametrize("n_out", [29, 1024])
@pytest.mark.parametrize("few_rows", [False, True])
def test_exact_f32_linear_rows_do_not_depend_on_the_row_count(n_out, few_rows):
    """the K-slice count of the exact-f32 layers was a function of M (slices for M <= 256 only), so the same rows
    came out with other roundings in a 200-row call than inside a 300-row call -- a one-utterance shard or a short push
    against the whole batch.  It now depends on (K, N, caller's flag) only: straddle the old boundary and compare bits."""
    from myrtlespeech_amd.model import fully_connected as F
    torch.manual_seed(5)
    lin = torch.nn.Linear(1024, n_out).cuda()
    x = torch.randn(700, 1024, device="cuda")
    whole = F.run_linear_stack(x, [(lin, (0.0, 20.0))], few_rows=few_rows)
    for m in (1, 37, 200, 256, 257, 300, 513):
        part = F.run_linear_stack(x[:m].contiguous(), [(lin, (0.0, 20.0))], few_rows=few_rows)
        assert torch.equal(part, whole[:m]), (m, n_out, few_rows)
    want = torch.clamp(x.double() @ lin.weight.double().t() + lin.bias.double(), 0.0, 20.0)
    assert float((whole.double() - want).abs().max()) < 2e-4


def test_paired_batches_equal_the_merged_forward_and_match_the_one_batch_path():
    """``PairedBatches`` (round 3): two batches per forward.  Bit-identical to ``model`` on the merged, length-sorted batch (it
    is that call); equal to the one-batch path within float32 rounding, greedy transcripts equal; ragged lengths (the merge
    permutes utterances), an odd batch count, a batch with another frame count (runs alone), the in-place masking of the
    callers' inputs."""
    from myrtlespeech_amd.pipeline import PairedBatches
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model = _small_ds2(256)
    dec = CTCGreedyDecoder(28)
    batches = _batches(5, 12, 90, 40, 31)
    other = _batches(1, 6, 70, 40, 32)[0]                   # 70 frames: cannot be paired with a 90-frame batch
    batches.insert(2, other)
    want = []
    for x, lens in batches:
        xc = x.clone()
        (y, ol), (hn, cn) = model((xc, lens))
        want.append((y, ol, hn, cn, dec(y, ol), xc))
    ins = [(x.clone(), lens) for x, lens in batches]
    pipe = PairedBatches(model, post=lambda out: (out, dec.launch(out[0][0], out[0][1])))
    got = pipe(ins)
    assert len(got) == len(batches)
    for k, ((((y, ol), (hn, cn)), pending), (wy, wol, whn, wcn, wdec, wx)) in enumerate(zip(got, want)):
        assert y.shape == wy.shape and torch.equal(ol.cpu(), wol.cpu()), k
        # utterances go through the same arithmetic whatever they are batched with: the same bits as the one-batch path
        assert torch.equal(y, wy) and torch.equal(hn, whn) and torch.equal(cn, wcn), k
        assert pending.result() == wdec, k
        assert torch.equal(ins[k][0], wx), f"batch {k}: the caller's input was not masked like the reference masks it"
    # batches 2 (other frame count) ran alone, 0+1 and 3+4 were paired, 5 is the odd one out: those equal the one-batch bits
    for k in (2, 5):
        assert torch.equal(got[k][0][0][0], want[k][0])
    # a pair IS the merged forward
    (xa, la), (xb, lb) = batches[0], batches[1]
    both = torch.cat([la, lb])
    order = torch.sort(both, descending=True, stable=True).indices
    (ym, olm), _ = model((torch.cat([xa, xb]).index_select(0, order.cuda()).clone(), both[order]))
    where = torch.empty_like(order)
    where[order] = torch.arange(order.numel())
    assert torch.equal(got[0][0][0][0], ym.index_select(1, where[:12].cuda()))
    assert torch.equal(got[1][0][0][0], ym.index_select(1, where[12:].cuda()))


def test_paired_batches_full_size_network_uses_the_wide_kernel_and_matches_the_reference():
    """configs[1] twice per forward: every batch against the reference's own outputs (golden summary) within the
    north-star's 1e-3, transcripts bit-exact -- through ``lstm_persistent_wide2_kernel`` (two 32-row groups per launch)."""
    import cfg_checks
    err = cfg_checks.paired_full(atol=1e-3)
    assert err < 1e-5


# ----------------------------------------------------------------------------- the overlapped stack schedule (round 6)
@pytest.mark.parametrize("kind,N,T_,In,nl,segs,with_hx,ragged", [
    ("LSTM", 32, 501, 640, 5, 8, False, False),      # the config-2 recurrent stack
    ("LSTM", 32, 257, 96, 2, 4, True, False),        # initial state given, two layers, a last segment of one step
    ("LSTM", 5, 300, 64, 3, 8, False, True),         # few sequences (256 x 128 GEMM tiles), lengths that differ
    ("HardLSTM", 17, 200, 32, 2, 3, False, False),
    ("LSTM", 32, 130, 640, 3, 16, True, True)])
def test_overlapped_stack_gives_the_layer_by_layer_bits(kind, N, T_, In, nl, segs, with_hx, ragged):
    """``ms_rnn_stack_forward``: layer l's recurrence as launches over time segments, layer l+1's projection
    beside it on a second stream as two K-half launches that share their accumulators through memory -- against the layer loop
    (``MS_RNN_OVERLAP=0``'s path) on the same inputs: outputs and final states ``torch.equal``, whatever the segmentation."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model import rnn as R
    cell = _lib.CELL_LSTM if kind == "LSTM" else _lib.CELL_HARD_LSTM
    H, ndir = 1024, 2
    torch.manual_seed(N * 1000 + T_)
    m = R.RNN(R.RNNType.LSTM, In, H, num_layers=nl, bidirectional=True, forget_gate_bias=1.0).eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():      # saturate some gates: a rounding difference would not stay hidden
            if "weight_ih" in k:
                v.mul_(8.0)
    x = torch.randn(T_, N, In, device="cuda")
    lens = torch.full((N,), T_, dtype=torch.int64)
    if ragged:
        lens = torch.sort(torch.randint(T_ // 3, T_ + 1, (N,)), descending=True).values
        lens[0] = T_
    h0 = c0 = None
    if with_hx:
        h0 = (torch.randn(nl * ndir, N, H, device="cuda") * 0.5).contiguous()
        c0 = (torch.randn(nl * ndir, N, H, device="cuda") * 0.5).contiguous()
    lib = _lib.load()
    assert lib.ms_rnn_stack_overlap_ok(cell, T_, N, In, H, ndir, nl) == 1

    def run(overlap):
        prev = (R._OVERLAP, R._OVERLAP_SEGMENTS)
        R._OVERLAP, R._OVERLAP_SEGMENTS = overlap, segs
        try:
            # (ragged=False: every row is computed, the lengths act through the recurrence's per-frame predicate -- the path a
            # batch takes when its rows are not packed; the packed-rows path never overlaps)
            return R.run_layers(cell, x, _lib.lens_i32(lens), T_, m._layer_params(), [R.PackedLayer() for _ in range(nl)], H,
                                h0, c0, _lib.Workspace(), ragged=False)
        finally:
            R._OVERLAP, R._OVERLAP_SEGMENTS = prev
    want, got = run(False), run(True)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert float(want[0].abs().max()) > 0.5          # (a live network: outputs are not all tiny)
    again = run(True)                                 # a second call re-uses the library's side stream and events
    assert all(torch.equal(a, b) for a, b in zip(want, again))


@pytest.mark.parametrize("N,T_,In,nl,with_hx,ragged", [(64, 16, 640, 5, True, False), (48, 16, 96, 2, False, False),
                                                       (33, 40, 64, 3, True, True), (64, 7, 32, 1, False, True)])
def test_half_batch_pipeline_gives_the_one_batch_bits(N, T_, In, nl, with_hx, ragged):
    """33 .. 64 short sequences (a streaming chunk) as two half-batches interleaved on two streams -- a half's recurrence beside
    the other half's projection, layer calls issued as MS_RNN_PROJECTION_ONLY + MS_RNN_RECURRENCE_ONLY -- against the same batch
    as one call per layer (two batch groups per recurrence launch): outputs and final states ``torch.equal``."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model import rnn as R
    H, ndir = 1024, 2
    torch.manual_seed(N + T_)
    m = R.RNN(R.RNNType.LSTM, In, H, num_layers=nl, bidirectional=True, forget_gate_bias=1.0).eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if "weight_ih" in k:
                v.mul_(8.0)
    x = torch.randn(T_, N, In, device="cuda")
    lens = torch.full((N,), T_, dtype=torch.int64)
    if ragged:
        lens = torch.sort(torch.randint(1, T_ + 1, (N,)), descending=True).values
        lens[0] = T_
    h0 = c0 = None
    if with_hx:
        h0 = (torch.randn(nl * ndir, N, H, device="cuda") * 0.5).contiguous()
        c0 = (torch.randn(nl * ndir, N, H, device="cuda") * 0.5).contiguous()

    def run(halves):
        prev = R._HALVES
        R._HALVES = halves
        try:
            return R.run_layers(_lib.CELL_LSTM, x, _lib.lens_i32(lens), T_, m._layer_params(), [R.PackedLayer() for _ in range(nl)], H,
                                h0, c0, _lib.Workspace(), ragged=False)
        finally:
            R._HALVES = prev
    want, got = run(False), run(True)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert float(want[0].abs().max()) > 0.3
