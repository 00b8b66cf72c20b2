#!/usr/bin/env python3
"""Randomised soak on the GPU box: many small seeded cases of the index-producing paths against the numpy oracle
(CTC prefix beam search with separators / word weights / a toy LM, RNN-T greedy + beam, CTC greedy), plus ragged
LSTM / GRU layers and masked convolutions (all three kernels) against the oracle within 1e-4.  Prints one line per family; exits non-zero on the first mismatch.
    python tests/soak.py [seconds per family, default 40] [first seed, default 0] [substring of the one family to run]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ds_oracle as O  # noqa: E402
from oracle import rnnt_oracle as RO  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


only = sys.argv[3] if len(sys.argv) > 3 else ""


def family(name, one_case):
    if only and only not in name:
        return
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        one_case(seed0 + n)
        n += 1
    print(f"{name:28s} {n:5d} cases ok", flush=True)


def beam_case(seed):
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    rng = np.random.default_rng(seed)
    T_, N, V = int(rng.integers(1, 40)), int(rng.integers(1, 5)), int(rng.integers(2, 9))
    x = torch.softmax(torch.from_numpy(rng.normal(size=(T_, N, V)).astype(np.float32)) * float(rng.uniform(0.5, 6)), dim=2)
    lens = np.sort(rng.integers(0, T_ + 1, size=N))[::-1].copy()
    blank = int(rng.integers(0, V))
    w = int(rng.integers(1, 10))
    prune = float(rng.choice([0.0, 0.001, 0.05]))
    kw = {}
    if V > 2 and rng.random() < 0.5:
        sep = int(rng.choice([v for v in range(V) if v != blank]))
        kw = dict(separator_index=sep, word_weight=float(rng.uniform(0.5, 2.0)))
        if rng.random() < 0.5:
            kw.update(language_model=O.toy_language_model, lm_weight=float(rng.uniform(0.5, 2.0)))
    got = CTCBeamDecoder(blank_index=blank, beam_width=w, prune_threshold=prune, **kw)(x, torch.from_numpy(lens))
    want = O.ctc_beam_decode(x.numpy(), lens, blank, w, prune, kw.get("language_model"), kw.get("lm_weight"),
                             kw.get("separator_index"), kw.get("word_weight", 1.0))
    assert got == want, ("beam", seed, got, want)


def greedy_case(seed):
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    rng = np.random.default_rng(seed)
    T_, N, V = int(rng.integers(1, 700)), int(rng.integers(1, 9)), int(rng.integers(1, 40) if seed % 4 else rng.integers(65, 700))   # (> 64: a wave per frame)
    x = (rng.normal(size=(T_, N, V)) * 2).round(1).astype(np.float32)   # rounded: many exact ties
    lens = rng.integers(0, T_ + 1, size=N)
    blank = int(rng.integers(0, V))
    assert CTCGreedyDecoder(blank)(torch.from_numpy(x), torch.from_numpy(lens)) == O.ctc_greedy_decode(x, lens, blank), ("greedy", seed)


_rnnt = {}
_ties = []   # RNN-T beam cases decided by a score difference of a few ulp (reported, not failed)


def rnnt_case(seed):
    from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder
    rng = np.random.default_rng(seed)
    V = int(rng.choice([3, 5, 11]))
    if V not in _rnnt:
        torch.manual_seed(V)
        pred, joint = RNNTPredictor(V, 8, 64, num_layers=2).eval(), RNNTJoint(24, 64, 32, V).eval()
        _rnnt[V] = (pred, joint, {k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()},
                    {k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()})
    pred, joint, psd, jsd = _rnnt[V]
    T_, N = int(rng.integers(1, 14)), int(rng.integers(1, 4))
    enc = (rng.normal(size=(T_, N, 24)) * float(rng.uniform(0.5, 3))).astype(np.float32)
    lens = np.sort(rng.integers(0, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    w, ms = int(rng.integers(1, 9)), int(rng.integers(1, 4))
    dec = RNNTBeamDecoder(pred, joint, beam_width=w, max_symbols=ms)
    got = dec(torch.from_numpy(enc), torch.from_numpy(lens))
    want, want_scores = RO.beam_decode(enc, lens, psd, jsd, 64, 2, V, w, ms)
    if got != want:
        # The transducer's matrix products are summed in a different order on the device (MFMA k-order, K slices) than
        # in numpy's BLAS, so two hypotheses whose total scores differ by a few ulp may swap.  Anything else is a bug.
        for g, x, gs, xs in zip(got, want, dec.last_scores, want_scores):
            assert g == x or abs(float(gs) - float(xs)) <= 4 * 1.2e-7 * max(1.0, abs(float(xs))), ("rnnt beam", seed, got, want)
        _ties.append(seed)
    got = RNNTGreedyDecoder(pred, joint, max_symbols=ms)(torch.from_numpy(enc), torch.from_numpy(lens))
    assert got == RO.greedy_decode(enc, lens, psd, jsd, 64, 2, V, ms), ("rnnt greedy", seed)


_rnn = {}


def rnn_case(seed):
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    rng = np.random.default_rng(seed)
    # (round 6: widths whose batch groups run side by side, a bidirectional GRU with one launch per direction, tanh-RNN stacks
    # on the GRU kernel, and batches beyond 64 rows)
    kind, H, bidir = [("LSTM", 64, True), ("LSTM", 256, False), ("GRU", 64, True), ("GRU", 128, False), ("BASIC_RNN", 64, True),
                      ("LSTM", 96, True), ("LSTM", 512, True), ("LSTM", 1024, False), ("LSTM", 768, True),
                      ("LSTM", 1024, True), ("GRU", 512, True), ("GRU", 1024, False), ("GRU", 1536, True),
                      ("BASIC_RNN", 512, True), ("BASIC_RNN", 600, False), ("LSTM", 768, False),
                      ("LSTM", 320, True), ("LSTM", 832, False), ("GRU", 2112, False)][seed % 19]     # (padded to 512 / 1 024 / 2 560)
    key = (kind, H, bidir)
    if key not in _rnn:
        torch.manual_seed(seed)
        m = RNN(getattr(RNNType, kind), 32, H, num_layers=2, bidirectional=bidir).eval()
        _rnn[key] = (m, {k[len("rnn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
    m, sd = _rnn[key]
    T_, N = int(rng.integers(1, 12)), int(rng.integers(1, 70) if seed % 3 else rng.integers(60, 141))
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    x = rng.normal(size=(T_, N, 32)).astype(np.float32)
    (out, _), hid = m((torch.from_numpy(x), torch.from_numpy(lens)))
    want, whid = O.rnn_forward(getattr(O, kind), x, lens, sd, H, 2, bidir, None)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    h = hid[0] if isinstance(hid, tuple) else hid
    np.testing.assert_allclose(h.cpu().numpy(), whid[0] if isinstance(whid, tuple) else whid, rtol=1e-4, atol=1e-4)


_packed_rnn = {}


def packed_rows_case(seed):
    """A ragged batch big enough for the packed-rows path (MS_RNN_PACKED_ROWS: >= 1536 rows, up to 64 sequences of very different
    lengths, one or two chained BiLSTM-1024 layers, sometimes a short batch group or one sequence much longer than the rest)."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    rng = np.random.default_rng(seed)
    layers = 1 + seed % 2
    if layers not in _packed_rnn:
        torch.manual_seed(900 + layers)
        m = RNN(RNNType.LSTM, 32, 1024, num_layers=layers, bidirectional=True, forget_gate_bias=1.0).eval()
        _packed_rnn[layers] = (m, {k[len("rnn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
    m, sd = _packed_rnn[layers]
    N = int(rng.integers(33, 65) if seed % 3 else rng.integers(65, 131))      # (beyond 64 rows: wide launches of 64, packed too)
    T_ = int(rng.integers(-(-1536 // N), 64))
    lo = int(rng.integers(1, T_ + 1))
    lens = np.sort(rng.integers(lo, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    if seed % 5 == 0:
        lens[1:] = np.minimum(lens[1:], max(1, T_ // 4))      # one long sequence, the rest short
    assert _lib.load().ms_rnn_layer_packs_rows(0, T_, N, 32, 1024, 2) == 1
    x = rng.normal(size=(T_, N, 32)).astype(np.float32)
    (out, _), (hn, cn) = m((torch.from_numpy(x), torch.from_numpy(lens)))
    want, (wh, wc) = O.rnn_forward(O.LSTM, x, lens, sd, 1024, layers, True, None)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hn.cpu().numpy(), wh, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cn.cpu().numpy(), wc, rtol=1e-4, atol=1e-4)


_overlap_rnn = {}


def overlap_case(seed):
    """The overlapped stack schedule (ms_rnn_stack_forward: time-segment launches of the recurrence, the next layer's projection as
    K-cut / whole GEMM pieces on a second stream) against the layer-by-layer schedule: `torch.equal` outputs and states for random
    step counts, batch sizes <= 32, 2 .. 4 layers, segment counts, lengths that differ (rows not packed) and given initial states
    -- every combination of sliver deferrals, remainders and piece kinds the segment arithmetic can produce."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model import rnn as R
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(2, 5))
    In = int(rng.choice([32, 64, 96]))
    key = (nl, In)
    if key not in _overlap_rnn:
        torch.manual_seed(700 + nl * 10 + In)
        m = R.RNN(R.RNNType.LSTM, In, 1024, num_layers=nl, bidirectional=True, forget_gate_bias=1.0).eval()
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if "weight_ih" in k:
                    v.mul_(6.0)
        _overlap_rnn[key] = m
    m = _overlap_rnn[key]
    N = int(rng.integers(1, 33))
    T_ = int(rng.integers(max(2, -(-300 // N)), 420))
    segs = int(rng.integers(2, 17))
    lib = _lib.load()
    if not lib.ms_rnn_stack_overlap_ok(_lib.CELL_LSTM, T_, N, In, 1024, 2, nl):
        return
    x = torch.from_numpy(rng.normal(size=(T_, N, In)).astype(np.float32)).cuda()
    lens = np.full(N, T_)
    if seed % 3 == 0:
        lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy()
        lens[0] = T_
    lens_t = torch.from_numpy(lens)
    h0 = c0 = None
    if seed % 4 == 1:
        h0 = torch.from_numpy(rng.normal(size=(nl * 2, N, 1024)).astype(np.float32) * 0.5).cuda()
        c0 = torch.from_numpy(rng.normal(size=(nl * 2, N, 1024)).astype(np.float32) * 0.5).cuda()

    def run(overlap):
        prev = (R._OVERLAP, R._OVERLAP_SEGMENTS)
        R._OVERLAP, R._OVERLAP_SEGMENTS = overlap, segs
        try:
            return R.run_layers(_lib.CELL_LSTM, x, _lib.lens_i32(lens_t), T_, m._layer_params(), m._packed, 1024, h0, c0,
                                m._workspace, ragged=False)
        finally:
            R._OVERLAP, R._OVERLAP_SEGMENTS = prev
    want, got = run(False), run(True)
    for a, b in zip(want, got):
        assert torch.equal(a, b), (seed, nl, In, N, T_, segs, float((a - b).abs().max()))


def ctc_case(seed):
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(seed)
    T_, N, V = int(rng.integers(1, 30)), int(rng.integers(1, 5)), int(rng.integers(2, 8) if seed % 4 else rng.integers(65, 1300))   # (> 64: phase 1 chip-wide)
    blank = int(rng.integers(0, V))
    labels = [v for v in range(V) if v != blank]
    x = (rng.normal(size=(T_, N, V)) * 2).astype(np.float32)
    xl = rng.integers(1, T_ + 1, size=N)
    S = int(rng.integers(1, 8))
    y = rng.choice(labels, size=(N, S)).astype(np.int32)
    yl = rng.integers(0, S + 1, size=N).astype(np.int32)
    red = ["none", "mean", "sum"][seed % 3]
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    out = CTCLoss(blank=blank, reduction=red, zero_infinity=True)((xt, torch.from_numpy(xl)), (torch.from_numpy(y), torch.from_numpy(yl)))
    want = O.ctc_loss(x, xl, y, yl, blank, red, True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    wts = rng.uniform(0.5, 1.5, size=N).astype(np.float32)
    (out * torch.from_numpy(wts).cuda()).sum().backward() if red == "none" else out.backward()
    gn = wts if red == "none" else (np.ones(N, np.float32) if red == "sum" else 1.0 / (np.maximum(yl.astype(np.float32), 1.0) * N))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), O.ctc_grad(x, xl, y, yl, gn, blank, True), rtol=1e-3, atol=1e-4)


def ctc_long_case(seed):
    """The four-wave alpha pipeline at sizes that use all of its waves and every states-per-lane form (targets of up to 520
    labels: S <= 256 / 512 / 1024 and the LDS-row kernel beyond), ragged inputs: against the LDS-row kernel (MS_CTC_WAVE=0)
    on every case and against the oracle on every fourth."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(seed)
    L = int(rng.choice([int(rng.integers(1, 128)), int(rng.integers(128, 256)), int(rng.integers(256, 521))]))
    T_ = int(rng.integers(max(2, L // 2), 2 * L + 40))
    N, V = int(rng.integers(1, 6)), int(rng.integers(3, 40))
    blank = int(rng.integers(0, V))
    labels = [v for v in range(V) if v != blank]
    x = (rng.normal(size=(T_, N, V)) * float(rng.uniform(0.5, 6))).astype(np.float32)
    xl = rng.integers(1, T_ + 1, size=N)
    xl[0] = T_
    y = rng.choice(labels, size=(N, L)).astype(np.int32)
    yl = rng.integers(0, L + 1, size=N).astype(np.int32)
    yl[0] = L
    loss = CTCLoss(blank=blank, reduction="none")
    args = ((torch.from_numpy(x), torch.from_numpy(xl)), (torch.from_numpy(y), torch.from_numpy(yl)))
    os.environ["MS_CTC_WAVE"] = "0"
    old = loss(*args).cpu().numpy()
    os.environ["MS_CTC_WAVE"] = "1"
    got = loss(*args).cpu().numpy()
    fin = np.isfinite(old)
    assert (np.isfinite(got) == fin).all(), (seed, got, old)
    np.testing.assert_array_equal(got[~fin], old[~fin])
    np.testing.assert_allclose(got[fin], old[fin], rtol=3e-6, atol=3e-5)
    if seed % 4 == 0:
        want = O.ctc_loss(x, xl, y, yl, blank, "none")
        assert (np.isfinite(want) == fin).all(), (seed, got, want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=2e-4, atol=1e-2)


def frontend_case(seed):
    from oracle import frontend_oracle as FO
    from myrtlespeech_amd.data.preprocess import AddContextFrames, MFCC, MFCCLegacy, Standardize
    rng = np.random.default_rng(seed)
    N = int(rng.integers(1, 5))
    lens = np.sort(rng.integers(201, 6000, size=N))[::-1].copy()
    w = np.zeros((N, lens[0]), np.float32)
    for i, l in enumerate(lens):
        w[i, :l] = (rng.normal(size=l) * 0.2).clip(-1, 1)
    n_mfcc, hop = int(rng.integers(1, 129)), int(rng.choice([80, 160, 200, 320]))
    y, fl = MFCC(n_mfcc=n_mfcc, melkwargs={"win_length": 400, "hop_length": hop}).batch(torch.from_numpy(w), torch.from_numpy(lens))
    want = FO.pad_sequence([FO.mfcc(w[i:i + 1, :l], n_mfcc, 400, hop) for i, l in enumerate(lens)])
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-4, atol=3e-3)
    s, _ = Standardize().batch(y, fl)
    wants = FO.pad_sequence([FO.standardize(want[i][..., :int(f)]) for i, f in enumerate(fl.tolist())])
    if int(fl.min()) * n_mfcc > 1:
        np.testing.assert_allclose(s.cpu().numpy(), wants, rtol=1e-3, atol=1e-3)
    c = int(rng.integers(0, 6))
    ctx, _ = AddContextFrames(c).batch(y, fl)
    wantc = FO.pad_sequence([FO.add_context_frames(y[i, :, :, :int(f)].cpu().numpy(), c) for i, f in enumerate(fl.tolist())])
    assert np.array_equal(ctx.cpu().numpy(), wantc)
    nm = int(rng.integers(1, 27))
    ly, lf = MFCCLegacy(nm, {"win_length": 400, "hop_length": hop}).batch(torch.from_numpy(w), torch.from_numpy(lens))
    wantl = FO.pad_sequence([FO.mfcc_legacy(w[i:i + 1, :l], nm, 400, hop) for i, l in enumerate(lens)])
    np.testing.assert_allclose(ly.cpu().numpy(), wantl, rtol=1e-5, atol=1e-5)


def conv_case(seed):
    """Convolutions large enough for the MFMA paths: single-channel tall filters (feature-window kernel), multi-channel
    filters (channels-last kernel) and -- every third case -- a small shape on the exact-f32 tap kernel."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    rng = np.random.default_rng(seed)
    kind = seed % 3
    if kind == 0:      # Cin = 1, tall filter, even feature stride
        cin, cout = 1, int(rng.choice([8, 32, 40]))
        k = [int(rng.integers(16, 49)), int(rng.integers(3, 14))]
        s = [int(rng.choice([2, 4])), int(rng.choice([1, 2]))]
        F = int(rng.integers(max(k[0], 32), 97))
    elif kind == 1:    # Cin % 16 == 0
        cin, cout = int(rng.choice([16, 32, 48])), int(rng.choice([16, 32, 33]))
        k = [int(rng.integers(1, 8)), int(rng.integers(1, 12))]
        s = [int(rng.choice([1, 2])), int(rng.choice([1, 2]))]
        F = int(rng.integers(max(k[0], 8), 33))
    else:
        cin, cout = int(rng.integers(1, 5)), int(rng.integers(1, 20))
        k = [int(rng.integers(1, 6)), int(rng.integers(1, 8))]
        s = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
        F = int(rng.integers(k[0], 20))
    same = bool(rng.random() < 0.6)
    N = int(rng.integers(1, 5))
    fo = (F + s[0] - 1) // s[0] if same else (F - k[0]) // s[0] + 1
    # frames: enough for the 1e9-flop routing threshold on the MFMA kinds, small on the tap kernel
    if kind < 2:
        per_frame = 2.0 * N * cout * max(fo, 1) * cin * k[0] * k[1] / s[1]
        Tn = int(min(max(1.05e9 / per_frame, 4 * k[1]), 6000)) + int(rng.integers(0, 40))
    else:
        Tn = int(rng.integers(k[1], 60))
    key = ("conv", cin, cout, tuple(k), tuple(s), same)
    torch.manual_seed(seed)
    m = MaskConv2d(cin, cout, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
    x = rng.normal(size=(N, cin, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(max(Tn // 2, k[1]), Tn + 1, size=N))[::-1].copy()
    y, nl = m((torch.from_numpy(x), torch.from_numpy(lens)), fused_activation=(0.0, 20.0))
    want, wl = O.mask_conv2d(x, lens, m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), tuple(s), same)
    np.testing.assert_allclose(y.cpu().numpy(), np.clip(want, 0.0, 20.0), rtol=1e-4, atol=3e-4, err_msg=str(key))
    assert np.array_equal(nl.cpu().numpy(), wl), key


def linear_case(seed):
    """The three GEMM kernels at random shapes: exact-f32 tiles (ms_linear_forward: every tile configuration, ragged edges,
    K not a multiple of 4), split-bf16 (ms_linear_split_forward: 256 x 256 and 256 x 128 tiles, K % 32 == 0)."""
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    big = seed % 4 < 2            # both the exact-f32 and the split kernels see large shapes (M * N >= 4 Mi takes the 256 x 256 tiles)
    M = int(rng.integers(1, 3000 if big else 400))
    N = int(rng.integers(1, 3000 if big else 400))
    split = seed % 2 == 1
    K = 32 * int(rng.integers(1, 20)) if split else int(rng.integers(1, 600))
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32) if rng.random() < 0.7 else None
    act = int(rng.random() < 0.5)
    lo, hi = (0.0, 1.5) if act else (0.0, 0.0)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    bd = None if b is None else torch.from_numpy(b).cuda()
    y = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    if split:
        ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ms_linear_split_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, act, lo, hi,
                                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "split")
        if M * N >= 4 * 1024 * 1024:   # the LDS-DMA kernels (8-wave and 4-wave co-tenant form) must agree bit for bit with the register-staged one
            for variant in (2, 7):
                y2 = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                lib.ms_gemm_set_variant(variant)
                _lib.check(lib.ms_linear_split_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y2), M, K, N, act, lo, hi,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "split (variant)")
                lib.ms_gemm_set_variant(0)
                assert torch.equal(y, y2), ("split kernels differ", variant, seed, M, K, N)
    else:
        _lib.check(lib.ms_linear_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, act, lo, hi,
                                         _lib.stream_ptr()), "linear")
    want = x.astype(np.float64) @ w.astype(np.float64).T + (0.0 if b is None else b.astype(np.float64))
    if act:
        want = np.clip(want, lo, hi)
    tol = 1e-4 if split else 2e-5   # bf16x3: ~2^-17 per product; exact f32: summation order only
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=tol, atol=tol, err_msg=str(("linear", seed, M, K, N, split, act)))


def lookahead_case(seed):
    from myrtlespeech_amd.model.lookahead import Lookahead
    rng = np.random.default_rng(seed)
    Fd, ctx = int(rng.integers(1, 700)), int(rng.integers(1, 90))
    N, T_ = int(rng.integers(1, 5)), int(rng.integers(1, 300))
    torch.manual_seed(seed)
    m = Lookahead(Fd, ctx).eval()
    x = rng.normal(size=(N, Fd, T_)).astype(np.float32)
    y, _ = m((torch.from_numpy(x), torch.full((N,), T_)))
    want = O.lookahead(x, m.weight.detach().cpu().numpy())
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-4, atol=1e-5, err_msg=str(("lookahead", seed, Fd, ctx, N, T_)))


def conv1d_case(seed):
    """MaskConv1d, small (tap kernel) and large (im2col + split GEMM lowering) shapes."""
    from myrtlespeech_amd.model.cnn import MaskConv1d, PaddingMode
    rng = np.random.default_rng(seed)
    large = seed % 2 == 0
    cin = int(rng.choice([64, 80, 128, 200])) if large else int(rng.integers(1, 40))
    cout = int(rng.choice([32, 100, 256])) if large else int(rng.integers(1, 40))
    k, st = int(rng.integers(1, 12)), int(rng.integers(1, 4))
    same = bool(rng.random() < 0.6)
    N = int(rng.integers(1, 5))
    if large:
        Tn = int(min(max(1.05e9 * st / (2.0 * N * cout * cin * k), 4 * k), 8000)) + int(rng.integers(0, 30))
    else:
        Tn = int(rng.integers(k, 200))
    torch.manual_seed(seed)
    m = MaskConv1d(cin, cout, k, st, PaddingMode.SAME if same else PaddingMode.NONE).eval()
    x = rng.normal(size=(N, cin, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(max(Tn // 2, k), Tn + 1, size=N))[::-1].copy()
    y, nl = m((torch.from_numpy(x), torch.from_numpy(lens)))
    want, wl = O.mask_conv1d(x, lens, m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), st, same)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-4, atol=3e-4, err_msg=str(("conv1d", seed, cin, cout, k, st, same, N, Tn)))
    assert np.array_equal(nl.cpu().numpy(), wl), ("conv1d lens", seed)


_build_ds2 = None


def ds2_case(seed):
    """A whole random DeepSpeech2 (1-2 masked conv2d layers, LSTM / GRU / tanh-RNN, optional lookahead, 0-1 hidden FC
    layers) against the oracle's forward: the glue between the kernels (layout changes, lengths, fused activations)."""
    global _build_ds2
    if _build_ds2 is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_gpu_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                  "test_gpu_parity.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        spec.loader.exec_module(mod)
        _build_ds2 = mod.build_ds2
    rng = np.random.default_rng(seed)
    F, Tn, N = int(rng.integers(8, 25)), int(rng.integers(20, 81)), int(rng.integers(1, 6))
    convs, cin, f = [], 1, F
    for i in range(int(rng.integers(1, 3))):
        cout = int(rng.integers(2, 9))
        k = [int(rng.integers(1, 6)), int(rng.integers(1, 8))]
        st = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
        convs.append(dict(kind="conv2d", idx=2 * i, in_channels=cin, out_channels=cout, kernel=k, stride=st, same=True,
                          act=(0.0, 20.0) if rng.random() < 0.7 else None))
        f, cin = -(-f // st[0]), cout
    bidir = bool(rng.random() < 0.5)
    H = int(rng.choice([32, 64, 96]))
    rnn = dict(kind=int(rng.integers(0, 3)), input=cin * f, hidden=H, layers=int(rng.integers(1, 3)), bidirectional=bidir,
               forget_gate_bias=1.0 if rng.random() < 0.5 else None)
    if rnn["kind"] != 0:
        rnn["forget_gate_bias"] = None
    la = None
    if not bidir and rng.random() < 0.6:
        la = dict(context=int(rng.integers(1, 12)), act=(0.0, 20.0) if rng.random() < 0.5 else None)
    nh = int(rng.integers(0, 2))
    fc = dict(in_features=H * (2 if bidir else 1), out_features=int(rng.integers(5, 31)), n_hidden=nh,
              hidden=int(rng.integers(16, 49)) if nh else None, act=(0.0, 20.0) if nh else None)
    cfg = dict(convs=convs, rnn=rnn, lookahead=la, fc=fc)
    torch.manual_seed(seed)
    m = _build_ds2(cfg).eval()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    x = rng.normal(size=(N, 1, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(Tn // 2, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    (y, yl), _ = m((torch.from_numpy(x.copy()), torch.from_numpy(lens)))
    want, wl, _ = O.deep_speech_2_forward(x, lens, cfg, sd)
    assert np.array_equal(yl.cpu().numpy(), wl), ("ds2 lens", seed)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-4, atol=2e-4, err_msg=str(("ds2", seed, cfg)))


def ds1_case(seed):
    """A whole random DeepSpeech1 (torch-LSTM and HardLSTM flavours) against the oracle's forward."""
    from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
    rng = np.random.default_rng(seed)
    C, Fd, H = int(rng.integers(1, 6)), int(rng.integers(4, 20)), int(rng.choice([24, 32, 64, 96]))
    V, N, Tn = int(rng.integers(5, 31)), int(rng.integers(1, 6)), int(rng.integers(1, 60))
    hard = bool(seed % 2)
    torch.manual_seed(seed)
    m = DeepSpeech1(Fd, C, H, V, drop_prob=0.25, relu_clip=20.0, forget_gate_bias=1.0, hard_lstm=hard).eval()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    x = rng.normal(size=(N, C, Fd, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    (y, yl), _ = m((torch.from_numpy(x.copy()), torch.from_numpy(lens)))
    want, wl, _ = O.deep_speech_1_forward(x, lens, sd, H, 20.0, hard)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-4, atol=2e-4, err_msg=str(("ds1", seed, C, Fd, H, V, N, Tn, hard)))


def stream_case(seed):
    """Chunked streaming against the unchunked run of the same network on the device: unidirectional recurrences, time
    kernel 1 (no conv context to carry), so feeding the state back chunk by chunk must reproduce the whole utterance."""
    global _build_ds2
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    if _build_ds2 is None:
        ds2_case(seed)
    rng = np.random.default_rng(seed)
    F, Tn, N = int(rng.integers(8, 25)), int(rng.integers(10, 90)), int(rng.integers(1, 6))
    cout, kf, sf = int(rng.integers(2, 9)), int(rng.integers(1, 6)), int(rng.integers(1, 3))
    convs = [dict(kind="conv2d", idx=0, in_channels=1, out_channels=cout, kernel=[kf, 1], stride=[sf, 1], same=True, act=(0.0, 20.0))]
    H = int(rng.choice([32, 64, 96]))
    kind = int(rng.integers(0, 3))
    rnn = dict(kind=kind, input=cout * (-(-F // sf)), hidden=H, layers=int(rng.integers(1, 3)), bidirectional=False,
               forget_gate_bias=1.0 if kind == 0 else None)
    fc = dict(in_features=H, out_features=int(rng.integers(5, 31)), n_hidden=0, hidden=None, act=None)
    torch.manual_seed(seed)
    m = _build_ds2(dict(convs=convs, rnn=rnn, lookahead=None, fc=fc)).eval()
    x = rng.normal(size=(N, 1, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    (y, yl), hid = m((torch.from_numpy(x.copy()), torch.from_numpy(lens)))
    (ys, ysl), hids = ChunkedDeepSpeech2(m, int(rng.integers(3, 18)))(torch.from_numpy(x.copy()), torch.from_numpy(lens))
    assert yl.tolist() == ysl.tolist(), ("stream lens", seed)
    yw, yc = y.cpu().numpy(), ys.cpu().numpy()
    for n in range(N):   # frames past an utterance's length hold the FC bias in one run and may be absent in the other
        np.testing.assert_allclose(yc[:lens[n], n], yw[:lens[n], n], rtol=1e-4, atol=1e-4, err_msg=str(("stream", seed, n)))
    h_w = hid[0] if isinstance(hid, tuple) else hid
    h_c = hids[0] if isinstance(hids, tuple) else hids
    np.testing.assert_allclose(h_c.cpu().numpy(), h_w.cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=str(("stream state", seed)))



def stream_ctx_case(seed):
    """Streaming with carried context (round 4): a random UNIDIRECTIONAL DeepSpeech2 (1-2 conv2d layers with time kernels up to 7
    and strides 1-3, optional lookahead, 0-1 hidden FC layers) fed in chunks of a random size against the ORACLE's
    full-utterance forward (which is pinned to the reference's): logits on every frame an utterance owns, lengths, final
    states."""
    global _build_ds2
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    if _build_ds2 is None:
        ds2_case(seed)
    rng = np.random.default_rng(seed)
    F, Tn, N = int(rng.integers(8, 25)), int(rng.integers(4, 120)), int(rng.integers(1, 7))
    convs, cin, f = [], 1, F
    for i in range(int(rng.integers(1, 3))):
        cout = int(rng.integers(2, 9))
        k = [int(rng.integers(1, 6)), int(rng.integers(1, 8))]
        st = [int(rng.integers(1, 3)), int(rng.integers(1, 4))]
        convs.append(dict(kind="conv2d", idx=2 * i, in_channels=cin, out_channels=cout, kernel=k, stride=st, same=True,
                          act=(0.0, 20.0) if rng.random() < 0.7 else None))
        f = -(-f // st[0]) if convs[-1]["same"] else (f - k[0]) // st[0] + 1
        cin = cout
        if f < 1:
            return
    # without SAME padding a layer may leave no frames for short inputs: keep the case only if every layer has output
    t = Tn
    for c in convs:
        t = -(-t // c["stride"][1]) if c["same"] else (t - c["kernel"][1]) // c["stride"][1] + 1
        if t < 1:
            return
    H = int(rng.choice([32, 64, 96]))
    kind = int(rng.integers(0, 3))
    rnn = dict(kind=kind, input=cin * f, hidden=H, layers=int(rng.integers(1, 3)), bidirectional=False,
               forget_gate_bias=1.0 if kind == 0 else None)
    la = dict(context=int(rng.integers(1, 12)), act=(0.0, 20.0) if rng.random() < 0.5 else None) if rng.random() < 0.6 else None
    nh = int(rng.integers(0, 2))
    fc = dict(in_features=H, out_features=int(rng.integers(5, 31)), n_hidden=nh, hidden=int(rng.integers(16, 49)) if nh else None,
              act=(0.0, 20.0) if nh else None)
    cfg = dict(convs=convs, rnn=rnn, lookahead=la, fc=fc)
    torch.manual_seed(seed)
    m = _build_ds2(cfg).eval()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    x = rng.normal(size=(N, 1, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    # (without SAME padding an utterance shorter than a kernel has no output frames: the reference's pack_padded_sequence
    # rejects it; keep lengths that leave every utterance at least one frame)
    want, wl, whid = O.deep_speech_2_forward(x, lens, cfg, sd)
    if int(np.min(wl)) < 1:
        return
    chunk = int(rng.integers(1, 40))
    (ys, ysl), hids = ChunkedDeepSpeech2(m, chunk, carry_context=True)(torch.from_numpy(x.copy()), torch.from_numpy(lens))
    assert ysl.tolist() == [int(v) for v in wl], ("stream ctx lens", seed)
    yc = ys.cpu().numpy()
    assert yc.shape == want.shape, ("stream ctx shape", seed, yc.shape, want.shape)
    for n in range(N):
        np.testing.assert_allclose(yc[:wl[n], n], want[:wl[n], n], rtol=2e-4, atol=2e-4, err_msg=str(("stream ctx", seed, n, chunk, cfg)))
    h_c = hids[0] if isinstance(hids, tuple) else hids
    h_w = whid[0] if isinstance(whid, tuple) else whid
    np.testing.assert_allclose(h_c.cpu().numpy(), h_w, rtol=2e-4, atol=2e-4, err_msg=str(("stream ctx state", seed)))


def conv_short_case(seed):
    """``maskconv_cl_short_kernel`` (32 input channels, <= 16 output frames, time stride 1) on random geometry against the oracle."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    os.environ["MS_CONV_MFMA_MIN_FLOPS"] = "0"
    rng = np.random.default_rng(seed)
    cout = int(rng.integers(1, 70))
    kf, kt = int(rng.integers(1, 24)), int(rng.integers(1, 12))
    sf, df = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    same = bool(rng.random() < 0.6)
    F = int(rng.integers(df * (kf - 1) + 1, df * (kf - 1) + 45))
    Tn = int(rng.integers(1, 17)) if same else int(rng.integers(kt, kt + 16))
    N = int(rng.integers(1, 80))
    torch.manual_seed(seed)
    m = MaskConv2d(32, cout, [kf, kt], [sf, 1], PaddingMode.SAME if same else PaddingMode.NONE, dilation=[df, 1]).eval()
    x = (rng.normal(size=(N, 32, F, Tn)) + 0.5).astype(np.float32)
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    act = (0.0, 20.0) if rng.random() < 0.5 else None
    y, nl = m((torch.from_numpy(x.copy()), torch.from_numpy(lens)), fused_activation=act)
    want, wl = O.mask_conv2d(x, lens, m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), (sf, 1), same, dilation=(df, 1))
    if act is not None:
        want = np.clip(want, *act)
    assert y.shape[-1] <= 16 or not same
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-4, atol=3e-3, err_msg=str(("conv short", seed, cout, kf, kt, sf, df, same, F, Tn, N)))
    assert np.array_equal(nl.cpu().numpy(), wl)


family("ctc loss+grad vs oracle", ctc_case)
family("ctc loss, long targets (alpha pipeline)", ctc_long_case)
family("front-end vs oracle", frontend_case)
family("ctc beam vs oracle", beam_case)
family("ctc greedy vs oracle", greedy_case)
family("rnn-t greedy+beam vs oracle", rnnt_case)
if _ties:
    print(f"  rnn-t beam: {len(_ties)} case(s) decided by a <= 4 ulp score difference (summation order): seeds {_ties[:8]}", flush=True)
family("lstm/gru/rnn vs oracle", rnn_case)
family("ragged LSTM-1024, packed rows", packed_rows_case)
family("overlapped stack == layer by layer", overlap_case)
family("mask-conv2d vs oracle", conv_case)
family("mask-conv1d vs oracle", conv1d_case)
family("linear kernels vs float64", linear_case)
family("lookahead vs oracle", lookahead_case)
family("whole random DS2 vs oracle", ds2_case)
family("whole random DS1 vs oracle", ds1_case)
family("chunked streaming vs whole", stream_case)
family("carried-context streaming vs oracle", stream_ctx_case)
family("short-input conv vs oracle", conv_short_case)
print("soak ok")
