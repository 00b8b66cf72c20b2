"""Pin the CPU oracle (oracle/ds_oracle.py) against the golden vectors produced by
the reference itself (tests/golden/gen_golden.py) and the reference's own KATs.
CPU only."""
import numpy as np
import torch
import pytest

from oracle import ds_oracle as O
from util import Golden, golden_names, unragged

TOL = dict(rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", golden_names("rnn_"))
def test_rnn(name):
    g = Golden(name)
    c = g.cfg
    hx = None
    if g.has("in/h0"):
        hx = (g["in/h0"], g["in/c0"]) if c["rnn_type"] == 0 else g["in/h0"]
    sd = {k[len("rnn."):]: v for k, v in g.sd().items()}
    out, hid = O.rnn_forward(c["rnn_type"], g["in/x"], g["in/lens"], sd, c["hidden_size"], c["num_layers"],
                             c["bidirectional"], hx, c["batch_first"])
    np.testing.assert_allclose(out, g["out/y"], **TOL)
    if c["rnn_type"] == 0:
        np.testing.assert_allclose(hid[0], g["out/hn"], **TOL)
        np.testing.assert_allclose(hid[1], g["out/cn"], **TOL)
    else:
        np.testing.assert_allclose(hid, g["out/hn"], **TOL)


@pytest.mark.parametrize("name", golden_names("hard_lstm_"))
def test_hard_lstm(name):
    g = Golden(name)
    c = g.cfg
    sd = {k[len("rnn."):]: v for k, v in g.sd().items()}
    out, hid = O.hard_lstm_forward(g["in/x"], sd, c["hidden_size"], c["num_layers"], c["bidirectional"],
                                   (g["in/h0"], g["in/c0"]), c["batch_first"])
    np.testing.assert_allclose(out, g["out/y"], **TOL)
    np.testing.assert_allclose(hid[0], g["out/hn"], **TOL)
    np.testing.assert_allclose(hid[1], g["out/cn"], **TOL)


@pytest.mark.parametrize("name", golden_names("conv2d_"))
def test_conv2d(name):
    g = Golden(name)
    c = g.cfg
    y, nl = O.mask_conv2d(g["in/x"], g["in/lens"], g["sd/weight"], g["sd/bias"], tuple(c["stride"]), c["same"])
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(nl, g["out/lens"])
    assert nl.dtype == g["out/lens"].dtype
    np.testing.assert_array_equal(O._mask_time(g["in/x"], g["in/lens"]), g["out/x_after"])


@pytest.mark.parametrize("name", golden_names("conv1d_"))
def test_conv1d(name):
    g = Golden(name)
    c = g.cfg
    y, nl = O.mask_conv1d(g["in/x"], g["in/lens"], g["sd/weight"], g["sd/bias"], c["stride"], c["same"])
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(nl, g["out/lens"])
    assert nl.dtype == g["out/lens"].dtype


def test_pad_same_values():
    # SURVEY 8a6, computed with the reference
    assert O.pad_same(80, 41, 2) == (20, 20)
    assert O.pad_same(1001, 11, 2) == (5, 6)
    assert O.pad_same(40, 21, 2) == (10, 10)
    assert O.pad_same(501, 11, 1) == (5, 5)
    for bad in [(0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0)]:
        with pytest.raises(ValueError):
            O.pad_same(*bad)


@pytest.mark.parametrize("name", golden_names("fc_"))
def test_fc(name):
    g = Golden(name)
    c = g.cfg
    sd = g.sd()
    if c["num_hidden_layers"] == 0:
        layers = [(sd["fully_connected.weight"], sd["fully_connected.bias"])]
    else:
        idx = sorted({int(k.split(".")[1]) for k in sd if k.endswith(".weight")})
        layers = [(sd[f"fully_connected.{i}.weight"], sd[f"fully_connected.{i}.bias"]) for i in idx]
    act = (0.0, np.inf) if c["act"] == "relu" else c["act"]
    y = O.fully_connected(g["in/x"], layers, act)
    np.testing.assert_allclose(y, g["out/y"], **TOL)


@pytest.mark.parametrize("name", golden_names("lookahead_"))
def test_lookahead(name):
    g = Golden(name)
    np.testing.assert_allclose(O.lookahead(g["in/x"], g["sd/weight"]), g["out/y"], **TOL)


@pytest.mark.parametrize("name", golden_names("ds2_tiny"))
def test_ds2_tiny(name):
    g = Golden(name)
    c = g.cfg
    hx = g["in/h0"] if g.has("in/h0") else None
    cfg = dict(convs=[dict(kind=v["kind"], idx=v["idx"], stride=v["stride"], same=v["same"], act=v["act"])
                      for v in c["convs"]],
               rnn=dict(kind=c["rnn"]["kind"], hidden=c["rnn"]["hidden"], layers=c["rnn"]["layers"],
                        bidirectional=c["rnn"]["bidirectional"]),
               lookahead=c["lookahead"], fc=dict(n_hidden=c["fc"]["n_hidden"], act=c["fc"]["act"]))
    y, nl, hid = O.deep_speech_2_forward(g["in/x"], g["in/lens"], cfg, g.sd(), hx)
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(nl, g["out/lens"])
    if g.has("out/greedy_flat"):
        assert O.ctc_greedy_decode(y, nl, c["blank"]) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])


def test_torch_cpu_baseline_matches_reference_golden():
    """The stock-torch CPU baseline timed by bench.py (oracle/torch_cpu.py) is pinned like the numpy oracle."""
    from oracle import torch_cpu as TC
    g = Golden("ds2_tiny_bilstm")
    c = g.cfg
    cfg = dict(convs=[dict(kind=v["kind"], idx=v["idx"], stride=v["stride"], same=v["same"], act=v["act"])
                      for v in c["convs"]],
               rnn=dict(kind=c["rnn"]["kind"], hidden=c["rnn"]["hidden"], layers=c["rnn"]["layers"],
                        bidirectional=c["rnn"]["bidirectional"]),
               lookahead=None, fc=dict(n_hidden=c["fc"]["n_hidden"], act=c["fc"]["act"]))
    y, nl = TC.deep_speech_2_forward(g["in/x"], g["in/lens"], cfg, g.sd())
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(nl, g["out/lens"])
    assert TC.ctc_greedy_decode(y, nl, c["blank"]) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])


def test_torch_cpu_ds1_baseline_matches_reference_golden():
    """The stock-torch DS1 assembly that bench.py's ``cfg1_ds1`` leg times as its CPU baseline, against the reference's output."""
    from oracle import torch_cpu as TC
    g = Golden("ds1_tiny")
    c = g.cfg
    y, nl = TC.deep_speech_1_forward(g["in/x"], g["in/lens"], g.sd(), c["n_hidden"], c["relu_clip"])
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(nl, g["in/lens"])


@pytest.mark.parametrize("name", golden_names("ds1_tiny"))
def test_ds1_tiny(name):
    g = Golden(name)
    c = g.cfg
    y, nl, hid = O.deep_speech_1_forward(g["in/x"], g["in/lens"], g.sd(), c["n_hidden"], c["relu_clip"],
                                         c["hard_lstm"])
    np.testing.assert_allclose(y, g["out/y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hid[0], g["out/hn"], rtol=1e-4, atol=1e-4)


def test_ctc_loss_small():
    g = Golden("ctc_loss_small")
    b = g.cfg["blank"]
    for red in ("none", "mean", "sum"):
        for zi in (0, 1):
            got = O.ctc_loss(g["in/x"], g["in/x_lens"], g["in/y"], g["in/y_lens"], b, red, bool(zi))
            np.testing.assert_allclose(got, g[f"out/{red}_{zi}"], rtol=1e-4, atol=1e-4)
    got = O.ctc_loss(g["in/x"], g["in/x_lens"], g["in/y_flat"], g["in/y_lens"], b, "none")
    np.testing.assert_allclose(got, g["out/none_flat"], rtol=1e-4, atol=1e-4)


def test_ctc_loss_dim():
    """CTCLoss(dim != -1) as the reference runs it (fixture made by the reference with dim in {0, 1, -3, -2, 2})."""
    g = Golden("ctc_loss_dim")
    for dim in g.cfg["dims"]:
        for red in ("none", "mean", "sum"):
            got = O.ctc_loss(g["in/x"], g["in/x_lens"], g["in/y"], g["in/y_lens"], g.cfg["blank"], red, dim=dim)
            np.testing.assert_allclose(got, g[f"out/dim{dim}_{red}"], rtol=1e-4, atol=1e-4)
    assert not np.allclose(g["out/dim0_none"], g["out/dim2_none"])


def test_ctc_loss_v29():
    g = Golden("ctc_loss_v29")
    for red in ("none", "mean", "sum"):
        got = O.ctc_loss(g["in/x"], g["in/x_lens"], g["in/y"], g["in/y_lens"], 28, red)
        np.testing.assert_allclose(got, g[f"out/{red}_0"], rtol=1e-4, atol=1e-3)


def test_greedy_golden():
    g = Golden("greedy_ties")
    for b in g.cfg["blanks"]:
        assert O.ctc_greedy_decode(g["in/x"], g["in/lens"], b) == unragged(g[f"out/flat_b{b}"], g[f"out/lens_b{b}"])


def test_greedy_reference_kat():
    # restates tests/post_process/test_ctc_greedy_decoder.py:16-102: logits whose argmax
    # spells a blank-separated sentence decode to exactly that sentence
    rng = np.random.default_rng(0)
    V, blank = 7, 6
    sent = [[1, 2, 2, 3], [4], []]
    rows = []
    for s in sent:
        path = []
        for k, c in enumerate(s):
            path += [c] * int(rng.integers(1, 4)) + [blank] * int(rng.integers(1, 3))
        rows.append(path)
    T = max(len(r) for r in rows) + 2
    x = rng.normal(size=(T, len(sent), V)).astype(np.float32)
    lens = []
    for n, r in enumerate(rows):
        for t, c in enumerate(r):
            x[t, n, c] = 10.0
        lens.append(len(r))
    for dt in (np.uint8, np.int8, np.int16, np.int32, np.int64):
        assert O.ctc_greedy_decode(x, np.array(lens, dtype=dt), blank) == sent
    with pytest.raises(ValueError):
        O.ctc_greedy_decode(x, np.array(lens, dtype=np.float32), blank)
    with pytest.raises(ValueError):
        O.ctc_greedy_decode(x, np.array(lens[:-1]), blank)
    with pytest.raises(ValueError):
        O.ctc_greedy_decode(x, np.array([T + 1] * len(sent)), blank)


def test_beam_kats():
    g = Golden("beam_kats")
    assert O.ctc_beam_decode(g["kat2x2/x"], np.array([2], np.int8), 1, 2, 0.0) == [list(g["kat2x2/out"])]
    x = g["katlm/x"]
    ln = np.array([4], np.int8)
    al = dict(zip("deouw_ ", range(7)))
    assert O.ctc_beam_decode(x, ln, al["_"], 20) == [[al[c] for c in "do"]] == [list(g["katlm/out_nolm"])]
    for target in ("dew", "due"):
        tt = tuple(al[c] for c in target) + (al[" "],)
        got = O.ctc_beam_decode(x, ln, al["_"], 20, language_model=lambda w, tt=tt: 2.0 if w == tt else 0.0,
                                lm_weight=10.0, separator_index=al[" "], word_weight=2.0)
        assert got == [[al[c] for c in target + " "]] == [list(g[f"katlm/out_{target}"])]


def test_beam_random():
    g = Golden("beam_random")
    for c in g.cfg["cases"]:
        s = c["set"]
        key = f"{s}/out{c['idx']}"
        want = unragged(g[key + "_flat"], g[key + "_lens"])
        lm = O.toy_language_model if c["lm"] else None
        got = O.ctc_beam_decode(g[f"{s}/x"], g[f"{s}/lens"], c["blank"], c["beam_width"], c["prune"], lm,
                                c.get("lm_weight"), c["sep"], c["word_weight"])
        assert got == want, c


@pytest.mark.parametrize("name", golden_names("stream_"))
def test_oracle_chunked_streaming(name):
    """The streaming definition (hx threaded through consecutive slices) restated on the oracle."""
    g = Golden(name)
    c = g.cfg
    cfg = dict(convs=[dict(kind=v["kind"], idx=v["idx"], stride=v["stride"], same=v["same"], act=v["act"])
                      for v in c["convs"]],
               rnn=dict(kind=c["rnn"]["kind"], hidden=c["rnn"]["hidden"], layers=c["rnn"]["layers"],
                        bidirectional=c["rnn"]["bidirectional"]),
               lookahead=None, fc=dict(n_hidden=c["fc"]["n_hidden"], act=c["fc"]["act"]))
    x, lens, chunk = g["in/x"], g["in/lens"], c["chunk_frames"]
    outs, hid, t0 = [], None, 0
    while t0 < x.shape[-1]:
        alive = int((lens > t0).sum())
        if alive == 0:
            break
        xc = x[:alive, :, :, t0:t0 + chunk]
        lc = np.minimum(lens[:alive] - t0, xc.shape[-1])
        hx = None if hid is None else (hid[0][:, :alive], hid[1][:, :alive])
        y, _, hid = O.deep_speech_2_forward(xc, lc, cfg, g.sd(), hx)
        full = np.zeros((y.shape[0], x.shape[0], y.shape[2]), np.float32)
        full[:, :alive] = y
        outs.append(full)
        t0 += chunk
    np.testing.assert_allclose(np.concatenate(outs, 0), g["out/y"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hid[0], g["out/hn_last"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["ctc_grad_small", "ctc_grad_v29"])
def test_ctc_grad_oracle_matches_reference_autograd(name):
    """The alpha-beta gradient restatement against x.grad obtained through the reference's CTCLoss module."""
    g = Golden(name)
    x, xl, y, yl, w = g["in/x"], g["in/x_lens"], g["in/y"], g["in/y_lens"], g["in/w"]
    n = x.shape[1]
    for key in [k for k in g.a if k.startswith("grad/")]:
        red, zi = key[len("grad/"):].rsplit("_", 1)
        if red == "none":
            gn = w
        elif red == "sum":
            gn = np.full(n, g.cfg["scale"], np.float32)
        else:
            gn = g.cfg["scale"] / (np.maximum(yl.astype(np.float32), 1.0) * n)
        got = O.ctc_grad(x, xl, y, yl, gn, g.cfg["blank"], bool(int(zi)))
        np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=2e-5)  # the reference subtracts two float32 exponentials


def test_ctc_grad_dim_oracle_matches_reference_autograd():
    """``CTCLoss(dim != -1)`` under autograd (ctc_loss.py:37-45, 95-101): LogSoftmax(dim)'s backward chained behind
    torch.nn.CTCLoss's; x.grad and the losses from the reference."""
    g = Golden("ctc_grad_dim")
    x, xl, y, yl, w = g["in/x"], g["in/x_lens"], g["in/y"], g["in/y_lens"], g["in/w"]
    n = x.shape[1]
    for key in [k for k in g.a if k.startswith("grad/")]:
        dim, red, zi = key[len("grad/dim"):].split("_")
        gn = w if red == "none" else (np.full(n, g.cfg["scale"], np.float32) if red == "sum" else
                                      g.cfg["scale"] / (np.maximum(yl.astype(np.float32), 1.0) * n))
        got = O.ctc_grad(x, xl, y, yl, gn, g.cfg["blank"], bool(int(zi)), dim=int(dim))
        np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(O.ctc_loss(x, xl, y, yl, g.cfg["blank"], red, bool(int(zi)), dim=int(dim)),
                                   g["out/" + key[len("grad/"):]], rtol=1e-4, atol=1e-4)


def test_wer_fixture_pins_levenshtein_alphabet_and_wer():
    """SURVEY 8 f2 against vectors made by the reference's own ``levenshtein`` (post_process/utils.py:4-60) and ``Alphabet``
    (data/alphabet.py:5-78): the oracle's restatement, the product's ``post_process.utils.levenshtein``, ``data.alphabet.
    Alphabet``, ``wer.WordSegmentor`` and ``wer.WordErrorRate`` (run/run.py:84-109's arithmetic)."""
    from myrtlespeech_amd.data.alphabet import Alphabet
    from myrtlespeech_amd.post_process.utils import levenshtein
    from myrtlespeech_amd.wer import WordErrorRate, WordSegmentor
    g = Golden("wer")
    c = g.cfg
    hyps = unragged(g["in/hyp_flat"], g["in/hyp_lens"])
    tgts = unragged(g["in/tgt_flat"], g["in/tgt_lens"])
    rts = unragged(g["out/roundtrip_flat"], g["out/roundtrip_lens"])
    alpha = Alphabet(c["symbols"])
    seg = WordSegmentor(c["separator"])
    assert len(alpha) == int(g["out/len"][0]) and alpha.get_index("?") is None and ord(alpha[5]) == int(g["out/symbol_5"][0])
    with pytest.raises(IndexError):
        alpha[len(alpha)]
    with pytest.raises(ValueError):
        Alphabet(["a", "b", "a"])
    for i, (hyp, tgt) in enumerate(zip(hyps, tgts)):
        hs, ts = alpha.get_symbols(hyp), alpha.get_symbols(tgt)
        assert "".join(hs) == c["hyp_text"][i] and "".join(ts) == c["tgt_text"][i]
        assert alpha.get_indices(hs + ["?", "ab"]) == rts[i]
        hw, tw = seg(hs), seg(ts)
        for fn in (levenshtein, O.levenshtein):
            assert fn(hyp, tgt) == int(g["out/idx_dist"][i])
            assert fn(hs, ts) == int(g["out/sym_dist"][i])
            assert fn(hw, tw) == int(g["out/word_dist"][i])
        assert len(tw) == int(g["out/word_len"][i])
    wer = WordErrorRate(alpha, seg)
    pad = max(len(t) for t in tgts)
    tgt_pad = np.zeros((len(tgts), pad), np.int64)
    for i, t in enumerate(tgts):
        tgt_pad[i, :len(t)] = t
    half = len(hyps) // 2          # two "batches", as run.py:84-103 accumulates over an epoch
    wer.update(hyps[:half], tgt_pad[:half], [len(t) for t in tgts[:half]])
    wer.update(hyps[half:], tgt_pad[half:], [len(t) for t in tgts[half:]])
    assert wer.distances == [int(v) for v in g["out/word_dist"]] and wer.lengths == [int(v) for v in g["out/word_len"]]
    assert wer.value() == c["wer"]


def test_beam_oracle_config_size_vs_reference():
    """The numpy beam search against the reference at T = 501, V = 29, beam 8 (one of the four utterances; the pure-Python
    oracle takes ~1.5 s per utterance like the reference)."""
    g = Golden("beam_cfg2")
    c = g.cfg
    torch.manual_seed(c["seed"])
    x = torch.softmax(torch.randn(c["T"], c["N"], c["V"]) * c["scale"], dim=2).numpy()
    lens = g["in/lens"]
    want = unragged(g["out/plain_flat"], g["out/plain_lens"])
    got = O.ctc_beam_decode(x[:, 2:3], lens[2:3], 28, c["beam_width"], c["prune"])
    assert got == want[2:3]


def test_oracles_vs_the_trained_scale_config2_fixture():
    """Both CPU oracles (the numpy restatement and the stock-torch re-assembly that is bench.py's cpu_baseline) on two
    utterances of the TRAINED-SCALE config-2 fixture (tests/golden/ds2_cfg2_trained_summary.npz, the reference with
    weight_ih x 16 / weight_hh x 2 / FC x 6: saturated gates, logits of O(1 .. 10)): logits on the stored sub-grid within
    1e-3 absolute (the reference's own float32 rounding is 5e-4 there), every frame's arg max and the greedy transcripts of
    those utterances equal, the CTC loss of the fixture's targets within 1e-4, the prefix beam search on the ORACLE's own
    posteriors of utterance 0 equal to the reference's transcript.  Weights are regenerated from seed 0 (checksums pinned)."""
    import bench
    from oracle import torch_cpu as TC
    from util import apply_trained_gains
    g = Golden("ds2_cfg2_trained_summary")
    model = bench.build_model()
    with torch.no_grad():
        apply_trained_gains(model, g.cfg["gains"])
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        want = g.cfg["weight_abs_sums"][k]
        assert abs(float(np.abs(v.astype(np.float64)).sum()) - want) <= 1e-6 * max(1.0, want), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    x = torch.randn(g.cfg["N"], 1, 80, g.cfg["T"], generator=gen)
    lens = torch.sort(torch.randint(501, 1002, (g.cfg["N"],), generator=gen), descending=True).values
    lens[0] = g.cfg["T"]
    cfg = dict(convs=[dict(kind="conv2d", idx=0, stride=(2, 2), same=True, act=(0.0, 20.0)),
                      dict(kind="conv2d", idx=2, stride=(2, 1), same=True, act=(0.0, 20.0))],
               rnn=dict(kind=O.LSTM, hidden=1024, layers=5, bidirectional=True), lookahead=None,
               fc=dict(n_hidden=1, act=(0.0, 20.0)))
    sel = [0, 4]                                           # columns 0 and 1 of the stored [::25, ::4] sub-grid
    want_dec = unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    for name, fwd in (("numpy", lambda: O.deep_speech_2_forward(x[sel].numpy(), lens[sel].numpy(), cfg, sd)[:2]),
                      ("torch_cpu", lambda: TC.deep_speech_2_forward(x[sel].numpy(), lens[sel].numpy(), cfg, sd))):
        y, ol = fwd()
        np.testing.assert_array_equal(ol, g["out/lens"][sel])
        np.testing.assert_allclose(y[::25], g["out/y_sub"][:, :2], rtol=0, atol=1e-3, err_msg=name)
        for j, n in enumerate(sel):
            assert np.array_equal(y[:ol[j], j].argmax(-1), g["out/argmax"][:ol[j], n].astype(np.int64)), name
        assert O.ctc_greedy_decode(y, ol, 28) == [want_dec[n] for n in sel], name
        loss = O.ctc_loss(y, ol, g["ctc/y"][sel], g["ctc/y_lens"][sel], blank=28, reduction="none")
        np.testing.assert_allclose(loss, g["ctc/none"][sel], rtol=1e-4, err_msg=name)
    # the reference beam search on the encoder's own posteriors: utterance 0 over its first 140 frames (beam/* entry 0)
    e = np.exp(y[:, 0] - y[:, 0].max(-1, keepdims=True))
    probs = (e / e.sum(-1, keepdims=True)).astype(np.float32)[:, None, :]
    assert int(g["beam/utts"][0]) == 0
    got = O.ctc_beam_decode(probs, np.array([int(g["beam/lens"][0])]), 28, 8, prune_threshold=0.001)
    assert got == unragged(g["beam/flat"], g["beam/out_lens"])[:1]
