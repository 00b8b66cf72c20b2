"""Front-end on the MI355X: every device step through the C ABI against the numpy oracle
(oracle/frontend_oracle.py) -- per-sample calls, ragged batches (must equal per-sample processing + zero padding),
seeded SpecAugment (bit-exact), and the pre-processing pipelines of the shipped config shapes end to end."""
import random

import numpy as np
import pytest
import torch

from oracle import frontend_oracle as FO
from tests.util import Golden

pytestmark = pytest.mark.gpu

from myrtlespeech_amd.data.preprocess import (AddContextFrames, MFCC, MFCCLegacy, SpecAugment,  # noqa: E402
                                              Standardize)


def cpu(t):
    return t.detach().cpu().numpy()


def ragged_waves(rng, lens, scale=0.1):
    L = max(lens)
    w = np.zeros((len(lens), L), np.float32)
    for i, l in enumerate(lens):
        w[i, :l] = (rng.standard_normal(l) * scale).astype(np.float32)
    return w


# ---- AddContextFrames --------------------------------------------------------------------------
def test_context_frames_reference_docstring_vector():
    g = Golden("context_frames_doc")
    y = AddContextFrames(g.cfg["n_context"])(torch.from_numpy(g["in/x"]))
    assert y.dtype == torch.int64 and np.array_equal(cpu(y), g["out/y"])
    assert repr(AddContextFrames(2)) == "AddContextFrames(n_context=2)"


@pytest.mark.parametrize("F,T,c", [(26, 201, 9), (3, 1, 2), (80, 37, 0), (5, 4, 7)])
def test_context_frames_per_sample_bit_exact(F, T, c):
    x = np.random.default_rng(F * T + c).standard_normal((1, F, T)).astype(np.float32)
    assert np.array_equal(cpu(AddContextFrames(c)(torch.from_numpy(x))), FO.add_context_frames(x, c))


def test_context_frames_ragged_batch_equals_per_sample_then_pad():
    rng = np.random.default_rng(0)
    lens = [50, 33, 33, 7, 1]
    F, c = 13, 4
    xs = [rng.standard_normal((1, F, l)).astype(np.float32) for l in lens]
    x = FO.pad_sequence(xs)                                              # [N, 1, F, T]
    y, out_lens = AddContextFrames(c).batch(torch.from_numpy(x), torch.tensor(lens))
    want = FO.pad_sequence([FO.add_context_frames(xi, c) for xi in xs])
    assert np.array_equal(cpu(y), want) and out_lens.tolist() == lens


# ---- Standardize -------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 80, 1001), (7,), (3, 5, 11), (2, 2)])
def test_standardize_per_sample(shape):
    x = (np.random.default_rng(len(shape)).standard_normal(shape) * 5 + 3).astype(np.float32)
    y = cpu(Standardize()(torch.from_numpy(x)))
    np.testing.assert_allclose(y, FO.standardize(x), rtol=1e-5, atol=1e-5)
    t = torch.from_numpy(x)
    np.testing.assert_allclose(y, ((t - t.mean()) / t.std()).numpy(), rtol=1e-5, atol=1e-5)


def test_standardize_reference_doctest_property_10m():
    """data/preprocess.py:46-54 at its documented size (10 M elements)."""
    x = 5 * torch.empty(10_000_000).normal_(generator=torch.Generator().manual_seed(0)) + 3
    y = Standardize()(x)
    assert -0.001 <= float(y.mean()) <= 0.001 and 0.999 <= float(y.std()) <= 1.001


def test_standardize_ragged_batch_equals_per_sample_then_pad():
    rng = np.random.default_rng(1)
    lens = [120, 77, 76, 3, 2]
    xs = [(rng.standard_normal((1, 20, l)) * (i + 1) - i).astype(np.float32) for i, l in enumerate(lens)]
    x = FO.pad_sequence(xs)
    y, _ = Standardize().batch(torch.from_numpy(x), torch.tensor(lens))
    want = FO.pad_sequence([FO.standardize(xi) for xi in xs])
    np.testing.assert_allclose(cpu(y), want, rtol=1e-5, atol=1e-5)
    assert np.all(cpu(y)[1, :, :, 77:] == 0)


def test_standardize_single_element_is_nan_like_torch():
    assert np.isnan(cpu(Standardize()(torch.tensor([3.0]))))[0]


# ---- SpecAugment -------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_spec_augment_seeded_equals_oracle_bit_exact(seed):
    rng = np.random.default_rng(seed)
    C, F, T = 1 + seed % 2, 26, 180
    x = rng.standard_normal((C, F, T)).astype(np.float32)
    want = FO.spec_augment(x.copy(), 3, 20, 2, 2, random.Random(seed))
    random.seed(seed)
    xd = torch.from_numpy(x).cuda()
    out = SpecAugment(3, 20, 2, 2)(xd)
    assert out is xd and np.array_equal(cpu(xd), want)


def test_spec_augment_bound_and_shape_restated():
    """tests/data/test_preprocess.py:52-87 on seeded cases."""
    r = random.Random(5)
    for _ in range(20):
        C, F, T = r.randint(1, 3), r.randint(1, 100), r.randint(1, 100)
        sa = SpecAugment(r.randint(0, 30), r.randint(0, 30), r.randint(0, 3), r.randint(0, 3))
        out = sa(torch.ones(C, F, T, device="cuda"))
        assert out.shape == (C, F, T)
        assert int((out == 0).sum()) <= C * (sa.n_feature_masks * sa.feature_mask * T + sa.n_time_masks * sa.time_mask * F)


def test_spec_augment_ragged_batch_draws_per_utterance_in_order():
    rng = np.random.default_rng(9)
    lens = [90, 60, 25]
    xs = [rng.standard_normal((1, 26, l)).astype(np.float32) for l in lens]
    r = random.Random(21)
    want = FO.pad_sequence([FO.spec_augment(xi.copy(), 3, 20, 2, 2, r) for xi in xs])
    random.seed(21)
    xd = torch.from_numpy(FO.pad_sequence(xs)).cuda()
    SpecAugment(3, 20, 2, 2).batch(xd, torch.tensor(lens))
    assert np.array_equal(cpu(xd), want)


# ---- MFCC (torchaudio 0.4.0 restated) ----------------------------------------------------------
def mfcc_tol(want):
    # dB-domain cepstra: |c| up to ~1e3; f32 DFT/mel/DCT contractions give ~1e-4 absolute
    return dict(rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("n_mfcc,win,hop,L", [(80, 400, 160, 16000), (26, 400, 320, 64000), (13, 320, 100, 3217),
                                              (40, 400, 200, 201)])
def test_mfcc_per_sample_matches_oracle(n_mfcc, win, hop, L):
    w = (np.random.default_rng(L).standard_normal((1, L)) * 0.1).astype(np.float32)
    got = cpu(MFCC(n_mfcc=n_mfcc, melkwargs={"win_length": win, "hop_length": hop})(torch.from_numpy(w)))
    want = FO.mfcc(w, n_mfcc, win, hop)
    assert got.shape == want.shape == (1, n_mfcc, 1 + L // hop)
    np.testing.assert_allclose(got, want, **mfcc_tol(want))


def test_mfcc_silence_hits_the_top_db_floor():
    w = (np.random.default_rng(3).standard_normal((1, 16000)) * 0.05).astype(np.float32)
    w[0, 4000:9000] = 0.0
    got = cpu(MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160})(torch.from_numpy(w)))
    np.testing.assert_allclose(got, FO.mfcc(w, 80, 400, 160), **mfcc_tol(None))


def test_mfcc_ragged_batch_equals_per_sample_then_pad():
    rng = np.random.default_rng(4)
    lens = [8000, 6400, 6399, 1234, 201]
    w = ragged_waves(rng, lens)
    m = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160})
    y, fl = m.batch(torch.from_numpy(w), torch.tensor(lens))
    assert fl.tolist() == [1 + l // 160 for l in lens] and y.shape == (5, 1, 80, 51)
    want = FO.pad_sequence([FO.mfcc(w[i:i + 1, :l], 80, 400, 160) for i, l in enumerate(lens)])
    np.testing.assert_allclose(cpu(y), want, **mfcc_tol(None))
    for i, f in enumerate(fl.tolist()):
        assert np.all(cpu(y)[i, :, :, f:] == 0)


def test_mfcc_rejects_waveforms_shorter_than_the_reflect_pad():
    m = MFCC(n_mfcc=20)
    with pytest.raises(ValueError):
        m(torch.randn(1, 200))
    with pytest.raises(ValueError):
        m.batch(torch.randn(2, 4000), torch.tensor([4000, 4001]))


def test_mfcc_full_size_batch_properties():
    """BASELINE configs[1] front-end: 32 x 10 s @ 16 kHz -> [32, 1, 80, 1001]; rows of identical audio agree exactly,
    and a clip's features do not depend on what it is batched with."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(32, 160000, generator=g) * 0.1
    w[5] = w[0]
    m = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160})
    y, fl = m.batch(w, torch.full((32,), 160000))
    assert y.shape == (32, 1, 80, 1001) and fl.tolist() == [1001] * 32
    assert torch.equal(y[0], y[5]) and bool(torch.isfinite(y).all())
    solo = m(w[7:8])
    assert torch.equal(solo, y[7])
    s, _ = Standardize().batch(y, fl)
    per = s.view(32, -1)
    assert float(per.mean(1).abs().max()) < 1e-4 and float((per.std(1) - 1).abs().max()) < 1e-4


# ---- MFCCLegacy (python_speech_features 0.6 restated, float64) ---------------------------------
@pytest.mark.parametrize("n_mfcc,win,hop,L", [(26, 400, 320, 64000), (13, 400, 160, 16000), (26, 400, 320, 300),
                                              (20, 512, 256, 5000)])
def test_mfcc_legacy_matches_oracle(n_mfcc, win, hop, L):
    w = np.clip(np.random.default_rng(L + n_mfcc).standard_normal((1, L)) * 0.3, -1, 1).astype(np.float32)
    got = cpu(MFCCLegacy(n_mfcc, {"win_length": win, "hop_length": hop})(torch.from_numpy(w)))
    want = FO.mfcc_legacy(w, n_mfcc, win, hop)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_mfcc_legacy_ragged_batch_and_silence():
    rng = np.random.default_rng(8)
    lens = [9000, 5000, 4999, 400, 17]
    w = ragged_waves(rng, lens, 0.3).clip(-1, 1)
    w[1, 1000:3000] = 0.0                                              # all-zero frames -> eps branches
    m = MFCCLegacy(26, {"win_length": 400, "hop_length": 320})
    y, fl = m.batch(torch.from_numpy(w), torch.tensor(lens))
    want = FO.pad_sequence([FO.mfcc_legacy(w[i:i + 1, :l], 26, 400, 320) for i, l in enumerate(lens)])
    assert fl.tolist() == [m.frames(l) for l in lens] and y.shape[-1] == want.shape[-1]
    np.testing.assert_allclose(cpu(y), want, rtol=1e-5, atol=1e-5)


# ---- whole pre-processing pipelines, shipped config shapes -------------------------------------
def test_ds1_pipeline_eval_and_train_modes():
    """configs/deep_speech_1_en.config:4-28: mfcc(26, 400, 320) -> [TRAIN: spec_augment] -> context_frames(9)."""
    from myrtlespeech_amd import protos as P
    from myrtlespeech_amd.builders.speech_to_text import build as build_stt
    from tests.test_builders_cpu import DS1_EN
    stt = build_stt(P.parse(DS1_EN, P.SpeechToText))
    rng = np.random.default_rng(12)
    lens = [64000, 40000, 12345]
    w = ragged_waves(rng, lens)
    stt.eval()
    x, fl = stt.pre_process_batch(torch.from_numpy(w), torch.tensor(lens))
    want = FO.pad_sequence([FO.add_context_frames(FO.mfcc(w[i:i + 1, :l], 26, 400, 320), 9) for i, l in enumerate(lens)])
    assert x.shape == (3, 19, 26, 201) and fl.tolist() == [201, 126, 39]
    np.testing.assert_allclose(cpu(x), want, **mfcc_tol(None))
    # per-sample path == batched path
    one = stt.pre_process(torch.from_numpy(w[1:2, :40000]))
    assert torch.equal(one, x[1, :, :, :126])
    # TRAIN adds the seeded masks between the two
    stt.train()
    random.seed(33)
    xt, _ = stt.pre_process_batch(torch.from_numpy(w), torch.tensor(lens))
    r = random.Random(33)
    want_t = FO.pad_sequence([FO.add_context_frames(FO.spec_augment(FO.mfcc(w[i:i + 1, :l], 26, 400, 320), 3, 20, 2, 2, r), 9)
                              for i, l in enumerate(lens)])
    np.testing.assert_allclose(cpu(xt), want_t, **mfcc_tol(None))
    assert np.array_equal(cpu(xt) == 0, np.abs(want_t) == 0)
    # and the encoder consumes it
    stt.eval()
    (logits, out_lens), _ = stt.model((x, fl))
    assert logits.shape == (201, 3, 29) and out_lens.tolist() == fl.tolist()


def test_ds2_pipeline_feeds_the_encoder():
    """configs/deep_speech_2_en.config:4-17: mfcc(80, 400, 160) -> standardize, then a DS2 stack and greedy decode."""
    from myrtlespeech_amd import protos as P
    from myrtlespeech_amd.builders.speech_to_text import build as build_stt
    cfg = P.parse('''
    alphabet: " abcdefghijklmnopqrstuvwxyz'_";
    pre_process_step { stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 80; win_length: 400; hop_length: 160; } }
    pre_process_step { stage: TRAIN_AND_EVAL; standardize { } }
    deep_speech_2 {
      conv_block { conv2d { output_channels: 32; kernel_feature: 41; kernel_time: 11; stride_feature: 2; stride_time: 2;
                            padding_mode: SAME; bias: true; } activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
      conv_block { conv2d { output_channels: 32; kernel_feature: 21; kernel_time: 11; stride_feature: 2; stride_time: 1;
                            padding_mode: SAME; bias: true; } activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
      rnn { rnn_type: LSTM; hidden_size: 256; num_layers: 2; bias: true; bidirectional: true; forget_gate_bias { value: 1.0 } }
      lookahead_block { no_lookahead {} activation { identity {} } }
      fully_connected { num_hidden_layers: 1; hidden_size: 128; activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
    }
    ctc_loss { blank_index: 28; reduction: SUM; }
    ctc_greedy_decoder { blank_index: 28; }
    ''', P.SpeechToText)
    torch.manual_seed(0)
    stt = build_stt(cfg).eval()
    rng = np.random.default_rng(13)
    lens = [32000, 24000, 8000]
    w = ragged_waves(rng, lens)
    x, fl = stt.pre_process_batch(torch.from_numpy(w), torch.tensor(lens))
    want = FO.pad_sequence([FO.standardize(FO.mfcc(w[i:i + 1, :l], 80, 400, 160)) for i, l in enumerate(lens)])
    assert x.shape == (3, 1, 80, 201)
    np.testing.assert_allclose(cpu(x), want, rtol=1e-4, atol=1e-4)
    # encoder on the device features vs the oracle encoder on the oracle features
    from oracle import ds_oracle as O
    sd = {k: v.detach().cpu().numpy() for k, v in stt.model.state_dict().items()}
    ocfg = dict(convs=[dict(kind="conv2d", idx=0, stride=(2, 2), same=True, act=(0.0, 20.0)),
                       dict(kind="conv2d", idx=2, stride=(2, 1), same=True, act=(0.0, 20.0))],
                rnn=dict(kind=O.LSTM, hidden=256, layers=2, bidirectional=True), lookahead=None,
                fc=dict(n_hidden=1, act=(0.0, 20.0)))
    want_y, want_l, _ = O.deep_speech_2_forward(want.copy(), np.array(fl.tolist()), ocfg, sd)
    (y, out_lens), _ = stt.model((x, fl))
    assert out_lens.tolist() == list(want_l)
    np.testing.assert_allclose(cpu(y), want_y, rtol=1e-3, atol=1e-3)
    assert stt.post_process(y, out_lens) == O.ctc_greedy_decode(want_y, want_l, 28)


def test_transcribe_waveforms_to_text():
    """SpeechToText.transcribe = pre_process_batch -> encoder -> decoder -> alphabet, nothing else."""
    from myrtlespeech_amd import protos as P
    from myrtlespeech_amd.builders.speech_to_text import build as build_stt
    from tests.test_checkpoint_cpu import DS2_TINY
    torch.manual_seed(3)
    stt = build_stt(P.parse(DS2_TINY, P.SpeechToText)).eval()
    rng = np.random.default_rng(14)
    lens = [16000, 9000, 4000]
    w = torch.from_numpy(ragged_waves(rng, lens))
    text, labels = stt.transcribe(w, torch.tensor(lens))
    x, fl = stt.pre_process_batch(w, torch.tensor(lens))
    (y, ol), _ = stt.model((x, fl))
    assert labels == stt.post_process(y, ol)
    assert text == ["".join(stt.alphabet.get_symbols(s)) for s in labels] and len(text) == 3
