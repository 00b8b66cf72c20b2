"""Front-end, CPU side: the oracle against the reference's fixtures and documented vectors, the
third-party stages against the libraries that ARE installed (torch.stft, scipy DCT), and the host logic
(collate, builder, tables, SpecAugment draw order).  No GPU, no compute calls into the library."""
import random

import numpy as np
import pytest
import torch

from oracle import frontend_oracle as FO
from tests.util import Golden

from myrtlespeech_amd import protos
from myrtlespeech_amd.builders.pre_process_step import build as build_step
from myrtlespeech_amd.data import batch as B
from myrtlespeech_amd.data.preprocess import (AddContextFrames, AddSequenceLength, MFCC, MFCCLegacy, SpecAugment,
                                              Standardize)
from myrtlespeech_amd.stage import Stage


# ---- oracle pins ---------------------------------------------------------------------------------
def test_oracle_context_frames_matches_reference_docstring_vector():
    g = Golden("context_frames_doc")
    assert np.array_equal(FO.add_context_frames(g["in/x"], g.cfg["n_context"]), g["out/y"])


def test_oracle_collate_matches_reference_fixture():
    g = Golden("collate")
    batch = [((g[f"in/x{i}"], np.array(g[f"in/x{i}"].shape[-1])), (g[f"in/y{i}"], np.array(g[f"in/y{i}"].shape[-1])))
             for i in range(g.cfg["n"])]
    (x, xl), (y, yl) = FO.seq_to_seq_collate(batch)
    assert np.array_equal(x, g["out/x"]) and np.array_equal(xl, g["out/x_lens"])
    assert np.array_equal(y, g["out/y"]) and np.array_equal(yl, g["out/y_lens"])
    assert np.array_equal(FO.pad_sequence([b[0][0] for b in batch], 7), g["out/pad7"])


def test_oracle_standardize_doctest_property():
    """data/preprocess.py:46-54: a shifted, scaled normal comes back with mean ~0 and std ~1."""
    x = (5 * np.random.default_rng(0).standard_normal(1_000_000) + 3).astype(np.float32)
    y = FO.standardize(x)
    assert abs(float(y.mean())) <= 1e-3 and 0.999 <= float(y.std(ddof=1)) <= 1.001
    t = torch.from_numpy(x)
    assert np.allclose(y, ((t - t.mean()) / t.std()).numpy(), atol=2e-6)


def test_oracle_spec_augment_bound():
    """tests/data/test_preprocess.py:67-87: zeros <= channels*(m_F*F*steps + m_T*T*features)."""
    rng = random.Random(3)
    for _ in range(50):
        C, F, T = rng.randint(1, 3), rng.randint(1, 40), rng.randint(1, 60)
        fm, tm, nf, nt = rng.randint(0, 30), rng.randint(0, 30), rng.randint(0, 3), rng.randint(0, 3)
        out = FO.spec_augment(np.ones((C, F, T), np.float32), fm, tm, nf, nt, rng)
        assert out.shape == (C, F, T)
        assert (out == 0).sum() <= C * (nf * fm * T + nt * tm * F)


@pytest.mark.parametrize("hop,win", [(160, 400), (320, 400), (100, 320)])
def test_oracle_stft_stage_matches_installed_torch(hop, win):
    w = (np.random.default_rng(1).standard_normal(5000) * 0.1).astype(np.float32)
    s = torch.stft(torch.from_numpy(w), 400, hop, win, torch.hann_window(win), center=True, pad_mode="reflect",
                   normalized=False, onesided=True, return_complex=True)
    ref = (s.real ** 2 + s.imag ** 2).numpy()
    got = FO.stft_power(w, 400, hop, win)
    assert got.shape == ref.shape == (201, 1 + 5000 // hop)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_oracle_dct_matches_scipy():
    import scipy.fftpack
    x = np.random.default_rng(2).standard_normal((9, 128)).astype(np.float64)
    assert np.allclose(x @ FO.create_dct(80, 128), scipy.fftpack.dct(x, type=2, norm="ortho", axis=1)[:, :80], atol=1e-4)  # table built in f32 like torchaudio


def test_oracle_mel_filterbank_shape_and_partition():
    fb = FO.mel_filterbank(201, 0.0, 8000.0, 128)
    assert fb.shape == (201, 128) and fb.min() >= 0 and fb.max() <= 1.0
    # neighbouring unnormalised triangles sum to one between the first and last centre frequency
    inner = fb.sum(1)[3:-6]
    assert np.allclose(inner, 1.0, atol=1e-3)


def test_oracle_mfcc_shapes_and_top_db_floor():
    w = (np.random.default_rng(3).standard_normal(16000) * 0.05).astype(np.float32)
    w[4000:9000] = 0.0            # silence: mel power 0 -> -100 dB, floored at max - 80
    power = FO.stft_power(w, 400, 160, 400)
    mel = power.T @ FO.mel_filterbank(201, 0.0, 8000.0, 128)
    db = FO.amplitude_to_db(mel.astype(np.float32))
    assert np.isclose(db.min(), db.max() - 80.0)
    assert FO.mfcc(w[None], 80, 400, 160).shape == (1, 80, 101)
    assert FO.mfcc(w[None], 26, 400, 320).shape == (1, 26, 51)


def test_oracle_mfcc_legacy_shape_energy_and_lifter():
    w = np.clip(np.random.default_rng(4).standard_normal(64000) * 0.2, -1, 1).astype(np.float32)
    out = FO.mfcc_legacy(w[None], 26, 400, 320)
    assert out.shape == (1, 26, 200) and out.dtype == np.float32      # 1 + ceil((64000 - 400) / 320)
    # c0 is the log of the frame's total power-spectrum energy (Parseval: sum of squares of the pre-emphasised frame,
    # the one-sided spectrum counts interior bins once)
    x = (w * np.float32(32768)).astype(np.int16).astype(np.float64)
    sig = np.append(x[0], x[1:] - 0.97 * x[:-1])
    spec = np.abs(np.fft.rfft(sig[:400], 512)) ** 2 / 512
    assert np.isclose(out[0, 0, 0], np.log(spec.sum()), rtol=1e-6)
    assert FO.mfcc_legacy(w[None, :300], 13, 400, 160).shape == (1, 13, 1)


# ---- host logic of the product -------------------------------------------------------------------
def test_collate_matches_reference_fixture():
    g = Golden("collate")
    batch = [((torch.from_numpy(g[f"in/x{i}"]), torch.tensor(g[f"in/x{i}"].shape[-1])),
              (torch.from_numpy(g[f"in/y{i}"]), torch.tensor(g[f"in/y{i}"].shape[-1]))) for i in range(g.cfg["n"])]
    (x, xl), (y, yl) = B.seq_to_seq_collate_fn(batch)
    assert torch.equal(x, torch.from_numpy(g["out/x"])) and torch.equal(xl, torch.from_numpy(g["out/x_lens"]))
    assert torch.equal(y, torch.from_numpy(g["out/y"])) and torch.equal(yl, torch.from_numpy(g["out/y_lens"]))
    assert torch.equal(B.pad_sequence([b[0][0] for b in batch], 7), torch.from_numpy(g["out/pad7"]))
    assert xl.tolist() == sorted(xl.tolist(), reverse=True)


def test_pad_sequence_size_and_values():
    """tests/data/test_batch.py:60-100 restated on seeded cases, several dtypes."""
    rng = random.Random(0)
    for dtype in (torch.float32, torch.int64, torch.float16, torch.uint8):
        lead = tuple(rng.randint(1, 4) for _ in range(rng.randint(0, 2)))
        seqs = [torch.randint(0, 100, lead + (rng.randint(1, 12),)).to(dtype) for _ in range(rng.randint(1, 6))]
        pad = rng.randint(0, 127)
        out = B.pad_sequence(seqs, pad)
        assert out.shape == (len(seqs),) + lead + (max(s.size(-1) for s in seqs),) and out.dtype == dtype
        for i, s in enumerate(seqs):
            assert torch.all(out[i, ..., :s.size(-1)].float() == s.float())
            assert torch.all(out[i, ..., s.size(-1):].float() == pad)


def test_add_sequence_length():
    x = torch.rand(5, 10, 3)
    for dim in range(3):
        out, n = AddSequenceLength(dim)(x)
        assert out is x and torch.equal(n, torch.tensor([x.size(dim)]))
    assert repr(AddSequenceLength(1)) == "AddSequenceLength(length_dim=1)"


def test_spec_augment_validation_and_draw_order():
    for kw in (dict(feature_mask=-1, time_mask=1), dict(feature_mask=1, time_mask=-1),
               dict(feature_mask=1, time_mask=1, n_feature_masks=-1), dict(feature_mask=1, time_mask=1, n_time_masks=-1)):
        with pytest.raises(ValueError):
            SpecAugment(**kw)
    sa = SpecAugment(3, 20, 2, 2)
    random.seed(11)
    got = sa._draw(26, 150)
    want = FO.spec_augment_bands(26, 150, 3, 20, 2, 2, random.Random(11))
    assert got == want


def _step(text):
    return protos.parse(text, protos.PreProcessStep)


def test_builder_maps_every_step_type():
    """tests/builders/test_pre_process_step.py:20-58 restated."""
    s, st = build_step(_step("stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 80; win_length: 400; hop_length: 160; }"))
    assert isinstance(s, MFCC) and st == Stage.TRAIN_AND_EVAL and s.n_mfcc == 80
    assert s.MelSpectrogram.win_length == 400 and s.MelSpectrogram.hop_length == 160
    s, st = build_step(_step("stage: EVAL; mfcc { n_mfcc: 13; win_length: 400; hop_length: 320; legacy: true; }"))
    assert isinstance(s, MFCCLegacy) and st == Stage.EVAL
    assert (s.numcep, s.samplerate, s.winlen, s.winstep) == (13, 16000, 400 / 16000, 320 / 16000)
    s, st = build_step(_step("stage: TRAIN; spec_augment { feature_mask: 3; time_mask: 20; n_feature_masks: 2; n_time_masks: 2; }"))
    assert isinstance(s, SpecAugment) and st == Stage.TRAIN and (s.feature_mask, s.time_mask) == (3, 20)
    s, _ = build_step(_step("stage: TRAIN_AND_EVAL; standardize { }"))
    assert isinstance(s, Standardize)
    s, _ = build_step(_step("stage: TRAIN_AND_EVAL; context_frames { n_context: 9; }"))
    assert isinstance(s, AddContextFrames) and s.n_context == 9
    with pytest.raises(ValueError):
        build_step(_step("stage: TRAIN;"))


def test_mfcc_constructor_errors_and_tables():
    with pytest.raises(ValueError):
        MFCC(n_mfcc=129)
    with pytest.raises(ValueError):
        MFCC(n_mfcc=40, melkwargs={"win_length": 401, "hop_length": 160})
    with pytest.raises(ValueError):
        MFCCLegacy(27, {"win_length": 400, "hop_length": 160})
    m = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160})
    window, dft, fb, dct = m.tables()
    assert np.allclose(window.numpy(), FO.hann_window(400), atol=1e-6)
    assert np.allclose(fb.numpy().T, FO.mel_filterbank(201, 0.0, 8000.0, 128), atol=1e-4)
    assert np.allclose(dct.numpy().T, FO.create_dct(80, 128), atol=1e-6)
    # the DFT table reproduces numpy's rfft
    x = np.random.default_rng(5).standard_normal(400)
    spec = dft.numpy().astype(np.float64) @ x
    ref = np.fft.rfft(x)
    assert np.allclose(spec[:201], ref.real, atol=1e-4) and np.allclose(spec[201:], ref.imag, atol=1e-4)
    assert m.frames(160000) == 1001
    leg = MFCCLegacy(26, {"win_length": 400, "hop_length": 320})
    assert leg.geometry() == (400, 320, 512) and leg.frames(64000) == 200 and leg.frames(300) == 1
    assert np.array_equal(leg.tables()[1].numpy(), FO.psf_filterbanks(26, 512, 16000))


def test_steps_fail_loudly_without_a_device():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        Standardize()(torch.randn(4, 4))
    with pytest.raises(RuntimeError):
        AddContextFrames(2)(torch.randn(1, 3, 5))
    with pytest.raises(RuntimeError):
        MFCC(n_mfcc=20)(torch.randn(1, 4000))
