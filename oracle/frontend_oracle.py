"""CPU restatement (numpy) of the reference's feature front-end -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (myrtlespeech_amd.data.preprocess) never does.

What is pinned and what is not
------------------------------
* ``standardize``, ``add_context_frames``, ``spec_augment``, ``add_sequence_length`` restate
  myrtlespeech/data/preprocess.py:13-218.  That module cannot be imported in the build container
  (``import python_speech_features`` at preprocess.py:9 raises ModuleNotFoundError), so the pins are
  the reference's own documented vectors: the AddContextFrames docstring example
  (preprocess.py:77-105, stored as tests/golden/context_frames_doc.npz), the Standardize doctest
  property (preprocess.py:46-54) and the SpecAugment bound of tests/data/test_preprocess.py:67-87.
* ``pad_sequence`` / ``seq_to_seq_collate`` restate myrtlespeech/data/batch.py:7-107 and ARE pinned by
  fixtures generated from the imported reference (tests/golden/collate.npz).
* ``mfcc`` restates torchaudio==0.4.0 ``transforms.MFCC`` (environment.yml:171; the call site is
  builders/pre_process_step.py:33-43 with melkwargs win_length/hop_length and every other argument at
  its 0.4.0 default: sample_rate 16000, n_fft 400, n_mels 128, f_min 0, f_max sr/2, hann window,
  power 2, dct type 2 norm 'ortho', log_mels False, top_db 80).  torchaudio is absent from the
  image: **parity unpinned** for the full transform; its STFT stage is cross-checked against the
  installed ``torch.stft`` in tests/test_oracle_golden.py.
* ``mfcc_legacy`` restates python_speech_features==0.6 ``mfcc`` (environment.yml:169) as driven by
  MFCCLegacy.__call__ (preprocess.py:262-319): int16 rescale, float64 arithmetic, pre-emphasis 0.97,
  rectangular window, NFFT = next power of two, 26 HTK-mel triangular filters, log, DCT-II ortho,
  lifter 22, c0 replaced by log frame energy.  The library is absent: **parity unpinned**.
"""
import math
import random as _random

import numpy as np

F32 = np.float32


# ---- data/preprocess.py ------------------------------------------------------------------------

def add_sequence_length(x, length_dim=0):
    """AddSequenceLength.__call__ (preprocess.py:32-37)."""
    return x, np.array([x.shape[length_dim]], dtype=np.int64)


def standardize(x):
    """Standardize.__call__ (preprocess.py:57-63): (x - mean) / std with the unbiased std."""
    x = np.asarray(x, dtype=F32)
    mean = F32(x.mean(dtype=np.float64))
    std = F32(x.astype(np.float64).std(ddof=1))
    return ((x - mean) / std).astype(F32)


def add_context_frames(x, n_context):
    """AddContextFrames.__call__ (preprocess.py:117-141): x [1, F, T] -> [2c+1, F, T]."""
    assert x.shape[0] == 1
    _, F, T = x.shape
    W = 2 * n_context + 1
    out = np.zeros((W, F, T), dtype=x.dtype)
    for w in range(W):
        for t in range(T):
            s = t + w - n_context
            if 0 <= s < T:
                out[w, :, t] = x[0, :, s]
    return out


def spec_augment_bands(n_features, n_time_steps, feature_mask, time_mask, n_feature_masks, n_time_masks, rng=_random):
    """The random draws of SpecAugment.__call__ (preprocess.py:204-216) in the reference's order:
    for each feature mask (width, start), then for each time mask (width, start)."""
    f_bands, t_bands = [], []
    for _ in range(n_feature_masks):
        width = rng.randint(0, feature_mask)
        start = rng.randint(0, max(0, n_features - width))
        f_bands.append((start, width))
    for _ in range(n_time_masks):
        width = rng.randint(0, time_mask)
        start = rng.randint(0, max(0, n_time_steps - width))
        t_bands.append((start, width))
    return f_bands, t_bands


def spec_augment(x, feature_mask, time_mask, n_feature_masks=1, n_time_masks=1, rng=_random):
    """SpecAugment.__call__ (preprocess.py:193-218) on x [C, F, T]; modifies and returns x."""
    _, F, T = x.shape
    f_bands, t_bands = spec_augment_bands(F, T, feature_mask, time_mask, n_feature_masks, n_time_masks, rng)
    for start, width in f_bands:
        x[:, start:start + width, :] = 0
    for start, width in t_bands:
        x[:, :, start:start + width] = 0
    return x


# ---- data/batch.py -----------------------------------------------------------------------------

def pad_sequence(sequences, padding_value=0):
    """pad_sequence (batch.py:7-42): stack [*, len_i] arrays into [batch, *, max_len]."""
    lead = sequences[0].shape[:-1]
    max_len = max(s.shape[-1] for s in sequences)
    out = np.full((len(sequences),) + lead + (max_len,), padding_value, dtype=sequences[0].dtype)
    for i, s in enumerate(sequences):
        out[i, ..., :s.shape[-1]] = s
    return out


def seq_to_seq_collate(batch):
    """seq_to_seq_collate_fn (batch.py:45-107): stable sort by input length, descending, then pad."""
    order = sorted(range(len(batch)), key=lambda i: batch[i][0][0].shape[-1], reverse=True)
    xs = [batch[i][0][0] for i in order]
    xl = np.array([int(np.asarray(batch[i][0][1]).reshape(-1)[0]) for i in order], dtype=np.int64)
    ys = [batch[i][1][0] for i in order]
    yl = np.array([int(np.asarray(batch[i][1][1]).reshape(-1)[0]) for i in order], dtype=np.int64)
    return (pad_sequence(xs), xl), (pad_sequence(ys), yl)


# ---- torchaudio 0.4.0 MFCC ---------------------------------------------------------------------

def hann_window(n):
    """torch.hann_window(n) (periodic)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(F32)


def stft_power(wave, n_fft=400, hop=160, win_length=400):
    """torchaudio.functional.spectrogram with power 2: torch.stft(center=True, 'reflect', onesided) -> re^2 + im^2.
    wave [L] -> [n_fft/2+1, 1 + L//hop]."""
    wave = np.asarray(wave, dtype=F32)
    window = np.zeros(n_fft, dtype=F32)
    left = (n_fft - win_length) // 2
    window[left:left + win_length] = hann_window(win_length)
    padded = np.pad(wave, n_fft // 2, mode="reflect")
    n_frames = 1 + len(wave) // hop
    frames = np.stack([padded[t * hop:t * hop + n_fft] for t in range(n_frames)]) * window
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    re, im = spec.real.astype(F32), spec.imag.astype(F32)
    return (re * re + im * im).T.astype(F32)


def mel_filterbank(n_freqs, f_min, f_max, n_mels):
    """torchaudio.functional.create_fb_matrix (0.4.0): HTK mel, unnormalised triangles; [n_freqs, n_mels]."""
    all_freqs = np.linspace(f_min, f_max, n_freqs, dtype=F32)
    m_min = 0.0 if f_min == 0 else 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2, dtype=F32)
    f_pts = (F32(700.0) * (np.power(F32(10.0), m_pts / F32(2595.0)) - F32(1.0))).astype(F32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (F32(-1.0) * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(F32(0.0), np.minimum(down, up)).astype(F32)


def create_dct(n_mfcc, n_mels):
    """torchaudio.functional.create_dct(norm='ortho'): [n_mels, n_mfcc]."""
    n = np.arange(n_mels, dtype=F32)
    k = np.arange(n_mfcc, dtype=F32)[:, None]
    dct = np.cos(F32(math.pi / float(n_mels)) * (n + F32(0.5)) * k).astype(F32)
    dct[0] *= F32(1.0 / math.sqrt(2.0))
    dct *= F32(math.sqrt(2.0 / float(n_mels)))
    return dct.T.copy()


def amplitude_to_db(x, multiplier=10.0, amin=1e-10, db_multiplier=0.0, top_db=80.0):
    """torchaudio.functional.amplitude_to_DB (0.4.0): the floor is taken against the max of the whole tensor."""
    x_db = F32(multiplier) * np.log10(np.maximum(x, F32(amin))).astype(F32)
    x_db = x_db - F32(multiplier * db_multiplier)
    if top_db is not None:
        x_db = np.maximum(x_db, F32(float(x_db.max()) - top_db))
    return x_db.astype(F32)


def mfcc(wave, n_mfcc, win_length, hop_length, sample_rate=16000, n_fft=400, n_mels=128, top_db=80.0):
    """torchaudio.transforms.MFCC.forward (0.4.0): wave [1, L] -> [1, n_mfcc, 1 + L//hop]."""
    assert wave.shape[0] == 1
    power = stft_power(wave[0], n_fft, hop_length, win_length)                      # [n_freqs, T]
    fb = mel_filterbank(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels)        # [n_freqs, n_mels]
    mel = (power.T.astype(np.float64) @ fb.astype(np.float64)).astype(F32)           # [T, n_mels]
    mel_db = amplitude_to_db(mel, top_db=top_db)
    cep = (mel_db.astype(np.float64) @ create_dct(n_mfcc, n_mels).astype(np.float64)).astype(F32)
    return cep.T[None]


# ---- python_speech_features 0.6 mfcc as driven by MFCCLegacy -----------------------------------

def _round_half_up(v):
    return int(math.floor(v + 0.5))


def psf_filterbanks(nfilt, nfft, samplerate, lowfreq=0.0, highfreq=None):
    """python_speech_features.get_filterbanks: [nfilt, nfft/2+1] float64."""
    highfreq = highfreq or samplerate / 2
    hz2mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
    mel2hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fb = np.zeros((nfilt, nfft // 2 + 1))
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def mfcc_legacy(wave, n_mfcc, win_length, hop_length, sample_rate=16000, nfilt=26, preemph=0.97, ceplifter=22):
    """MFCCLegacy.__call__ (preprocess.py:262-319): wave [1, L] in [-1, 1] -> [1, n_mfcc, frames] float32."""
    x = (np.asarray(wave, dtype=F32) * F32(1 << 15)).astype(np.int16).reshape(-1).astype(np.float64)
    winlen, winstep = win_length / sample_rate, hop_length / sample_rate
    nfft = 1
    while nfft < winlen * sample_rate:
        nfft *= 2
    sig = np.append(x[0], x[1:] - preemph * x[:-1])
    frame_len, frame_step = _round_half_up(winlen * sample_rate), _round_half_up(winstep * sample_rate)
    slen = len(sig)
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = (numframes - 1) * frame_step + frame_len
    padded = np.concatenate((sig, np.zeros(padlen - slen)))
    frames = np.stack([padded[t * frame_step:t * frame_step + frame_len] for t in range(numframes)])
    pspec = (1.0 / nfft) * np.square(np.abs(np.fft.rfft(frames, nfft)))
    energy = pspec.sum(1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    feat = pspec @ psf_filterbanks(nfilt, nfft, sample_rate).T
    feat = np.log(np.where(feat == 0, np.finfo(float).eps, feat))
    n = np.arange(nfilt)
    basis = np.cos(np.pi * (n[None, :] + 0.5) * n[:, None] / nfilt) * np.sqrt(2.0 / nfilt)   # DCT-II, norm='ortho'
    basis[0] *= 1.0 / np.sqrt(2.0)
    cep = (feat @ basis.T)[:, :n_mfcc]
    lift = 1.0 + (ceplifter / 2.0) * np.sin(np.pi * np.arange(n_mfcc) / ceplifter)
    cep = cep * lift
    cep[:, 0] = np.log(energy)
    return cep.astype(F32).T[None]
