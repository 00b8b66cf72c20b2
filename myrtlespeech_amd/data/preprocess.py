"""Feature front-end on the GPU, API of myrtlespeech/data/preprocess.py plus the
``torchaudio.transforms.MFCC`` the reference builder instantiates (builders/pre_process_step.py:33-43).

Every step has the reference's per-sample call (``step(x)`` on one utterance, as the dataset applies it)
and a ``batch`` method over a zero-padded ragged batch ``(x, lens)`` that gives, utterance by utterance, what
per-sample processing followed by data/batch.py's padding gives -- one launch set for the whole batch.
``batch`` always returns ``(x, lens)`` so the steps chain (``SeqToSeq.pre_process_batch``).
All arithmetic runs in libms_hotpath.so (csrc/frontend.hip); without a HIP device the calls raise.
"""
import math
import random
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch

from myrtlespeech_amd import _lib


class AddSequenceLength:
    """``x -> (x, tensor([x.size(length_dim)]))`` (preprocess.py:13-40); host-only bookkeeping."""

    def __init__(self, length_dim: int = 0):
        self.length_dim = length_dim

    def __call__(self, data: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return data, torch.tensor([data.size(self.length_dim)], requires_grad=False)

    def __repr__(self) -> str:
        return f"{type(self).__name__}(length_dim={self.length_dim})"


class Standardize:
    """Zero mean, unit (unbiased) standard deviation over the whole utterance (preprocess.py:43-66)."""

    def __init__(self):
        self._ws = _lib.Workspace()

    def _run(self, x: torch.Tensor, lens: Optional[torch.Tensor], N: int, inner: int, T: int) -> torch.Tensor:
        lib = _lib.load()
        xd = _lib.f32c(x.detach())
        y = torch.empty_like(xd)
        lens_d = _lib.lens_i32(lens) if lens is not None else None
        nbytes = lib.ms_standardize_workspace_bytes(N)
        ws = self._ws.get(nbytes)
        _lib.check(lib.ms_standardize_forward(_lib.ptr(xd), _lib.ptr(lens_d), _lib.ptr(y), N, inner, T, _lib.ptr(ws),
                                              nbytes, _lib.stream_ptr()), "ms_standardize_forward")
        return y

    def __call__(self, tensor: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu()
        if tensor.numel() == 0:
            return _lib.f32c(tensor.detach())
        return self._run(tensor, None, 1, 1, tensor.numel()).view(tensor.shape)

    def batch(self, x: torch.Tensor, lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [N, ..., T] zero-padded, lens [N]: each utterance is standardised over its own valid frames."""
        _lib.require_gpu()
        N, T = x.size(0), x.size(-1)
        return self._run(x, lens, N, x[0].numel() // T, T).view(x.shape), lens

    def __repr__(self) -> str:
        return f"{type(self).__name__}()"


class AddContextFrames:
    """Stacks the ``n_context`` frames either side of every frame as channels:
    ``[1, features, T] -> [2*n_context + 1, features, T]``, zeros beyond the utterance (preprocess.py:69-144)."""

    def __init__(self, n_context: int):
        self.n_context = n_context

    def _run(self, x: torch.Tensor, lens: Optional[torch.Tensor], N: int, F: int, T: int) -> torch.Tensor:
        lib = _lib.load()
        dtype = x.dtype
        xd = _lib.f32c(x)
        y = torch.empty((N, 2 * self.n_context + 1, F, T), dtype=torch.float32, device=xd.device)
        lens_d = _lib.lens_i32(lens) if lens is not None else None
        if y.numel():
            _lib.check(lib.ms_context_frames_forward(_lib.ptr(xd), _lib.ptr(lens_d), _lib.ptr(y), N, F, T,
                                                     self.n_context, _lib.stream_ptr()), "ms_context_frames_forward")
        return y if dtype == torch.float32 else y.to(dtype)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu()
        assert x.dim() == 3 and x.size(0) == 1, "expected size (1, features, seq_len)"
        return self._run(x, None, 1, x.size(1), x.size(2))[0]

    def batch(self, x: torch.Tensor, lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [N, 1, features, T], lens [N] -> ([N, 2*n_context + 1, features, T], lens)."""
        _lib.require_gpu()
        assert x.dim() == 4 and x.size(1) == 1
        return self._run(x, lens, x.size(0), x.size(2), x.size(3)), lens

    def __repr__(self) -> str:
        return f"{type(self).__name__}(n_context={self.n_context})"


class SpecAugment:
    """SpecAugment masking (preprocess.py:147-230): ``n_feature_masks`` bands of up to ``feature_mask`` feature
    rows and ``n_time_masks`` bands of up to ``time_mask`` frames are zeroed in place.  The band widths and
    starts are drawn on the host from Python's ``random`` in the reference's order (per mask: width, then
    start), so a seeded run masks exactly what the reference masks; the zeroing is one kernel."""

    def __init__(self, feature_mask: int, time_mask: int, n_feature_masks: int = 1, n_time_masks: int = 1):
        for name, value in (("feature_mask", feature_mask), ("time_mask", time_mask),
                            ("n_feature_masks", n_feature_masks), ("n_time_masks", n_time_masks)):
            if value < 0:
                raise ValueError(f"{name}={value} < 0")
        self.feature_mask = feature_mask
        self.time_mask = time_mask
        self.n_feature_masks = n_feature_masks
        self.n_time_masks = n_time_masks

    def _draw(self, n_features: int, n_time_steps: int):
        f_bands, t_bands = [], []
        for _ in range(self.n_feature_masks):
            width = random.randint(0, self.feature_mask)
            f_bands.append((random.randint(0, max(0, n_features - width)), width))
        for _ in range(self.n_time_masks):
            width = random.randint(0, self.time_mask)
            t_bands.append((random.randint(0, max(0, n_time_steps - width)), width))
        return f_bands, t_bands

    def _zero(self, x: torch.Tensor, f_bands, t_bands, N: int, C: int, F: int, T: int) -> None:
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("SpecAugment works in place on a contiguous float32 device tensor")
        fb = torch.tensor(f_bands, dtype=torch.int32).reshape(N, self.n_feature_masks, 2).cuda()
        tb = torch.tensor(t_bands, dtype=torch.int32).reshape(N, self.n_time_masks, 2).cuda()
        if x.numel():
            _lib.check(_lib.load().ms_spec_augment_(_lib.ptr(x), _lib.ptr(fb), _lib.ptr(tb), N, C, F, T,
                                                    self.n_feature_masks, self.n_time_masks, _lib.stream_ptr()),
                       "ms_spec_augment_")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x ``[channels, features, T]`` (device tensor), masked in place and returned."""
        _lib.require_gpu()
        C, F, T = x.size()
        f_bands, t_bands = self._draw(F, T)
        self._zero(x, [f_bands], [t_bands], 1, C, F, T)
        return x

    def batch(self, x: torch.Tensor, lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x ``[N, channels, features, T]``: utterance n draws its own bands against its own length, in batch order."""
        _lib.require_gpu()
        N, C, F, T = x.size()
        drawn = [self._draw(F, int(n_steps)) for n_steps in lens.tolist()]
        self._zero(x, [d[0] for d in drawn], [d[1] for d in drawn], N, C, F, T)
        return x, lens

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(feature_mask={self.feature_mask}, time_mask={self.time_mask},"
                f" n_feature_masks={self.n_feature_masks}, n_time_masks={self.n_time_masks})")


class MFCC:
    """``torchaudio.transforms.MFCC`` of torchaudio 0.4.0 (environment.yml:171) with the arguments the reference
    builder passes (``n_mfcc`` and ``melkwargs = {win_length, hop_length}``; everything else at that release's
    defaults): centred reflect-padded STFT (n_fft 400, periodic Hann window) -> power -> 128 HTK-mel triangles over
    0..sample_rate/2 -> ``10 log10(max(., 1e-10))`` floored at ``max - 80 dB`` -> orthonormal DCT-II, first ``n_mfcc``.
    The tables are built on the host with torch float32 ops in torchaudio's operation order; the DFT, mel and DCT
    contractions are exact-f32 MFMA GEMMs on the device."""

    def __init__(self, sample_rate: int = 16000, n_mfcc: int = 40, dct_type: int = 2, norm: str = "ortho",
                 log_mels: bool = False, melkwargs: Optional[Dict] = None):
        if dct_type != 2:
            raise ValueError("DCT type not supported")
        if norm != "ortho":
            raise ValueError("only norm='ortho' is supported")
        if log_mels:
            raise ValueError("log_mels=True is not supported (the reference builder never sets it)")
        kw = dict(melkwargs or {})
        n_fft = kw.pop("n_fft", 400)
        win_length = kw.pop("win_length", None) or n_fft
        hop_length = kw.pop("hop_length", None) or win_length // 2
        n_mels = kw.pop("n_mels", 128)
        f_min = kw.pop("f_min", 0.0)
        f_max = kw.pop("f_max", None)
        if kw:
            raise ValueError(f"unsupported melkwargs {sorted(kw)}")
        if n_mfcc > n_mels:
            raise ValueError("Cannot select more MFCC coefficients than # mel bins")
        if not 0 < win_length <= n_fft:
            raise ValueError(f"win_length={win_length} must be in (0, n_fft={n_fft}]")
        if hop_length <= 0:
            raise ValueError(f"hop_length={hop_length} must be positive")
        self.sample_rate = sample_rate
        self.n_mfcc = n_mfcc
        self.dct_type = dct_type
        self.norm = norm
        self.log_mels = log_mels
        self.top_db = 80.0
        self.MelSpectrogram = SimpleNamespace(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length,
                                              hop_length=hop_length, n_mels=n_mels, f_min=f_min,
                                              f_max=float(f_max if f_max is not None else sample_rate // 2), pad=0)
        self._tables = None
        self._ws = _lib.Workspace()

    # -- host-side tables ---------------------------------------------------------------------------------------
    def tables(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(window [n_fft], dft [2*(n_fft/2+1), n_fft], mel_fb [n_mels, n_fft/2+1], dct [n_mfcc, n_mels]) on CPU."""
        ms = self.MelSpectrogram
        n_fft, n_freqs = ms.n_fft, ms.n_fft // 2 + 1
        window = torch.zeros(n_fft)
        left = (n_fft - ms.win_length) // 2
        window[left:left + ms.win_length] = torch.hann_window(ms.win_length)
        # DFT basis in float64, rounded once: row k = cos(2 pi k j / n_fft), row n_freqs + k = -sin(2 pi k j / n_fft);
        # the angle is reduced exactly as (k * j) mod n_fft before the trigonometry
        kj = (torch.arange(n_freqs, dtype=torch.int64)[:, None] * torch.arange(n_fft, dtype=torch.int64)[None, :]) % n_fft
        ang = kj.to(torch.float64) * (2.0 * math.pi / n_fft)
        dft = torch.cat([torch.cos(ang), -torch.sin(ang)]).to(torch.float32)
        # mel filterbank (HTK scale, unnormalised triangles)
        all_freqs = torch.linspace(ms.f_min, ms.f_max, n_freqs)
        m_min = 0.0 if ms.f_min == 0 else 2595.0 * math.log10(1.0 + ms.f_min / 700.0)
        m_max = 2595.0 * math.log10(1.0 + ms.f_max / 700.0)
        f_pts = 700.0 * (10 ** (torch.linspace(m_min, m_max, ms.n_mels + 2) / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        fb = torch.max(torch.zeros(1), torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
        # orthonormal DCT-II
        n = torch.arange(float(ms.n_mels))
        k = torch.arange(float(self.n_mfcc)).unsqueeze(1)
        dct = torch.cos(math.pi / float(ms.n_mels) * (n + 0.5) * k)
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(ms.n_mels))
        return window, dft.contiguous(), fb.t().contiguous(), dct.contiguous()

    def _device_tables(self):
        if self._tables is None:
            self._tables = tuple(t.cuda() for t in self.tables())
        return self._tables

    def frames(self, samples: int) -> int:
        return 1 + samples // self.MelSpectrogram.hop_length

    # -- device path --------------------------------------------------------------------------------------------
    def batch(self, waves: torch.Tensor, wave_lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """waves [N, max_samples] zero-padded, wave_lens [N] -> (features [N, 1, n_mfcc, T], frame lengths [N])."""
        _lib.require_gpu()
        lib = _lib.load()
        ms = self.MelSpectrogram
        N, L = waves.size()
        lens_host = wave_lens.detach().cpu().to(torch.int64)
        if N == 0 or int(lens_host.min()) <= ms.n_fft // 2 or int(lens_host.max()) > L:
            raise ValueError(f"every waveform needs more than n_fft/2 = {ms.n_fft // 2} samples (reflect padding) "
                             f"and at most max_samples = {L}")
        T = self.frames(L)
        window, dft, fb, dct = self._device_tables()
        x = _lib.f32c(waves)
        lens_d = _lib.lens_i32(lens_host)
        out = torch.empty((N, 1, self.n_mfcc, T), dtype=torch.float32, device=x.device)
        nbytes = lib.ms_mfcc_workspace_bytes(N, T, ms.n_fft, ms.n_mels, self.n_mfcc)
        ws = self._ws.get(nbytes)
        _lib.check(lib.ms_mfcc_forward(_lib.ptr(x), _lib.ptr(lens_d), _lib.ptr(window), _lib.ptr(dft), _lib.ptr(fb),
                                       _lib.ptr(dct), _lib.ptr(out), N, L, T, ms.n_fft, ms.hop_length, ms.n_mels,
                                       self.n_mfcc, self.top_db, _lib.ptr(ws), nbytes, _lib.stream_ptr()),
                   "ms_mfcc_forward")
        return out, 1 + lens_host // ms.hop_length

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        """waveform ``[channel, time]`` -> ``[channel, n_mfcc, frames]`` (each channel on its own, like torchaudio)."""
        C, L = waveform.size()
        out, _ = self.batch(waveform, torch.full((C,), L, dtype=torch.int64))
        return out[:, 0]

    forward = __call__

    def __repr__(self) -> str:
        ms = self.MelSpectrogram
        return (f"{type(self).__name__}(n_mfcc={self.n_mfcc}, n_fft={ms.n_fft}, win_length={ms.win_length}, "
                f"hop_length={ms.hop_length}, n_mels={ms.n_mels}, sample_rate={self.sample_rate})")


class MFCCLegacy:
    """``python_speech_features.mfcc`` (0.6, environment.yml:169) as MFCCLegacy drives it (preprocess.py:233-328):
    samples scaled to int16, float64 arithmetic, pre-emphasis 0.97, rectangular ``winlen`` frames every ``winstep``,
    NFFT = next power of two, 26 HTK-mel filters, log, orthonormal DCT-II, lifter 22, c0 := log frame energy.
    One float64 workgroup per frame on the device (``ms_mfcc_legacy_forward``)."""

    NFILT = 26
    PREEMPH = 0.97
    CEPLIFTER = 22

    def __init__(self, n_mfcc: int, melkwargs: Dict, sample_rate: int = 16000):
        self.n_mfcc = n_mfcc
        self.samplerate = sample_rate
        self.numcep = n_mfcc
        self.winlen = melkwargs["win_length"] / sample_rate
        self.winstep = melkwargs["hop_length"] / sample_rate
        if n_mfcc > self.NFILT:
            raise ValueError(f"n_mfcc={n_mfcc} exceeds the {self.NFILT} filterbank channels of the legacy pipeline")
        self._tables = None

    @staticmethod
    def _round_half_up(v: float) -> int:
        return int(math.floor(v + 0.5))

    def geometry(self) -> Tuple[int, int, int]:
        """(frame_len, frame_step, nfft) in samples."""
        window = self.winlen * self.samplerate
        nfft = 1
        while nfft < window:
            nfft *= 2
        return self._round_half_up(window), self._round_half_up(self.winstep * self.samplerate), nfft

    def frames(self, samples: int) -> int:
        frame_len, frame_step, _ = self.geometry()
        return 1 if samples <= frame_len else 1 + -(-(samples - frame_len) // frame_step)

    def tables(self):
        """(twiddle [nfft, 2], fbank [26, nfft/2+1], dct [numcep, 26], lifter [numcep]) float64 on CPU."""
        _, _, nfft = self.geometry()
        j = torch.arange(nfft, dtype=torch.float64) * (2.0 * math.pi / nfft)
        twiddle = torch.stack([torch.cos(j), torch.sin(j)], dim=1)
        hz2mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
        mel_pts = torch.linspace(hz2mel(0.0), hz2mel(self.samplerate / 2), self.NFILT + 2, dtype=torch.float64)
        bins = torch.floor((nfft + 1) * (700.0 * (10.0 ** (mel_pts / 2595.0) - 1.0)) / self.samplerate).tolist()
        fbank = torch.zeros(self.NFILT, nfft // 2 + 1, dtype=torch.float64)
        for m in range(self.NFILT):
            lo, mid, hi = bins[m], bins[m + 1], bins[m + 2]
            for i in range(int(lo), int(mid)):
                fbank[m, i] = (i - lo) / (mid - lo)
            for i in range(int(mid), int(hi)):
                fbank[m, i] = (hi - i) / (hi - mid)
        n = torch.arange(self.NFILT, dtype=torch.float64)
        dct = torch.cos(math.pi * (n[None, :] + 0.5) * n[:, None] / self.NFILT) * math.sqrt(2.0 / self.NFILT)
        dct[0] *= 1.0 / math.sqrt(2.0)
        lifter = 1.0 + (self.CEPLIFTER / 2.0) * torch.sin(math.pi * torch.arange(self.numcep, dtype=torch.float64)
                                                          / self.CEPLIFTER)
        return twiddle.contiguous(), fbank, dct[:self.numcep].contiguous(), lifter

    def batch(self, waves: torch.Tensor, wave_lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """waves [N, max_samples] in [-1, 1], wave_lens [N] -> (features [N, 1, n_mfcc, T], frame lengths [N])."""
        _lib.require_gpu()
        N, L = waves.size()
        lens_host = wave_lens.detach().cpu().to(torch.int64)
        if N == 0 or int(lens_host.min()) < 1 or int(lens_host.max()) > L:
            raise ValueError("every waveform needs between 1 and max_samples samples")
        if self._tables is None:
            self._tables = tuple(t.cuda() for t in self.tables())
        twiddle, fbank, dct, lifter = self._tables
        frame_len, frame_step, nfft = self.geometry()
        T = self.frames(L)
        x = _lib.f32c(waves)
        lens_d = _lib.lens_i32(lens_host)
        out = torch.empty((N, 1, self.numcep, T), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().ms_mfcc_legacy_forward(_lib.ptr(x), _lib.ptr(lens_d), _lib.ptr(twiddle), _lib.ptr(fbank),
                                                      _lib.ptr(dct), _lib.ptr(lifter), _lib.ptr(out), N, L, T, frame_len,
                                                      frame_step, nfft, self.NFILT, self.numcep, self.PREEMPH,
                                                      _lib.stream_ptr()), "ms_mfcc_legacy_forward")
        return out, torch.tensor([self.frames(int(v)) for v in lens_host.tolist()], dtype=torch.int64)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x ``[1, time_samples]`` -> ``[1, n_mfcc, frames]``."""
        flat = x.reshape(1, -1)
        out, _ = self.batch(flat, torch.tensor([flat.size(1)]))
        return out[0]

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(numcep={self.numcep}, winlen={self.winlen}, winstep={self.winstep}, "
                f"samplerate={self.samplerate})")
