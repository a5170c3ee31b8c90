"""Host-side mirror of the reference audio front end (audio_processing.py:96-134, 172-235):
TacotronSTFT.mel_spectrogram as ONE HIP kernel (ft_stft_mel: reflect pad + windowed radix-2 FFT
in LDS + |X| + mel filterbank + log-compression) instead of a 1026x1024 dense-DFT conv1d.

The mel filterbank constants come from librosa in the reference (third-party dependency that is
not vendored: requirements.txt:4 pins 0.6.3, Dockerfile:6 pins 0.8.0; call site
audio_processing.py:104-105 -> htk=False, Slaney area normalisation).  librosa is not installed
here, so `slaney_mel_filterbank` restates the published formula; if librosa IS importable it is
used, exactly like the reference.  PARITY UNPINNED for these constants (no reference test holds them).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib as L


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None) -> np.ndarray:
    """[n_mels, n_fft//2+1] float32 triangular filters, Slaney mel scale, area-normalised."""
    if fmax is None:
        fmax = sr / 2.0
    try:                                         # the reference's own source of these constants, when present
        from librosa.filters import mel as librosa_mel_fn
        return np.asarray(librosa_mel_fn(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax), dtype=np.float32)
    except Exception:
        pass
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def hann_window(win_length: int, filter_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win_length, fftbins=True) zero-centre-padded to filter_length
    (audio_processing.py:193-197)."""
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    lpad = (filter_length - win_length) // 2
    out = np.zeros(filter_length, dtype=np.float64)
    out[lpad:lpad + win_length] = w
    return out.astype(np.float32)


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    return torch.exp(x) / C


def filterbank_csr(mel_basis: np.ndarray):
    """[n_mel, n_bins] triangular filterbank -> (first bin per band int32[n_mel], row pointer int32[n_mel+1], weights fp32[nnz]):
    every band's non-zeros are one run of consecutive bins (a triangle), ~1 000 weights instead of 80 x 513."""
    bin0, ptr, w = [], [0], []
    for row in np.asarray(mel_basis, dtype=np.float32):
        nz = np.nonzero(row)[0]
        if len(nz) == 0:
            bin0.append(0)
            ptr.append(ptr[-1])
            continue
        bin0.append(int(nz[0]))
        w.extend(row[nz[0]:nz[-1] + 1].tolist())
        ptr.append(len(w))
    return np.asarray(bin0, np.int32), np.asarray(ptr, np.int32), np.asarray(w, np.float32)


class STFT(torch.nn.Module):
    """Analysis half of the reference STFT (audio_processing.py:172-235): `transform(y)` -> (magnitude, phase), both
    [B, n_fft/2+1, N // hop + 1], one HIP kernel (real FFT, csrc/stft_r8.hip).  The inverse (Griffin-Lim side) is out of scope."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        assert window == "hann" and filter_length >= win_length
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        self.register_buffer("fft_window", torch.from_numpy(hann_window(win_length, filter_length)))

    def fast_path(self):
        return self.filter_length == 1024 and self.hop_length <= 256

    def transform(self, input_data):
        """audio_processing.py:207-235."""
        L.require_cuda(input_data)
        if not self.fast_path():
            raise NotImplementedError("STFT.transform is built for filter_length 1024 / hop <= 256 (config.json:32-34)")
        y = input_data.contiguous().float()
        if self.fft_window.device != y.device:
            self.to(y.device)
        B, N = y.shape
        n_frames = N // self.hop_length + 1
        mag = torch.empty(B, self.filter_length // 2 + 1, n_frames, device=y.device, dtype=torch.float32)
        phase = torch.empty_like(mag)
        L.check(L.lib().ft_stft_r8(L.ptr(y), L.ptr(self.fft_window), None, None, None, None, L.ptr(mag), L.ptr(phase), B, N,
                                   self.hop_length, 0, L.stream()), "ft_stft_r8")
        return mag, phase


class TacotronSTFT(torch.nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=None):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        basis = slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer("mel_basis", torch.from_numpy(basis).float())
        bin0, ptr, w = filterbank_csr(basis)          # sparse form of the same matrix for the rFFT kernel (not in the state_dict)
        self.register_buffer("fb_bin0", torch.from_numpy(bin0), persistent=False)
        self.register_buffer("fb_ptr", torch.from_numpy(ptr), persistent=False)
        self.register_buffer("fb_w", torch.from_numpy(w), persistent=False)

    def spectral_normalize(self, magnitudes):
        return dynamic_range_compression(magnitudes)

    def spectral_de_normalize(self, magnitudes):
        return dynamic_range_decompression(magnitudes)

    def mel_spectrogram_ragged(self, y, n_samples, max_t=None):
        """The collated batch of the data path in one launch (ft_stft_r8_ragged): y [B,N] zero-padded audio on the device,
        n_samples [B] (int32, device) -> [B, n_mel_channels, max_t]; utterance b's frames t < n_samples[b] // hop + 1 equal
        mel_spectrogram(y[b:b+1, :n_samples[b]]), later frames are zero (DataCollate's padding, data.py:215-229)."""
        L.require_cuda(y, n_samples)
        y = y.contiguous().float()
        if self.mel_basis.device != y.device:
            self.to(y.device)
        B, N = y.shape
        st = self.stft_fn
        T_out = N // st.hop_length + 1 if max_t is None else int(max_t)
        if not (st.fast_path() and self.n_mel_channels <= 128 and N > st.filter_length // 2):
            raise NotImplementedError("the ragged front end is built for n_fft = 1024, hop <= 256 (config.json:32-34)")
        mel = torch.empty(B, self.n_mel_channels, T_out, device=y.device, dtype=torch.float32)
        L.check(L.lib().ft_stft_r8_ragged(L.ptr(y), L.ptr(n_samples.to(torch.int32)), L.ptr(st.fft_window), L.ptr(self.fb_bin0),
                                          L.ptr(self.fb_ptr), L.ptr(self.fb_w), L.ptr(mel), B, N, st.hop_length, self.n_mel_channels,
                                          T_out, L.stream()), "ft_stft_r8_ragged")
        return mel

    def mel_spectrogram(self, y):
        """y [B,N] float32 in [-1,1] (device tensor) -> [B, n_mel_channels, N // hop + 1]."""
        L.require_cuda(y)
        assert y.dim() == 2
        # the reference asserts the value range with two host syncs (audio_processing.py:127-128); done only when the caller asks
        # (TacotronSTFT.check_audio_range = True), to keep the front end sync-free
        if getattr(self, "check_audio_range", False):
            assert float(y.min()) >= -1 and float(y.max()) <= 1
        y = y.contiguous().float()
        if self.mel_basis.device != y.device:
            self.to(y.device)
        B, N = y.shape
        st = self.stft_fn
        n_frames = N // st.hop_length + 1
        mel = torch.empty(B, self.n_mel_channels, n_frames, device=y.device, dtype=torch.float32)
        if st.fast_path() and self.n_mel_channels <= 128 and N > st.filter_length // 2:
            # rFFT (512-point complex FFT + split) + sparse triangular filterbank, one wave per frame (csrc/stft_r8.hip)
            L.check(L.lib().ft_stft_r8(L.ptr(y), L.ptr(st.fft_window), L.ptr(self.fb_bin0), L.ptr(self.fb_ptr), L.ptr(self.fb_w),
                                       L.ptr(mel), None, None, B, N, st.hop_length, self.n_mel_channels, L.stream()), "ft_stft_r8")
        else:                                           # general n_fft: complex radix-2 FFT + dense filterbank (csrc/stft.hip)
            L.check(L.lib().ft_stft_mel(L.ptr(y), L.ptr(st.fft_window), L.ptr(self.mel_basis), L.ptr(mel), B, N,
                                        st.filter_length, st.hop_length, self.n_mel_channels, L.stream()), "ft_stft_mel")
        return mel
