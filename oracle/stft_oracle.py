"""CPU oracle for the STFT / mel frontend.

TEST INFRASTRUCTURE ONLY (see oracle/waveglow_oracle.py header for the import rule).
numpy restatement written from SURVEY.md §8(a) rows S1-S5, following
/root/reference/CookieTTS/utils/audio/stft.py:46-77 (windowed DFT basis), :79-111
(reflect pad, strided conv, magnitude), :180-207 (mel projection, log clamp) and
utils/audio/audio_processing.py:78-84 (dynamic_range_compression).

Parity pin: tests/golden/stft_mel.npz holds outputs of the reference's own
``STFT.transform`` / ``TacotronSTFT.mel_spectrogram`` run by tests/golden/make_golden.py.
PARITY UNPINNED at one boundary: the mel filterbank.  The reference obtains it from
``librosa.filters.mel`` (stft.py:163-164; librosa is a third-party dependency, unpinned in
requirements.txt:4 and absent from this image), so ``slaney_mel_filterbank`` below restates the
published Slaney-style algorithm (linear below 1 kHz at 200/3 Hz per mel, log above with step
ln(6.4)/27, triangular weights, area normalisation 2/(f[i+2]-f[i])) and the golden generator feeds
that same matrix to the reference; every other stage is pinned by import.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def hann_periodic(win_length):
    """scipy.signal.get_window('hann', N, fftbins=True)."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def pad_center(w, size):
    lpad = (size - len(w)) // 2
    out = np.zeros(size, dtype=w.dtype)
    out[lpad:lpad + len(w)] = w
    return out


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels, n_fft//2+1] float32, Slaney scale, area ('slaney') normalised."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(F32)


def forward_basis(filter_length, win_length):
    """[2*cutoff, filter_length] fp32: rows [Re; Im] of the DFT times the zero-centre-padded window."""
    k = np.arange(filter_length // 2 + 1)[:, None]
    n = np.arange(filter_length)[None, :]
    ang = 2.0 * np.pi * k * n / filter_length
    basis = np.vstack([np.cos(ang), -np.sin(ang)])             # fft(eye): exp(-2*pi*i*k*n/N)
    win = pad_center(hann_periodic(win_length), filter_length)
    return (basis.astype(F32) * win.astype(F32)[None, :]).astype(F32)


def stft_magnitude(y, filter_length, hop_length, win_length):
    """y [B, T] -> magnitude [B, filter_length//2+1, T//hop + 1] (fp32)."""
    y = np.asarray(y, dtype=F32)
    B, T = y.shape
    half = filter_length // 2
    yp = np.pad(y, ((0, 0), (half, half)), mode="reflect")
    n_frames = T // hop_length + 1
    idx = np.arange(n_frames)[:, None] * hop_length + np.arange(filter_length)[None, :]
    frames = yp[:, idx]                                       # [B, n_frames, N]
    fb = forward_basis(filter_length, win_length)             # [2c, N]
    spec = np.matmul(frames, fb.T)                            # fp32 dot products, like the conv
    c = filter_length // 2 + 1
    re, im = spec[..., :c], spec[..., c:]
    mag = np.sqrt(re * re + im * im).astype(F32)
    return np.ascontiguousarray(mag.transpose(0, 2, 1))


def mel_spectrogram(y, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                    sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0, clamp_val=1e-5, mel_basis=None):
    y = np.asarray(y, dtype=F32)
    assert y.min() >= -1.0 and y.max() <= 1.0                  # stft.py:191-192
    mag = stft_magnitude(y, filter_length, hop_length, win_length)
    if mel_basis is None:
        mel_basis = slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
    mel = np.matmul(mel_basis.astype(F32), mag)
    return np.log(np.maximum(mel, F32(clamp_val))).astype(F32)


# ---------------------------------------------------------------- phase / inverse / denoiser ----
def stft_transform(y, filter_length, hop_length, win_length):
    """(magnitude, phase) as STFT.transform(return_phase=True) (stft.py:99-111)."""
    y = np.asarray(y, dtype=F32)
    B, T = y.shape
    half = filter_length // 2
    yp = np.pad(y, ((0, 0), (half, half)), mode="reflect")
    n_frames = T // hop_length + 1
    idx = np.arange(n_frames)[:, None] * hop_length + np.arange(filter_length)[None, :]
    spec = np.matmul(yp[:, idx], forward_basis(filter_length, win_length).T)
    c = filter_length // 2 + 1
    re, im = spec[..., :c].transpose(0, 2, 1), spec[..., c:].transpose(0, 2, 1)
    return np.sqrt(re * re + im * im).astype(F32), np.arctan2(im, re).astype(F32)


def inverse_basis(filter_length, hop_length, win_length):
    """[2*cutoff, filter_length]: pinv(scale * fourier_basis).T times the window (stft.py:62-63, 74)."""
    N = filter_length
    scale = N / hop_length
    fb = np.fft.fft(np.eye(N))
    c = N // 2 + 1
    fb = np.vstack([np.real(fb[:c, :]), np.imag(fb[:c, :])])
    inv = np.linalg.pinv(scale * fb).T.astype(F32)
    win = pad_center(hann_periodic(win_length), N).astype(F32)
    return (inv * win[None, :]).astype(F32)


def stft_inverse(mag, phase, filter_length, hop_length, win_length):
    """STFT.inverse (stft.py:117-146): [B, c, frames] x2 -> [B, 1, (frames-1)*hop]."""
    mag, phase = np.asarray(mag, F32), np.asarray(phase, F32)
    B, c, frames = mag.shape
    N = filter_length
    rec = np.concatenate([mag * np.cos(phase), mag * np.sin(phase)], axis=1).astype(F32)     # [B, 2c, frames]
    ib = inverse_basis(N, hop_length, win_length)                                            # [2c, N]
    Y = np.einsum("bmn,mk->bkn", rec, ib).astype(F32)                                        # [B, N, frames]
    total = N + hop_length * (frames - 1)
    out = np.zeros((B, total), dtype=F32)
    wss = np.zeros(total, dtype=F32)
    win_sq = pad_center((hann_periodic(win_length) ** 2), N).astype(F32)
    for n in range(frames):
        out[:, n * hop_length:n * hop_length + N] += Y[:, :, n]
        wss[n * hop_length:n * hop_length + N] += win_sq
    nz = wss > np.finfo(np.float32).tiny
    out[:, nz] /= wss[nz]
    out *= F32(N) / F32(hop_length)
    return out[:, None, N // 2:-(N // 2)]


def denoise(audio, bias_spec, strength, filter_length, hop_length, win_length):
    """Denoiser.forward (denoiser.py:55-72) for one shared bias spectrum [c], or one per utterance [B, c]
    (speaker_dependant: ``bias_spec[speaker_ids]``, denoiser.py:65-66)."""
    mag, phase = stft_transform(audio, filter_length, hop_length, win_length)
    bias = np.asarray(bias_spec, dtype=F32)
    bias = bias[None] if bias.ndim == 1 else bias
    den = np.maximum(mag - bias[:, :, None] * F32(strength), 0).astype(F32)
    return stft_inverse(den, phase, filter_length, hop_length, win_length)
