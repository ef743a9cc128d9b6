"""STFT / mel frontend on the MI355X HIP path.

Host-side mirror of ``/root/reference/CookieTTS/utils/audio/stft.py``: ``STFT`` (:44-151) and
``TacotronSTFT`` (:154-207) with the same constructor arguments, buffers (``forward_basis``,
``mel_basis``) and methods (``transform``, ``mel_spectrogram``).  The transform itself runs in the
C-ABI HIP library (``ctts_stft_mel_f32``); there is no CPU fallback.

The reference takes its mel filterbank from ``librosa.filters.mel`` (stft.py:163-164, a
third-party dependency that is not vendored); ``slaney_mel_filterbank`` restates that published
algorithm (Slaney scale, area normalisation) - constructor-time host code, like the reference's.
Not built yet (next row): ``STFT.inverse`` / phase output (used by the Denoiser).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["STFT", "TacotronSTFT", "slaney_mel_filterbank"]


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * 200.0 / 3.0
    log = 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0))
    return np.where(m >= 15.0, log, lin)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Triangular filters on the Slaney mel scale, each scaled by 2 / (f_hi - f_lo)."""
    if fmax is None:
        fmax = sr / 2.0
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, freqs.size))
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        fb[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


class STFT(torch.nn.Module):
    def __init__(self, filter_length=800, hop_length=200, win_length=800, window='hann', dtype=torch.float32):
        super().__init__()
        if window != 'hann':
            raise NotImplementedError("only the 'hann' window is built")
        if dtype != torch.float32:
            raise NotImplementedError("stft_dtype other than float32 is not built")
        assert filter_length >= win_length
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length
        self.window = window
        cutoff = filter_length // 2 + 1
        fourier_basis = np.fft.fft(np.eye(filter_length))
        fourier_basis = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
        forward_basis = torch.FloatTensor(fourier_basis[:, None, :])
        n = np.arange(win_length)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)            # periodic Hann (fftbins=True)
        lpad = (filter_length - win_length) // 2
        fft_window = np.zeros(filter_length)
        fft_window[lpad:lpad + win_length] = win
        forward_basis *= torch.from_numpy(fft_window).float()
        self.register_buffer('forward_basis', forward_basis.float())
        self._mel_basis_for_pack = None
        self._packed = None
        self._ws = {}

    def _apply(self, fn, *a, **kw):
        self._packed, self._ws = None, {}
        return super()._apply(fn, *a, **kw)

    def _c_config(self, n_mel=0, clamp=1e-5):
        return _lib.StftConfig(filter_length=self.filter_length, hop_length=self.hop_length,
                               win_length=self.win_length, n_mel_channels=n_mel, clamp_val=clamp)

    def _run(self, y, want_mag, mel_basis=None, clamp=1e-5):
        if y.device.type != 'cuda':
            raise _lib.HipLibraryError("STFT HIP path needs GPU tensors (no CPU fallback)")
        lib = _lib.lib()
        device = y.device
        n_mel = 0 if mel_basis is None else mel_basis.shape[0]
        cfg = self._c_config(n_mel, clamp)
        y = y.detach().float().contiguous()
        B, T = y.shape
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            key = (device, n_mel)
            if self._packed is None or self._packed[0] != key:
                nbytes = lib.ctts_stft_packed_bytes(C.byref(cfg))
                if nbytes == 0:
                    raise _lib.HipLibraryError("unsupported STFT config: " + lib.ctts_last_error().decode())
                blob = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                fb = self.forward_basis.detach().float().squeeze(1).contiguous().to(device)
                mb = None if mel_basis is None else mel_basis.detach().float().contiguous().to(device)
                _lib.check(lib.ctts_stft_pack(C.byref(cfg), _lib.ptr(fb), _lib.ptr(mb), _lib.ptr(blob), stream),
                           "ctts_stft_pack")
                torch.cuda.current_stream(device).synchronize()
                self._packed = (key, blob)
            blob = self._packed[1]
            wkey = (device, B, T)
            ws = self._ws.get(wkey)
            if ws is None:
                nbytes = lib.ctts_stft_workspace_bytes(C.byref(cfg), B, T)
                if nbytes == 0:
                    raise _lib.HipLibraryError("STFT workspace query failed: " + lib.ctts_last_error().decode())
                self._ws = {wkey: torch.zeros(nbytes // 4, dtype=torch.float32, device=device)}
                ws = self._ws[wkey]
            frames = T // self.hop_length + 1
            cutoff = self.filter_length // 2 + 1
            mag = torch.empty(B, cutoff, frames, dtype=torch.float32, device=device) if want_mag else None
            mel = torch.empty(B, n_mel, frames, dtype=torch.float32, device=device) if n_mel else None
            _lib.check(lib.ctts_stft_mel_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(y), _lib.ptr(mag), _lib.ptr(mel),
                                            B, T, _lib.ptr(ws), ws.numel() * 4, stream), "ctts_stft_mel_f32")
        return mag, mel

    def transform(self, input_data, return_phase=True):
        """[B, T] -> (magnitude [B, N/2+1, T//hop+1], phase)   (stft.py:113-115)"""
        if return_phase:
            raise NotImplementedError("phase output is not built yet (needed only by STFT.inverse / Denoiser)")
        mag, _ = self._run(input_data, want_mag=True)
        return mag, None

    def inverse(self, magnitude, phase):
        raise NotImplementedError("STFT.inverse is a next-row item (SURVEY.md §8f)")


class TacotronSTFT(torch.nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                 sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0, clamp_val=1e-5, stft_dtype=torch.float32):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.clip_val = clamp_val
        self.stft_fn = STFT(filter_length, hop_length, win_length, dtype=stft_dtype)
        mel_basis = slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer('mel_basis', torch.from_numpy(mel_basis).float())

    @torch.no_grad()
    def mel_spectrogram(self, y):
        """y [B, T] in [-1, 1] -> log-mel [B, n_mel, T//hop + 1]   (stft.py:180-207)"""
        assert torch.min(y) >= -1., f'Tensor.min() of {torch.min(y).item()} is less than -1.0'
        assert torch.max(y) <= 1., f'Tensor.max() of {torch.max(y).item()} is greater than 1.0'
        _, mel = self.stft_fn._run(y, want_mag=False, mel_basis=self.mel_basis, clamp=self.clip_val)
        return mel
