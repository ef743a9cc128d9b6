"""STFT / mel frontend on the MI355X HIP path.

Host-side mirror of ``/root/reference/CookieTTS/utils/audio/stft.py``: ``STFT`` (:44-151) and
``TacotronSTFT`` (:154-207) with the same constructor arguments, buffers (``forward_basis``,
``mel_basis``) and methods (``transform``, ``mel_spectrogram``).  The transform itself runs in the
C-ABI HIP library (``ctts_stft_mel_f32``); there is no CPU fallback.

The reference takes its mel filterbank from ``librosa.filters.mel`` (stft.py:163-164, a
third-party dependency that is not vendored); ``slaney_mel_filterbank`` restates that published
algorithm (Slaney scale, area normalisation) - constructor-time host code, like the reference's.
``STFT.transform(return_phase=True)`` / ``STFT.inverse`` (stft.py:99-146) and ``Denoiser``
(``_4_mtw/waveglow/denoiser.py:7-72``) are built on the same library (``ctts_stft_transform_f32`` /
``ctts_stft_inverse_f32``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["STFT", "TacotronSTFT", "Denoiser", "slaney_mel_filterbank"]


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
        scale = filter_length / hop_length
        inverse_basis = torch.FloatTensor(np.linalg.pinv(scale * fourier_basis).T[:, None, :])
        inverse_basis *= torch.from_numpy(fft_window).float()
        self.register_buffer('inverse_basis', inverse_basis.float())
        self._window_sq = torch.from_numpy((fft_window ** 2).astype(np.float32))   # audio_processing.py:47-50
        self._mel_basis_for_pack = None
        self._packed = {}            # (device, n_mel) -> packed blob (forward, inverse and mel parts)
        self._ws = {}                # (device, n_mel, B, T) -> workspace sized for exactly that config

    def _apply(self, fn, *a, **kw):
        self._packed, self._ws = {}, {}
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
            if key not in self._packed:
                nbytes = lib.ctts_stft_packed_bytes(C.byref(cfg))
                if nbytes == 0:
                    raise _lib.HipLibraryError("unsupported STFT config: " + lib.ctts_last_error().decode())
                blob = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                fb = self.forward_basis.detach().float().squeeze(1).contiguous().to(device)
                mb = None if mel_basis is None else mel_basis.detach().float().contiguous().to(device)
                _lib.check(lib.ctts_stft_pack(C.byref(cfg), _lib.ptr(fb), _lib.ptr(mb), _lib.ptr(blob), stream),
                           "ctts_stft_pack")
                ib = self.inverse_basis.detach().float().squeeze(1).contiguous().to(device)
                wsq = self._window_sq.to(device)
                _lib.check(lib.ctts_stft_pack_inverse(C.byref(cfg), _lib.ptr(ib), _lib.ptr(wsq), _lib.ptr(blob), stream),
                           "ctts_stft_pack_inverse")
                torch.cuda.current_stream(device).synchronize()
                self._packed[key] = blob
            blob = self._packed[key]
            ws = self._workspace(cfg, device, n_mel, B, T)
            frames = T // self.hop_length + 1
            cutoff = self.filter_length // 2 + 1
            mag = torch.empty(B, cutoff, frames, dtype=torch.float32, device=device) if want_mag else None
            mel = torch.empty(B, n_mel, frames, dtype=torch.float32, device=device) if n_mel else None
            _lib.check(lib.ctts_stft_mel_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(y), _lib.ptr(mag), _lib.ptr(mel),
                                            B, T, _lib.ptr(ws), ws.numel() * 4, stream), "ctts_stft_mel_f32")
        return mag, mel

    def _blob_ws(self, device, B, T):
        """Packed blob + workspace for (B, T) without running anything (shared by transform/inverse)."""
        lib = _lib.lib()
        cfg = self._c_config(0)
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            key = (device, 0)
            if key not in self._packed:
                self._run(torch.zeros(1, max(self.filter_length, self.hop_length * 2), device=device), want_mag=True)
            ws = self._workspace(cfg, device, 0, B, T)
        return cfg, self._packed[key], ws, stream

    def _workspace(self, cfg, device, n_mel, B, T):
        """Workspace for exactly (n_mel, B, T): the size query depends on n_mel, so it is part of the key; the two
        most recent geometries are kept (mel_spectrogram and transform/inverse alternate on one object)."""
        wkey = (device, n_mel, B, T)
        ws = self._ws.pop(wkey, None)
        if ws is None:
            nbytes = _lib.lib().ctts_stft_workspace_bytes(C.byref(cfg), B, T)
            if nbytes == 0:
                raise _lib.HipLibraryError("STFT workspace query failed: " + _lib.lib().ctts_last_error().decode())
            ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            while len(self._ws) >= 2:
                self._ws.pop(next(iter(self._ws)))
        self._ws[wkey] = ws
        return ws

    def transform(self, input_data, return_phase=True):
        """[B, T] -> (magnitude [B, N/2+1, T//hop+1], phase or None)   (stft.py:99-115)"""
        if not return_phase:
            mag, _ = self._run(input_data, want_mag=True)
            return mag, None
        if input_data.device.type != 'cuda':
            raise _lib.HipLibraryError("STFT HIP path needs GPU tensors (no CPU fallback)")
        device = input_data.device
        y = input_data.detach().float().contiguous()
        B, T = y.shape
        cfg, blob, ws, stream = self._blob_ws(device, B, T)
        frames, cutoff = T // self.hop_length + 1, self.filter_length // 2 + 1
        mag = torch.empty(B, cutoff, frames, dtype=torch.float32, device=device)
        phase = torch.empty_like(mag)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().ctts_stft_transform_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(y), _lib.ptr(mag),
                                                         _lib.ptr(phase), B, T, _lib.ptr(ws), ws.numel() * 4, stream),
                       "ctts_stft_transform_f32")
        return mag, phase

    def inverse(self, magnitude, phase, _bias_spec=None, _strength=0.0):
        """(magnitude, phase) [B, N/2+1, frames] -> audio [B, 1, (frames-1)*hop]   (stft.py:117-146)"""
        device = magnitude.device
        if device.type != 'cuda':
            raise _lib.HipLibraryError("STFT HIP path needs GPU tensors (no CPU fallback)")
        mag = magnitude.detach().float().contiguous()
        ph = phase.detach().to(device).float().contiguous()
        B, cutoff, frames = mag.shape
        T = (frames - 1) * self.hop_length
        cfg, blob, ws, stream = self._blob_ws(device, B, max(T, self.filter_length))
        out = torch.empty(B, 1, T, dtype=torch.float32, device=device)
        bias, bstride = None, 0
        if _bias_spec is not None:      # [cutoff] shared, or [B, cutoff] one spectrum per utterance
            bias = _bias_spec.detach().to(device).float().reshape(-1, cutoff).contiguous()
            assert bias.shape[0] in (1, B), (tuple(bias.shape), B)
            bstride = cutoff if bias.shape[0] == B and B > 1 else 0
        with torch.cuda.device(device):
            _lib.check(_lib.lib().ctts_stft_inverse_bias_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(mag), _lib.ptr(ph),
                                                            _lib.ptr(bias), bstride, float(_strength), _lib.ptr(out), B,
                                                            frames, _lib.ptr(ws), ws.numel() * 4, stream),
                       "ctts_stft_inverse_bias_f32")
        return out

    def forward(self, input_data):
        self.magnitude, self.phase = self.transform(input_data)
        return self.inverse(self.magnitude, self.phase)


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


class Denoiser(torch.nn.Module):
    """Removes the vocoder's bias spectrum from generated audio (``_4_mtw/waveglow/denoiser.py:7-72``).

    Same constructor arguments; ``stft_device`` is ignored (everything runs on the vocoder's GPU).  The bias
    subtraction + clamp is fused into the inverse STFT's recombination kernel.  ``speaker_dependant=True`` keeps one
    bias spectrum per speaker id (denoiser.py:29-45) and ``forward(..., speaker_ids=...)`` subtracts each utterance's
    own (denoiser.py:65-66)."""

    def __init__(self, waveglow, sampling_rate=48000, filter_length=None, hop_length=None, win_length=None,
                 n_mel_channels=160, n_frames=20, mu=0, var=0.01, wg_sigma=0.01, stft_device='cpu',
                 speaker_dependant=False, speaker_id=0):
        super().__init__()
        filter_length = filter_length or sampling_rate // 40
        win_length = win_length or sampling_rate // 40
        hop_length = hop_length or sampling_rate // 400
        p = next(waveglow.parameters())
        self.stft = STFT(filter_length=filter_length, hop_length=hop_length, win_length=win_length).to(p.device)
        mel_input = torch.randn((1, n_mel_channels, n_frames), dtype=p.dtype, device=p.device) * float(var) + float(mu)

        def infer(ids):
            mel = mel_input.expand(ids.shape[0], -1, -1).contiguous()
            for kw in ({"speaker_ids": ids}, {"speaker_id": ids}, {}):      # ax core / glow.py:314 / single-speaker stubs
                try:
                    return waveglow.infer(mel, sigma=wg_sigma, **kw)
                except TypeError as e:
                    if "unexpected keyword" not in str(e):
                        raise
            raise TypeError("vocoder.infer accepts neither speaker_ids nor speaker_id nor a bare call")
        with torch.no_grad():
            if speaker_dependant:      # denoiser.py:29-45: one vocoder pass per speaker, batched here
                if hasattr(waveglow, 'speaker_embed'):
                    n_speakers = waveglow.speaker_embed.num_embeddings
                elif hasattr(waveglow, 'WN') and hasattr(waveglow.WN[0], 'WN') and hasattr(waveglow.WN[0].WN, 'speaker_embed'):
                    n_speakers = waveglow.WN[0].WN.speaker_embed.num_embeddings
                else:
                    n_speakers = 1
                chunks = [infer(torch.arange(i, min(i + 64, n_speakers), device=p.device, dtype=torch.int64))
                          for i in range(0, n_speakers, 64)]
                bias_audio = torch.cat([c.to(device=p.device, dtype=torch.float) for c in chunks], dim=0)
            else:
                bias_audio = infer(torch.tensor([speaker_id], device=p.device, dtype=torch.int64))
                bias_audio = bias_audio.to(device=p.device, dtype=torch.float)
            assert torch.isfinite(bias_audio).all(), 'Inf/NaN elements found in Vocoder Output'
            bias_spec, _ = self.stft.transform(bias_audio, return_phase=False)
        self.register_buffer('bias_spec', bias_spec.mean(dim=2, keepdim=True))      # [n_speakers or 1, cutoff, 1]

    @torch.no_grad()
    def forward(self, wg_audio, speaker_ids=None, strength=0.1):
        audio = wg_audio.to(self.bias_spec.device).float()
        audio_spec, audio_angles = self.stft.transform(audio, return_phase=True)
        if speaker_ids is None or self.bias_spec.shape[0] == 1:
            bias = self.bias_spec[0]
        else:
            bias = self.bias_spec[speaker_ids.to(self.bias_spec.device)]            # [B, cutoff, 1]
        return self.stft.inverse(audio_spec, audio_angles, _bias_spec=bias, _strength=strength)
