"""STFT / mel frontend: oracle vs reference golden (CPU), HIP vs golden and oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import stft_oracle as so

MEL_TOL = 1e-4            # BASELINE.json: mel L_inf <= 1e-4


def _golden():
    return np.load(os.path.join(GOLDEN, "stft_mel.npz"))


def test_oracle_matches_reference_golden():
    g = _golden()
    mel = so.mel_spectrogram(g["y"])
    assert mel.shape == g["mel"].shape == (2, 80, 22050 // 256 + 1)
    assert np.abs(mel - g["mel"]).max() < MEL_TOL
    mag = so.stft_magnitude(g["y"], 1024, 256, 1024)
    scale = float(mag.max())
    assert np.abs(mag[:, ::32, :] - g["mag_rows"]).max() < 1e-6 * scale
    assert abs(float(mag.astype(np.float64).sum()) - float(g["mag_sum"])) < 1e-6 * float(g["mag_sum"])
    m8 = so.stft_magnitude(g["y"][:, :5000], 800, 200, 800)          # class defaults, non-power-of-two
    assert m8.shape == g["mag800"].shape and np.abs(m8 - g["mag800"]).max() < 1e-6 * float(m8.max())


def test_product_filterbank_matches_oracle_and_is_sane():
    """The filterbank is the one boundary with no reference pin (librosa absent): cross-check the two
    independent restatements and the properties the published algorithm guarantees."""
    from cookietts_amd.audio import slaney_mel_filterbank
    a = slaney_mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    b = so.slaney_mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert a.shape == (80, 513) and np.abs(a - b).max() < 1e-7
    assert np.allclose(a.sum(axis=1), _golden()["mel_basis_rowsum"], atol=1e-6)
    assert (a >= 0).all() and (a.max(axis=1) > 0).all()
    peaks = a.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()                                # centre bins strictly increase
    freqs = np.linspace(0, 11025, 513)
    assert a[:, freqs > 8000.0 + 22].sum() == 0                      # nothing above fmax


def test_reflect_pad_edge_cases():
    y = np.linspace(-1, 1, 700, dtype=np.float32)[None]
    mag = so.stft_magnitude(y, 1024, 256, 1024)                      # T < filter_length: reflect still valid (T > N/2)
    assert mag.shape == (1, 513, 700 // 256 + 1) and np.isfinite(mag).all()


@pytest.mark.gpu
def test_hip_mel_and_magnitude_match_golden(hip_lib_path):
    from cookietts_amd import TacotronSTFT
    g = _golden()
    taco = TacotronSTFT().cuda()
    y = torch.from_numpy(g["y"]).cuda()
    mel = taco.mel_spectrogram(y).cpu().numpy()
    assert mel.shape == g["mel"].shape
    print("mel Linf vs reference:", np.abs(mel - g["mel"]).max())
    assert np.abs(mel - g["mel"]).max() < MEL_TOL
    mag, phase = taco.stft_fn.transform(y, return_phase=False)
    assert phase is None
    mag = mag.cpu().numpy()
    scale = float(mag.max())
    assert np.abs(mag[:, ::32, :] - g["mag_rows"]).max() < 2e-6 * scale
    with pytest.raises(AssertionError):
        taco.mel_spectrogram(y * 3.0)                                # range assert of stft.py:191-192


@pytest.mark.gpu
@pytest.mark.parametrize("N,hop,T,B", [(800, 200, 5000, 2), (1024, 256, 600, 1), (1024, 256, 40000, 3), (2048, 512, 33333, 1)])
def test_hip_stft_matches_oracle_shapes(hip_lib_path, N, hop, T, B):
    from cookietts_amd import STFT
    rng = np.random.default_rng(T)
    y = np.clip(rng.standard_normal((B, T)).astype(np.float32) * 0.3, -1, 1)
    ref = so.stft_magnitude(y, N, hop, N)
    st = STFT(N, hop, N).cuda()
    mag, _ = st.transform(torch.from_numpy(y).cuda(), return_phase=False)
    mag = mag.cpu().numpy()
    assert mag.shape == ref.shape
    assert np.abs(mag - ref).max() < 2e-6 * float(ref.max()) + 1e-6
    if N == 800:
        g = _golden()
        m8, _ = st.transform(torch.from_numpy(g["y"][:, :5000]).cuda(), return_phase=False)
        assert np.abs(m8.cpu().numpy() - g["mag800"]).max() < 2e-6 * float(g["mag800"].max())


def _inv_golden():
    return np.load(os.path.join(GOLDEN, "stft_inverse.npz"))


def _phase_err(a, b, mag, thr=1e-2):
    d = np.abs(np.angle(np.exp(1j * (a.astype(np.float64) - b))))
    return float(d[mag > thr].max())


def test_oracle_inverse_and_denoiser_match_reference():
    g = _inv_golden()
    mag, ph = so.stft_transform(g["y"], 1024, 256, 1024)
    assert np.abs(mag - g["mag"]).max() < 1e-6 * float(g["mag"].max()) + 1e-6
    assert _phase_err(ph, g["phase"], g["mag"]) < 1e-3
    rt = so.stft_inverse(g["mag"], g["phase"], 1024, 256, 1024)
    assert rt.shape == g["roundtrip"].shape and np.abs(rt - g["roundtrip"]).max() < 1e-5
    assert np.abs(rt[:, 0] - g["y"]).max() < 1e-5                      # STFT -> ISTFT reconstructs the input
    dn = so.denoise(g["y"], g["bias_spec"], float(g["strength"]), 1024, 256, 1024)
    assert np.abs(dn - g["denoised"]).max() < 1e-5
    dn = so.denoise(g["y"], g["bias_spec_spk"][g["speaker_ids"]], float(g["strength_spk"]), 1024, 256, 1024)
    assert np.abs(dn - g["denoised_spk"]).max() < 1e-5                 # speaker-dependent bias (denoiser.py:65-66)


@pytest.mark.gpu
def test_hip_phase_inverse_denoiser_match_reference(hip_lib_path):
    from cookietts_amd import STFT, Denoiser
    g = _inv_golden()
    st = STFT(1024, 256, 1024).cuda()
    y = torch.from_numpy(g["y"]).cuda()
    mag, ph = st.transform(y, return_phase=True)
    assert np.abs(mag.cpu().numpy() - g["mag"]).max() < 2e-6 * float(g["mag"].max()) + 1e-6
    assert _phase_err(ph.cpu().numpy(), g["phase"], g["mag"]) < 1e-3
    rt = st.inverse(torch.from_numpy(g["mag"]).cuda(), torch.from_numpy(g["phase"]).cuda()).cpu().numpy()
    print("inverse Linf vs reference:", np.abs(rt - g["roundtrip"]).max())
    assert rt.shape == g["roundtrip"].shape and np.abs(rt - g["roundtrip"]).max() < MEL_TOL
    assert np.abs(st(y).cpu().numpy()[:, 0] - g["y"]).max() < MEL_TOL                  # forward(): round trip

    class _Vocoder(torch.nn.Module):                 # the Denoiser only needs .parameters() and .infer()
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def infer(self, mel, sigma=1.0):
            return torch.rand(mel.shape[0], mel.shape[2] * 256, device=mel.device) * 0.1
    den = Denoiser(_Vocoder().cuda(), sampling_rate=22050, filter_length=1024, hop_length=256, win_length=1024,
                   n_mel_channels=80)
    assert den.bias_spec.shape == (1, 513, 1) and torch.isfinite(den.bias_spec).all()
    den.bias_spec = torch.from_numpy(g["bias_spec"]).cuda()[None, :, None]
    dn = den(y, strength=float(g["strength"])).cpu().numpy()
    print("denoiser Linf vs reference:", np.abs(dn - g["denoised"]).max())
    assert dn.shape == g["denoised"].shape and np.abs(dn - g["denoised"]).max() < MEL_TOL
    # speaker-dependent mode: bias_spec [n_speakers, bins, 1], one row picked per utterance (denoiser.py:29-45, 65-66)
    den.bias_spec = torch.from_numpy(g["bias_spec_spk"]).cuda()[:, :, None]
    ids = torch.from_numpy(g["speaker_ids"]).cuda()
    dn = den(y, speaker_ids=ids, strength=float(g["strength_spk"])).cpu().numpy()
    print("speaker-dependent denoiser Linf vs reference:", np.abs(dn - g["denoised_spk"]).max())
    assert np.abs(dn - g["denoised_spk"]).max() < MEL_TOL
    shared = den(y, speaker_ids=None, strength=float(g["strength_spk"])).cpu().numpy()       # ids None -> row 0 for all
    assert np.abs(shared[1] - dn[1]).max() < 1e-6 and np.abs(shared[0] - dn[0]).max() > 1e-4


@pytest.mark.gpu
def test_hip_speaker_dependent_denoiser_constructor(hip_lib_path):
    """Denoiser(speaker_dependant=True) on a multispeaker ax vocoder: one bias spectrum per speaker id, taken from
    WN[0].WN.speaker_embed.num_embeddings like denoiser.py:33-34."""
    from cookietts_amd import Denoiser, synthetic
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEGLOW_AX_CONFIGS["notebook_toy"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=5)))
    m = m.cuda().eval()
    den = Denoiser(m, sampling_rate=6400, n_mel_channels=cfg["n_mel_channels"], n_frames=12, speaker_dependant=True)
    assert den.bias_spec.shape == (512, 6400 // 40 // 2 + 1, 1) and torch.isfinite(den.bias_spec).all()
    assert (den.bias_spec[3] - den.bias_spec[4]).abs().max() > 0                         # per-speaker, not a copy
    audio = m.infer(torch.from_numpy(synthetic.synthetic_mel(2, 9, cfg["n_mel_channels"])).cuda(),
                    speaker_ids=torch.tensor([3, 4]).cuda(), sigma=0.5, return_CPU=False)
    out = den(audio, speaker_ids=torch.tensor([3, 4]).cuda(), strength=0.2)
    assert out.shape == (2, 1, audio.shape[1] // 16 * 16) and torch.isfinite(out).all()



def test_slaney_filterbank_against_closed_form_numbers():
    """Third witness for the one unpinned boundary (librosa.filters.mel is absent and unpinned, stft.py:163-164):
    numbers worked out here from the PUBLISHED constants of the Slaney scale (200/3 Hz per mel below 1 kHz; above,
    27 mels per factor 6.4), not by calling either restatement's helpers.  Both restatements must hit them."""
    import math
    from cookietts_amd import audio
    sr, n_fft, n_mels, fmin, fmax = 22050, 1024, 80, 0.0, 8000.0
    df = sr / n_fft                                                     # 21.533203125 Hz per bin
    top = 15.0 + 27.0 * math.log(fmax / 1000.0) / math.log(6.4)         # mel(8000 Hz) = 15 + 27 * 1.120209 = 45.24564
    assert abs(top - 45.245640) < 1e-5
    step = top / (n_mels + 1)

    def edge(k):                                                        # k-th of the 82 mel-spaced edges, in Hz
        m = k * step
        return m * 200.0 / 3.0 if m < 15.0 else 1000.0 * math.exp((m - 15.0) * math.log(6.4) / 27.0)
    assert abs(edge(1) - 37.2392) < 1e-3 and abs(edge(81) - 8000.0) < 1e-9
    k15 = 15.0 / step                                                   # the scale turns logarithmic at edge 26.85
    assert 26 < k15 < 27 and abs(edge(27) - 1000.0 * 6.4 ** ((27 * step - 15.0) / 27.0)) < 1e-9
    for name, fb in (("product", audio.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)),
                     ("oracle", so.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax))):
        fb = np.asarray(fb, dtype=np.float64)
        assert fb.shape == (80, 513), name
        for i in (0, 1, 13, 26, 27, 40, 60, 79):                        # linear region, the knee, log region, last
            lo, ce, hi = edge(i), edge(i + 1), edge(i + 2)
            for k in range(513):
                f = k * df
                w = max(0.0, min((f - lo) / (ce - lo), (hi - f) / (hi - ce))) * 2.0 / (hi - lo)
                assert abs(fb[i, k] - w) <= 1e-7 * max(1.0, w) + 1e-9, (name, i, k, fb[i, k], w)
            # support is exactly the open interval (lo, hi); the triangle has unit area (area normalisation):
            # the bin sum times the bin width approximates it to within the kink error of a triangle sampled at df
            nz = np.nonzero(fb[i])[0]
            assert nz.min() * df > lo - 1e-9 and nz.max() * df < hi + 1e-9, (name, i)
            area = fb[i].sum() * df
            assert abs(area - 1.0) < 0.5 * df / (hi - lo) + 1e-6, (name, i, area)
        # two hand-checked entries: filter 0 at bin 1 (rising edge from 0 Hz) and the peak height of the last filter
        assert abs(fb[0, 1] - (df / edge(1)) * 2.0 / edge(2)) < 1e-9, name
        assert fb[79].max() <= 2.0 / (edge(81) - edge(79)) + 1e-12, name
