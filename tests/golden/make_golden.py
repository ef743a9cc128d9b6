#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py [waveglow]

What is captured is data only - inputs and the reference's outputs.  Weights are not
stored: they come from the deterministic numpy recipe ``cookietts_amd.synthetic``
(seed recorded in each fixture), loaded into the reference model through its own
``load_state_dict``.

Oracle-only shims, living only here:
  * ``torch.cuda.FloatTensor`` aliased to the CPU ``torch.FloatTensor`` so that the
    reference's early-output noise draw (glow.py:343-346) runs without a GPU.  The noise
    it draws is recorded by replaying the same torch RNG sequence from the same seed.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from cookietts_amd import synthetic  # noqa: E402


def _ref_waveglow(cfg, sd_np):
    from CookieTTS._4_mtw.waveglow import glow
    model = glow.WaveGlow(**cfg)
    missing = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()},
                                    strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.eval()


def _run_ref_infer(model, cfg, mel, sigma, torch_seed, speaker_id=None):
    """Run the reference's own ``infer`` and return (wave, z_scaled[B,G,L]) with the noise it drew."""
    B, _, F = mel.shape
    G = cfg["n_group"]
    L = F * cfg["hop_length"] // G
    chans = synthetic.waveglow_flow_channels(cfg)
    n_rem_final = chans[-1][0]
    esize = cfg["n_early_size"]
    early_flows = [k for k in reversed(range(cfg["n_flows"])) if k % cfg["n_early_every"] == 0 and k > 0]

    # replay of the RNG sequence infer() will consume (glow.py:326 then :343-346 per early flow)
    torch.manual_seed(torch_seed)
    z = np.zeros((B, G, L), dtype=np.float32)
    lo = len(early_flows) * esize
    assert lo + n_rem_final == G
    z[:, lo:] = torch.ones(B, n_rem_final, L).normal_(std=sigma).numpy()
    for _ in early_flows:
        lo -= esize
        z[:, lo:lo + esize] = (sigma * torch.FloatTensor(B, esize, L).normal_()).numpy()

    saved = getattr(torch.cuda, "FloatTensor", None)
    torch.cuda.FloatTensor = torch.FloatTensor          # shim: CPU stand-in for the CUDA-only ctor
    try:
        torch.manual_seed(torch_seed)
        with torch.no_grad():
            wave = model.infer(torch.from_numpy(mel.copy()), speaker_id=speaker_id, sigma=sigma)
    finally:
        if saved is not None:
            torch.cuda.FloatTensor = saved
    return wave.numpy().astype(np.float32), z


def make_waveglow(full_length=False, options=False):
    torch.set_num_threads(8)
    cases = [
        # name, config key, batch, frames, sigma, seed
        ("toy", "toy", 2, 12, 1.0, 11),
        ("toy_early", "toy_early", 2, 9, 0.8, 12),          # exercises early-output re-injection
        ("small", "small", 1, 200, 0.6, 1234),              # BASELINE config 1, end to end
        ("full_short", "full", 1, 16, 0.6, 1234),           # BASELINE config 2 topology, short mel
    ]
    if options:
        # glow.py options: WN speaker embeddings + ReZero (glow.py:127-133, 193-196, 211-212), grouped upsampling (:241)
        cases = [("toy_spk_rezero", "toy_spk_rezero", 3, 10, 0.8, 31), ("toy_simple", "toy_simple", 2, 7, 1.0, 32),
                 # hop_length / n_group away from the benchmark's 256 / 8
                 ("toy_hop512_g16", "toy_hop512_g16", 2, 7, 0.8, 33), ("toy_hop384_g12", "toy_hop384_g12", 2, 9, 0.9, 34)]
        only = sys.argv[2:]
        if only:
            cases = [c for c in cases if c[0] in only]
    if full_length:
        # BASELINE config 2 at the metric's utterance length: one 80x900 mel through the 12x512 model
        # (~70 s of CPU here).  Only the reference's waveform and the noise it drew are stored.
        cases = [("full_len", "full", 1, 900, 0.6, 4321)]
    for name, key, B, F, sigma, seed in cases:
        cfg = synthetic.WAVEGLOW_CONFIGS[key]
        sd = synthetic.waveglow_state_dict(cfg, seed=seed)
        model = _ref_waveglow(cfg, sd)
        mel = synthetic.synthetic_mel(B, F, cfg["n_mel_channels"], seed=seed)
        ids = np.array([7, 300, 511, 0][:B], np.int64) if cfg["WN_config"]["speaker_embed_dim"] else None
        wave, z = _run_ref_infer(model, cfg, mel, sigma, torch_seed=seed,
                                 speaker_id=None if ids is None else torch.from_numpy(ids))
        assert wave.shape == (B, F * cfg["hop_length"]) and np.isfinite(wave).all()
        extras = {} if ids is None else {"speaker_ids": ids}
        if name in ("toy", "small"):
            # per-stage intermediates from the reference's own sub-modules (flow n_flows-1)
            with torch.no_grad():
                up = model.upsample(torch.from_numpy(mel))
                up = up[:, :, :-(cfg["win_length"] - cfg["hop_length"])]
                sp = up.unfold(2, cfg["n_group"], cfg["n_group"]).permute(0, 2, 1, 3)
                sp = sp.contiguous().view(sp.size(0), sp.size(1), -1).permute(0, 2, 1)
                k = cfg["n_flows"] - 1
                n_rem = synthetic.waveglow_flow_channels(cfg)[k][0]
                a = torch.from_numpy(z[:, cfg["n_group"] - n_rem:, :].copy())
                b_, s_ = model.WN[k](a[:, :n_rem // 2], sp)
            extras = dict(spect_head=sp[:, :, :64].numpy().astype(np.float32),
                          spect_checksum=np.float64(sp.double().sum().item()),
                          wn_last_b=b_.numpy().astype(np.float32),
                          wn_last_s=s_.numpy().astype(np.float32))
        path = os.path.join(HERE, f"waveglow_{name}.npz")
        if full_length:
            extras["mel_recipe"] = "cookietts_amd.synthetic.synthetic_mel(B, F, n_mel, seed=seed)"
            np.savez_compressed(path, config_key=key, seed=seed, sigma=np.float32(sigma), B=B, F=F,
                                z_scaled=z, wave=wave, **extras)
        else:
            np.savez_compressed(path, config_key=key, seed=seed, sigma=np.float32(sigma), mel=mel,
                                z_scaled=z, wave=wave, **extras)
        rms = float(np.sqrt(np.mean(wave.astype(np.float64) ** 2)))
        print(f"[golden] {name}: wave {wave.shape} rms={rms:.4f} max={np.abs(wave).max():.3f} -> "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def _stub_audio_deps():
    """librosa / iso226 are not installed: minimal stand-ins so the reference's stft.py imports.
    `mel` is the Slaney restatement from the oracle (PARITY UNPINNED there, see its header)."""
    import types
    from oracle import stft_oracle as so
    librosa = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    filters = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1, **kw):
        return so.pad_center(np.asarray(data), size)

    def tiny(x):
        return np.finfo(np.asarray(x).dtype if np.issubdtype(np.asarray(x).dtype, np.floating) else np.float32).tiny

    def normalize(x, norm=None, **kw):
        assert norm is None
        return x
    util.pad_center, util.tiny, util.normalize = pad_center, tiny, normalize
    filters.mel = lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None: so.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    librosa.util, librosa.filters = util, filters
    sys.modules.update({"librosa": librosa, "librosa.util": util, "librosa.filters": filters})


def make_stft():
    _stub_audio_deps()
    from CookieTTS.utils.audio.stft import STFT, TacotronSTFT
    rng = np.random.default_rng(77)
    B, T = 2, 22050
    t = np.arange(T) / 22050.0
    y = np.stack([0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3300 * t + 1.0),
                  0.5 * np.sin(2 * np.pi * (100 + 2000 * t) * t)]).astype(np.float32)
    y += 0.05 * rng.standard_normal((B, T)).astype(np.float32)
    y = np.clip(y, -1.0, 1.0)
    taco = TacotronSTFT()                                    # defaults: 1024/256/1024, 80 mel, 22050 Hz, 0-8000 Hz
    with torch.no_grad():
        mel = taco.mel_spectrogram(torch.from_numpy(y)).numpy()
        mag, _ = taco.stft_fn.transform(torch.from_numpy(y), return_phase=False)
        small = STFT(filter_length=800, hop_length=200, win_length=800)     # the class defaults (stft.py:48)
        y2 = y[:, :5000]
        mag2, _ = small.transform(torch.from_numpy(y2), return_phase=False)
    path = os.path.join(HERE, "stft_mel.npz")
    np.savez_compressed(path, y=y, mel=mel.astype(np.float32), mag_rows=mag.numpy()[:, ::32, :].astype(np.float32),
                        mag_sum=np.float64(mag.double().sum().item()),
                        mel_basis_rowsum=taco.mel_basis.numpy().sum(axis=1).astype(np.float32),
                        mag800=mag2.numpy().astype(np.float32))
    print(f"[golden] stft_mel: mel {mel.shape} mag {tuple(mag.shape)} -> {os.path.getsize(path) / 1024:.0f} KiB")

    # phase / inverse / denoiser (the reference Denoiser with a FIXED bias spectrum: its constructor draws
    # random mels and vocoder noise, which is not what is being pinned)
    from CookieTTS._4_mtw.waveglow import denoiser as ref_den
    N, hop = 1024, 256
    st = STFT(N, hop, N)
    y3 = y[:, :8192]
    with torch.no_grad():
        m3, p3 = st.transform(torch.from_numpy(y3), return_phase=True)
        rt = st.inverse(m3, p3)

        class _FakeVocoder(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.zeros(1))

            def infer(self, mel, speaker_ids=None, sigma=1.0):
                return torch.zeros(mel.shape[0], mel.shape[2] * hop)
        den = ref_den.Denoiser(_FakeVocoder(), sampling_rate=22050, filter_length=N, hop_length=hop, win_length=N,
                               n_mel_channels=80)
        bias = (np.abs(rng.standard_normal(N // 2 + 1)) * 0.5).astype(np.float32)
        den.bias_spec = torch.from_numpy(bias)[None, :, None]
        dn = den(torch.from_numpy(y3), strength=0.3)
        # speaker-dependent mode (denoiser.py:29-45, 65-66): one bias spectrum per speaker id, picked per utterance
        bias_spk = (np.abs(rng.standard_normal((3, N // 2 + 1))) * 0.5).astype(np.float32)
        den.bias_spec = torch.from_numpy(bias_spk)[:, :, None]
        spk = np.array([2, 0], np.int64)
        dn_spk = den(torch.from_numpy(y3), speaker_ids=torch.from_numpy(spk), strength=0.45)
    path = os.path.join(HERE, "stft_inverse.npz")
    np.savez_compressed(path, y=y3, mag=m3.numpy().astype(np.float32), phase=p3.numpy().astype(np.float32),
                        roundtrip=rt.numpy().astype(np.float32), bias_spec=bias, strength=np.float32(0.3),
                        denoised=dn.numpy().astype(np.float32), bias_spec_spk=bias_spk, speaker_ids=spk,
                        strength_spk=np.float32(0.45), denoised_spk=dn_spk.numpy().astype(np.float32))
    print(f"[golden] stft_inverse: roundtrip {tuple(rt.shape)} max err vs input "
          f"{float(np.abs(rt.numpy()[:, 0] - y3).max()):.2e} -> {os.path.getsize(path) / 1024:.0f} KiB")


def _ref_waveflow(cfg, sd_np):
    """Import the reference's ax core (needs librosa / iso226 stand-ins and np.product)."""
    import types
    _stub_audio_deps()
    if "iso226" not in sys.modules:
        iso = types.ModuleType("iso226")
        iso.iso226_spl_itpl = lambda *a, **k: None
        sys.modules["iso226"] = iso
    if not hasattr(np, "product"):
        np.product = np.prod
    from CookieTTS._4_mtw.waveglow import efficient_model_ax as ax
    model = ax.WaveGlow(**cfg)
    res = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.eval()


def make_waveflow(full_length=False):
    import copy
    torch.set_num_threads(8)
    cases = [("toy", "toy", 2, 6, 0.7, 5), ("toy_odd", "toy", 1, 11, 1.0, 6), ("full_short", "full", 1, 5, 0.6, 1234),
             # SURVEY 8f.4 option set: speaker ids, cond stacks, separable in-layers, logvar channels, de-emphasis
             ("author_toy", "author_toy", 2, 6, 0.7, 3), ("author_short", "author", 1, 3, 0.6, 4),
             # the UnTTS notebook's WaveFlow: shift_spect / scale_spect on entry
             ("untts_toy", "untts_toy", 2, 7, 0.8, 8),
             # merge_res_skip + GLU on the dense 2-D core; merge_res_skip + GSIRRU on the separable C = 128 core
             ("toy_merge", "toy_merge", 2, 5, 0.7, 9), ("author_toy_gate", "author_toy_gate", 1, 5, 0.7, 10),
             ("toy_groupconv", "toy_groupconv", 2, 5, 0.7, 11),
             ("toy_wn_tconv", "toy_wn_tconv", 2, 6, 0.7, 12), ("toy_wn_tconv_crop", "toy_wn_tconv_crop", 2, 7, 0.7, 13),
             # InvertibleConv1x1 / PermuteHeight mixing in both orders with early outputs on the 2-D core
             ("toy_conv_early", "toy_conv_early", 2, 6, 0.7, 14),
             ("toy_permute_mixfirst_early", "toy_permute_mixfirst_early", 1, 5, 0.8, 15),
             ("toy_conv_mixlast", "toy_conv_mixlast", 2, 5, 0.7, 16),
             ("toy_upsample_first", "toy_upsample_first", 2, 6, 0.7, 17), ("toy_no_res_skip", "toy_no_res_skip", 2, 5, 0.7, 18),
             ("toy_dilations", "toy_dilations", 2, 5, 0.7, 19),
             ("toy_dilations_h", "toy_dilations_h", 2, 6, 0.7, 20), ("author_toy_dilations_h", "author_toy_dilations_h", 1, 5, 0.7, 21),
             # corners of the reference's published sweep ("WaveFlow Inference Times.png"): n_group 50 / 20 / 12, 128 ... 512 channels
             ("table_g50_c128", "table_g50_c128", 2, 5, 0.7, 22), ("table_g50_c256_sep", "table_g50_c256_sep", 1, 5, 0.7, 23),
             ("table_g20_c512", "table_g20_c512", 1, 5, 0.7, 24), ("table_g12_c256_sep", "table_g12_c256_sep", 2, 4, 0.7, 25)]
    only = [a for a in sys.argv[2:]]
    if full_length:
        # BASELINE config 4 at the metric's utterance length: one 80x900 mel (~10 s of CPU here)
        cases, only = [("full_len", "full", 1, 900, 0.6, 4322)], []
    for name, key, B, F, sigma, seed in cases:
        if only and name not in only:
            continue
        cfg = synthetic.WAVEFLOW_CONFIGS[key]
        sd = synthetic.waveflow_state_dict(cfg, seed=seed)
        model = _ref_waveflow(copy.deepcopy(cfg), sd)
        n_in = cfg["n_mel_channels"] * (2 if cfg.get("use_logvar_channels") else 1)
        mel = synthetic.synthetic_mel(B, F, n_in, seed=seed)
        multispeaker = bool(cfg["speaker_embed"] or cfg["WN_config"]["speaker_embed_dim"])
        ids = np.array([3, 17, 250, 511][:B], np.int64) if multispeaker else None
        tids = None if ids is None else torch.from_numpy(ids)
        # the reference's own infer() (ax:359-388); the noise it draws is replayed from the same seed
        torch.manual_seed(seed)
        z = torch.empty(B, F * cfg["hop_length"]).normal_(std=sigma).numpy()
        torch.manual_seed(seed)
        with torch.no_grad():
            audio = model.infer(torch.from_numpy(mel.copy()), speaker_ids=tids, sigma=sigma).numpy()
            melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
            inv, _ = model.inverse(torch.from_numpy(z.copy()), torch.from_numpy(melp.copy()), speaker_ids=tids)
        inv = inv.numpy()
        assert audio.shape == (B, (F - 1) * cfg["hop_length"]) and np.isfinite(audio).all()
        assert np.array_equal(inv[:, :audio.shape[1]], audio), "noise replay out of sync with infer()"
        path = os.path.join(HERE, f"waveflow_{name}.npz")
        extra = {} if ids is None else {"speaker_ids": ids}
        if full_length:
            np.savez_compressed(path, config_key=key, seed=seed, sigma=np.float32(sigma), B=B, F=F, z=z, audio=audio,
                                mel_recipe="cookietts_amd.synthetic.synthetic_mel(B, F, n_mel, seed=seed)")
        else:
            np.savez_compressed(path, config_key=key, seed=seed, sigma=np.float32(sigma), mel=mel, z=z, audio=audio,
                                inverse_full=inv.astype(np.float32), **extra)
        print(f"[golden] waveflow {name}: audio {audio.shape} rms={audio.std():.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def make_waveglow_ax(full_length=False, untts=False, gates=False):
    """efficient_model_ax.WaveGlow with waveflow=False (AffineCouplingBlock + 1-D WN; InvertibleConv1x1 / PermuteHeight
    mixing in both orders; early outputs; the timed notebook config's option set)."""
    import copy
    torch.set_num_threads(8)
    cases = [("toy_conv", 2, 7, 0.8, 21), ("toy_conv_mixlast", 1, 6, 1.0, 22), ("toy_permute", 2, 5, 0.7, 23),
             ("toy_permute_mixfirst", 1, 9, 0.9, 24), ("notebook_toy", 2, 6, 0.8, 25)]
    if full_length:
        # the notebook config at full width (48 flows x 8 x 256, n_group 24, 160 mel channels), short mel
        cases = [("notebook", 1, 5, 0.9, 26)]
    if gates:
        # the thirteen non-GTU gated units (glow_ax.py:45-165) and merge_res_skip, one tiny model each
        cases = [(k, 1, 4, 0.8, 40 + i) for i, k in enumerate(sorted(
            k for k in synthetic.WAVEGLOW_AX_CONFIGS if k.startswith("toy_gate_") or k == "toy_merge"))]
        # ... and the per-flow (grouped / dense) 1x1 conv of the conditioning
        cases += [("toy_groupconv", 2, 5, 0.8, 60), ("toy_groupconv_dense", 2, 6, 0.8, 61),
                  # the WN's own TransposedUpsampleNet, interpolated and cropped
                  ("toy_wn_tconv", 2, 6, 0.8, 62), ("toy_wn_tconv_crop", 2, 7, 0.8, 63),
                  # sigmoid conditioning activations + preceived_vol_scaling
                  ("toy_sigmoid_vol", 2, 5, 0.5, 64),
                  # res_skip=False
                  ("toy_no_res_skip", 2, 5, 0.8, 65), ("toy_no_res_skip_1layer", 1, 6, 0.8, 66),
                  # per-layer width dilations
                  ("toy_dilations", 2, 5, 0.8, 67), ("toy_dilations_const", 1, 6, 0.8, 68),
                  # n_channels not a multiple of 128
                  ("toy_c96", 2, 5, 0.8, 69), ("toy_c160", 1, 6, 0.8, 70),
                  # n_group = 32 (the widest latent), 1x1-conv mixing before / permutation after the coupling
                  ("toy_g32", 2, 5, 0.8, 71), ("toy_g32_permute", 2, 6, 0.8, 72)]
        only = sys.argv[2:]
        if only:
            cases = [c for c in cases if c[0] in only]
    if untts:
        # the untts notebook's vocoder config (model-level TransposedUpsampleNet, 1x1-conv cond residual, spect shift /
        # scale in the toy), toy and full width (24 flows x 8 x 384, 256 mel channels) on a short mel
        cases = [("untts_toy", 2, 6, 0.8, 27), ("untts", 1, 4, 0.9, 28)]
    for key, B, F, sigma, seed in cases:
        cfg = synthetic.WAVEGLOW_AX_CONFIGS[key]
        sd = synthetic.waveglow_ax_state_dict(cfg, seed=seed)
        model = _ref_waveflow(copy.deepcopy(cfg), sd)
        n_in = cfg["n_mel_channels"] * (2 if cfg.get("use_logvar_channels") else 1)
        mel = synthetic.synthetic_mel(B, F, n_in, seed=seed)
        multispeaker = bool(cfg["speaker_embed"] or cfg["WN_config"]["speaker_embed_dim"])
        ids = np.array([3, 17, 250, 511][:B], np.int64) if multispeaker else None
        tids = None if ids is None else torch.from_numpy(ids)
        samples = F * cfg["hop_length"]
        samples -= samples % cfg["n_group"]
        torch.manual_seed(seed)
        z = torch.empty(B, samples).normal_(std=sigma).numpy()
        torch.manual_seed(seed)
        with torch.no_grad():
            audio = model.infer(torch.from_numpy(mel.copy()), speaker_ids=tids, sigma=sigma).numpy()
            melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
            inv, _ = model.inverse(torch.from_numpy(z.copy()), torch.from_numpy(melp.copy()), speaker_ids=tids)
        inv = inv.numpy()
        assert audio.shape == (B, samples - cfg["hop_length"]) and np.isfinite(audio).all()
        assert np.array_equal(inv[:, :audio.shape[1]], audio), "noise replay out of sync with infer()"
        path = os.path.join(HERE, f"waveglow_ax_{key}.npz")
        extra = {} if ids is None else {"speaker_ids": ids}
        np.savez_compressed(path, config_key=key, seed=seed, sigma=np.float32(sigma), mel=mel, z=z, audio=audio,
                            inverse_full=inv.astype(np.float32), **extra)
        print(f"[golden] waveglow_ax {key}: audio {audio.shape} rms={audio.std():.4f} max={np.abs(audio).max():.2f} -> "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def _ref_tacotron(hp, seed, attention_drive=None, stop_drive=None, shapes_file="tacotron_state_shapes.json"):
    """Reference Tacotron2 with the recipe weights.  Shims: no-op RNNCellBase input checks
    (removed in torch 2.x, called at utils/model/layers.py:375-379)."""
    import json
    import torch.nn.modules.rnn as rnn
    if not hasattr(rnn.RNNCellBase, "check_forward_input"):
        rnn.RNNCellBase.check_forward_input = lambda self, x: None
        rnn.RNNCellBase.check_forward_hidden = lambda self, x, h, s='': None
    from CookieTTS._2_ttm.tacotron2_tm import model as ref_model
    torch.manual_seed(0)
    m = ref_model.Tacotron2(hp)
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(HERE, shapes_file), "w") as f:
        json.dump(shapes, f, indent=0, sort_keys=True)
    sd = synthetic.tacotron_state_dict(hp, seed=seed, shapes=shapes, attention_drive=attention_drive, stop_drive=stop_drive)
    res = m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.eval(), ref_model, sd


class _MaskedDropout:
    """Stand-in for F.dropout inside the reference's model.py that applies OUR keep-masks to the prenet's
    always-on dropout (model.py:189-190) so the golden is reproducible without torch's RNG."""

    def __init__(self, masks):
        self.masks, self.calls = masks, 0

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return x
        assert p == 0.5 and x.shape[-1] == self.masks.shape[-1]
        step, layer = divmod(self.calls, 2)
        self.calls += 1
        keep = torch.from_numpy(self.masks[step, layer].astype(np.float32))
        return x * keep * 2.0


def make_tacotron():
    torch.set_num_threads(8)
    hp = synthetic.tacotron_hparams()
    seed = 1234
    model, ref_model, sd = _ref_tacotron(hp, seed)
    rng = np.random.default_rng(99)
    # ---- decoder-only golden: Decoder.inference on a synthetic memory, fixed number of steps
    B, T_txt, n_steps = 2, 60, 14
    lengths = np.array([60, 41], dtype=np.int64)
    memory_in = (rng.standard_normal((B, T_txt, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)
    masks = synthetic.prenet_dropout_masks(n_steps, B, hp.prenet_dim, seed=seed)
    saved = ref_model.F.dropout
    ref_model.F.dropout = _MaskedDropout(masks)
    try:
        model.decoder.max_decoder_steps = n_steps
        model.decoder.gate_threshold = 2.0            # sigmoid never exceeds it: run all n_steps
        with torch.no_grad():
            mel, gate, align, _ = model.decoder.inference(torch.from_numpy(memory_in), torch.from_numpy(lengths))
    finally:
        ref_model.F.dropout = saved
    path = os.path.join(HERE, "tacotron_decoder.npz")
    np.savez_compressed(path, seed=seed, memory_in=memory_in, lengths=lengths, masks=masks,
                        mel=mel.numpy().astype(np.float32), gate_sigmoid=gate.numpy().astype(np.float32),
                        alignments=align.numpy().astype(np.float32))
    print(f"[golden] tacotron_decoder: mel {tuple(mel.shape)} align {tuple(align.shape)} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")

    # ---- full model golden: Tacotron2.inference (embedding, encoder, memory, decoder, postnet)
    B, T_txt, n_steps = 3, 30, 10
    text = rng.integers(1, hp.n_symbols, size=(B, T_txt)).astype(np.int64)
    lengths = np.array([30, 22, 9], dtype=np.int64)
    for b in range(B):
        text[b, lengths[b]:] = 0                                   # pad symbol id beyond each length
    speakers = np.array([0, 5, 17], dtype=np.int64)
    tm = rng.standard_normal((B, hp.torchMoji_attDim)).astype(np.float32)
    masks = synthetic.prenet_dropout_masks(n_steps, B, hp.prenet_dim, seed=seed + 1)
    ref_model.F.dropout = _MaskedDropout(masks)
    try:
        model.decoder.max_decoder_steps = n_steps
        with torch.no_grad():
            enc_out, _, sylps = model.encoder(model.embedding(torch.from_numpy(text)).transpose(1, 2),
                                              torch.from_numpy(lengths), speaker_ids=torch.from_numpy(speakers))
            out = model.inference(torch.from_numpy(text), torch.from_numpy(lengths), torch.from_numpy(speakers),
                                  torch.from_numpy(tm))
    finally:
        ref_model.F.dropout = saved
    path = os.path.join(HERE, "tacotron_full.npz")
    np.savez_compressed(path, seed=seed, text=text, lengths=lengths, speakers=speakers, torchmoji=tm, masks=masks,
                        encoder_outputs=enc_out.numpy().astype(np.float32), pred_sylps=sylps.numpy().astype(np.float32),
                        pred_mel_postnet=out["pred_mel_postnet"].numpy().astype(np.float32),
                        pred_gate=out["pred_gate"].numpy().astype(np.float32),
                        alignments=out["alignments"].numpy().astype(np.float32))
    print(f"[golden] tacotron_full: postnet mel {tuple(out['pred_mel_postnet'].shape)} enc {tuple(enc_out.shape)} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def make_tacotron_small():
    """Tacotron2.inference of a model whose hparams are NOT the repo defaults (every width roughly halved,
    synthetic.TACOTRON_SMALL_OVERRIDES): what loading somebody's checkpoint amounts to (text2speech.py:299-316)."""
    torch.set_num_threads(8)
    hp = synthetic.tacotron_hparams(**synthetic.TACOTRON_SMALL_OVERRIDES)
    seed = 4321
    model, ref_model, sd = _ref_tacotron(hp, seed, shapes_file="tacotron_small_state_shapes.json")
    rng = np.random.default_rng(123)
    B, T_txt, n_steps = 3, 40, 24
    text = rng.integers(1, hp.n_symbols, size=(B, T_txt)).astype(np.int64)
    lengths = np.array([40, 29, 12], dtype=np.int64)
    for b in range(B):
        text[b, lengths[b]:] = 0
    speakers = np.array([2, 9, 300], dtype=np.int64)
    tm = rng.standard_normal((B, hp.torchMoji_attDim)).astype(np.float32)
    masks = synthetic.prenet_dropout_masks(n_steps, B, hp.prenet_dim, seed=seed + 1)
    saved = ref_model.F.dropout
    ref_model.F.dropout = _MaskedDropout(masks)
    try:
        model.decoder.max_decoder_steps = n_steps
        model.decoder.gate_threshold = 2.0
        with torch.no_grad():
            enc_out, _, sylps = model.encoder(model.embedding(torch.from_numpy(text)).transpose(1, 2),
                                              torch.from_numpy(lengths), speaker_ids=torch.from_numpy(speakers))
            out = model.inference(torch.from_numpy(text), torch.from_numpy(lengths), torch.from_numpy(speakers),
                                  torch.from_numpy(tm))
    finally:
        ref_model.F.dropout = saved
    path = os.path.join(HERE, "tacotron_small.npz")
    np.savez_compressed(path, seed=seed, text=text, lengths=lengths, speakers=speakers, torchmoji=tm, masks=masks,
                        encoder_outputs=enc_out.numpy().astype(np.float32), pred_sylps=sylps.numpy().astype(np.float32),
                        pred_mel_postnet=out["pred_mel_postnet"].numpy().astype(np.float32),
                        pred_gate=out["pred_gate"].numpy().astype(np.float32),
                        alignments=out["alignments"].numpy().astype(np.float32))
    print(f"[golden] tacotron_small: postnet mel {tuple(out['pred_mel_postnet'].shape)} enc {tuple(enc_out.shape)} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


TACOTRON_LONG = {"long": None, "long_peaked": (2.0, 8.0, 1.0), "long_sharp": (4.0, 12.0, 10.0)}   # name -> synthetic attention_drive


def make_tacotron_long():
    """Config 5 at full size over a long horizon (BASELINE.json config 5; model.py:1044-1080, :851-916, :131-146):
    Tacotron2.inference on B=4, 200 symbols, lengths [200,195,150,100], 256 forced steps, prenet masks injected.
    Three weight recipes: the plain one (near-uniform attention; the window drifts to the right clamp of every item),
    a peaked one (weights 0.5-0.7 advancing ~1.5 tokens per step, diffuse once the window sits at the right clamp) and a
    sharp one (weights up to 0.99, jittering at the clamp: a trajectory on which the reference's own fp32 rounding is
    amplified to 3e-2 in the weights by step 256 - kept to pin the growth law, tests/test_tacotron_long.py).
    Captured besides the outputs: the decoder's mel before the postnet, the attention position the decoder carries
    into each step (the ``current_pos`` argument of Attention.forward) and the window start the reference derives
    from it (same three lines as model.py:131-139, applied to the captured position)."""
    torch.set_num_threads(8)
    hp = synthetic.tacotron_hparams()
    seed, n_steps, B, T_txt = 1234, 256, 4, 200
    lengths = np.array([200, 195, 150, 100], dtype=np.int64)
    rng = np.random.default_rng(2026)
    text = rng.integers(1, hp.n_symbols, size=(B, T_txt)).astype(np.int64)
    for b in range(B):
        text[b, lengths[b]:] = 0
    speakers = np.array([0, 1, 2, 3], dtype=np.int64)
    tm = rng.standard_normal((B, hp.torchMoji_attDim)).astype(np.float32)
    mask_seed = seed + 2
    masks = synthetic.prenet_dropout_masks(n_steps, B, hp.prenet_dim, seed=mask_seed)
    for name, drive in TACOTRON_LONG.items():
        model, ref_model, sd = _ref_tacotron(hp, seed, attention_drive=drive)
        pos, dec_mel = [], []
        h1 = model.decoder.attention_layer.register_forward_pre_hook(lambda m, a: pos.append(a[7].detach().clone()))
        h2 = model.postnet.register_forward_pre_hook(lambda m, a: dec_mel.append(a[0].detach().clone()))
        saved = ref_model.F.dropout
        ref_model.F.dropout = _MaskedDropout(masks)
        try:
            model.decoder.max_decoder_steps = n_steps
            model.decoder.gate_threshold = 2.0
            with torch.no_grad():
                enc_out, _, sylps = model.encoder(model.embedding(torch.from_numpy(text)).transpose(1, 2),
                                                  torch.from_numpy(lengths), speaker_ids=torch.from_numpy(speakers))
                out = model.inference(torch.from_numpy(text), torch.from_numpy(lengths), torch.from_numpy(speakers),
                                      torch.from_numpy(tm))
        finally:
            ref_model.F.dropout = saved
            h1.remove()
            h2.remove()
        pos = torch.stack(pos)                                         # [steps, B]: position carried INTO each step
        att = model.decoder.attention_layer
        R = att.windowed_attention_range
        cur = pos + att.windowed_att_pos_offset.detach()
        cur = torch.min(cur.clamp(min=R), (torch.from_numpy(lengths) - 1 - R).to(cur))
        start = (cur - R).clamp(min=0).round().to(torch.int64).numpy()  # [steps, B]
        align = out["alignments"].numpy().astype(np.float32)
        for b in range(B):                                             # the captured start IS the support of the weights
            for i in range(n_steps):
                nz = np.nonzero(align[b, i])[0]
                assert nz.min() >= start[i, b] and nz.max() <= min(start[i, b] + 2 * R, lengths[b] - 1)
        extra = {"encoder_outputs": enc_out.numpy().astype(np.float32)} if drive is None else {}
        path = os.path.join(HERE, f"tacotron_{name}.npz")
        np.savez_compressed(path, seed=seed, mask_seed=mask_seed, n_steps=n_steps,
                            attention_drive=np.array(drive if drive else [], dtype=np.float32),
                            text=text, lengths=lengths, speakers=speakers, torchmoji=tm,
                            pred_sylps=sylps.numpy().astype(np.float32),
                            decoder_mel=dec_mel[0].numpy().astype(np.float32),
                            pred_mel_postnet=out["pred_mel_postnet"].numpy().astype(np.float32),
                            pred_gate=out["pred_gate"].numpy().astype(np.float32), alignments=align,
                            attention_position=pos.numpy().astype(np.float32), window_start=start.astype(np.int32), **extra)
        print(f"[golden] tacotron_{name}: postnet mel {tuple(out['pred_mel_postnet'].shape)}, window start at steps "
              f"0/64/128/255 {start[0].tolist()} {start[64].tolist()} {start[128].tolist()} {start[255].tolist()}, "
              f"max weight {align.max(-1).mean():.3f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def make_tacotron_batch8():
    """Tacotron2.inference on EIGHT ragged utterances (more than the four one persistent-decoder workspace holds: the HIP side
    decodes them as one batched MFMA call, csrc/tacotron_batched.h), 48 forced steps, the peaked attention recipe so that the
    windows advance; texts of 12 and 20 symbols are shorter than the 33-token window (model.py:131-146 clamps)."""
    torch.set_num_threads(8)
    hp = synthetic.tacotron_hparams()
    seed, n_steps, B, T_txt = 1234, 48, 8, 64
    drive = TACOTRON_LONG["long_peaked"]
    model, ref_model, sd = _ref_tacotron(hp, seed, attention_drive=drive)
    lengths = np.array([64, 61, 50, 47, 33, 20, 12, 40], dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")                          # pack_padded_sequence wants them sorted (model.py:299)
    lengths = lengths[order]
    rng = np.random.default_rng(808)
    text = rng.integers(1, hp.n_symbols, size=(B, T_txt)).astype(np.int64)
    for b in range(B):
        text[b, lengths[b]:] = 0
    speakers = np.array([3, 1, 4, 1, 5, 9, 2, 6], dtype=np.int64)
    tm = rng.standard_normal((B, hp.torchMoji_attDim)).astype(np.float32)
    mask_seed = seed + 8
    masks = synthetic.prenet_dropout_masks(n_steps, B, hp.prenet_dim, seed=mask_seed)
    dec_mel = []
    h2 = model.postnet.register_forward_pre_hook(lambda m, a: dec_mel.append(a[0].detach().clone()))
    saved = ref_model.F.dropout
    ref_model.F.dropout = _MaskedDropout(masks)
    try:
        model.decoder.max_decoder_steps = n_steps
        model.decoder.gate_threshold = 2.0
        with torch.no_grad():
            out = model.inference(torch.from_numpy(text), torch.from_numpy(lengths), torch.from_numpy(speakers),
                                  torch.from_numpy(tm))
    finally:
        ref_model.F.dropout = saved
        h2.remove()
    align = out["alignments"].numpy().astype(np.float32)
    path = os.path.join(HERE, "tacotron_batch8.npz")
    np.savez_compressed(path, seed=seed, mask_seed=mask_seed, n_steps=n_steps, attention_drive=np.array(drive, dtype=np.float32),
                        text=text, lengths=lengths, speakers=speakers, torchmoji=tm,
                        decoder_mel=dec_mel[0].numpy().astype(np.float32),
                        pred_mel_postnet=out["pred_mel_postnet"].numpy().astype(np.float32),
                        pred_gate=out["pred_gate"].numpy().astype(np.float32), alignments=align)
    print(f"[golden] tacotron_batch8: postnet mel {tuple(out['pred_mel_postnet'].shape)}, mean max weight "
          f"{align.max(-1).mean():.3f}, last-step arg-max per item {align[:, -1].argmax(-1).tolist()} -> "
          f"{os.path.getsize(path) / 1024:.0f} KiB")


# name -> (gate_threshold, gate_delay, max_decoder_steps) set on the decoder the way the server does (text2speech.py:410-412,457)
TACOTRON_STOP_CASES = {
    "delay0": (0.5, 0, 200), "delay2": (0.5, 2, 200), "delay3": (0.5, 3, 200), "delay4": (0.5, 4, 200),
    "delay10": (0.5, 10, 200), "thr_hi": (0.75, 2, 90), "cap_before": (0.5, 0, 50), "cap_in_delay": (0.5, 4, 64),
    "never": (0.97, 0, 80),
}
# the step clock built into the second decoder LSTM (synthetic.tacotron_state_dict stop_drive): (rate, sharpness, step times)
TACOTRON_STOP_DRIVE = (0.01, 800.0, [1, 4, 23, 25, 40, 52, 61])
# what the gate layer adds per step unit: f(t) = -1 + sum_j amp_j [t >= t_j]  ->  f = +1 on steps 1-3 and 23-24, -1 between,
# 3 from step 40, 5 from 52, 7 from 61; item b's logit is f(t) - level_b
TACOTRON_STOP_AMPS = [2.0, -2.0, 2.0, -2.0, 4.0, 2.0, 2.0]
TACOTRON_STOP_LEVELS = [0.0, 2.0, 6.0, 4.0]
# => first step >= 5 at which sigmoid(gate) > 0.5 per item.  Item 0 also spikes on steps 1-3 (ignored by the rule, model.py:893:
# ``if i > 4``) and is over the threshold on steps 23, 24 only until step 40 (the rule keeps the running max, :894)
TACOTRON_STOP_CROSS = [23, 40, 61, 52]
TACOTRON_STOP_PROBE = 72


def make_tacotron_stop():
    """Goldens in which the reference's OWN stop rule ends Decoder.inference (model.py:879-904): full-size model (config 5
    hparams), B=4, 72 symbols, lengths [72,64,48,56], peaked attention recipe, and the step clock of
    ``synthetic.tacotron_state_dict(stop_drive=...)`` in the second decoder LSTM.  The gate layer (weight and bias stored in
    the fixture: part of the weight recipe) reads the clock's step units with the amplitudes above, and tells the items apart
    through the attention context, whose speaker part is constant per item: a forced probe run of the reference captures the
    context per step and the context half of the gate weight is the ridge solution that maps it to -level_b (the gate output
    does not feed back into the recurrence, model.py:668-767, so the probe's trajectory is the cases' trajectory).
    Each case then runs ``Tacotron2.inference`` with the reference's rule live and ``gate_threshold`` / ``gate_delay`` /
    ``max_decoder_steps`` set on the decoder as the server sets them."""
    torch.set_num_threads(8)
    hp = synthetic.tacotron_hparams()
    seed, B, T_txt = 1234, 4, 72
    lengths = np.array([72, 64, 48, 56], dtype=np.int64)
    rng = np.random.default_rng(4242)
    text = rng.integers(1, hp.n_symbols, size=(B, T_txt)).astype(np.int64)
    for b in range(B):
        text[b, lengths[b]:] = 0
    speakers = np.array([3, 2, 1, 0], dtype=np.int64)
    tm = rng.standard_normal((B, hp.torchMoji_attDim)).astype(np.float32)
    mask_seed = seed + 3
    max_cap = max(c[2] for c in TACOTRON_STOP_CASES.values())
    masks = synthetic.prenet_dropout_masks(max_cap, B, hp.prenet_dim, seed=mask_seed)
    drive = TACOTRON_LONG["long_peaked"]
    model, ref_model, sd = _ref_tacotron(hp, seed, attention_drive=drive, stop_drive=TACOTRON_STOP_DRIVE)
    args = (torch.from_numpy(text), torch.from_numpy(lengths), torch.from_numpy(speakers), torch.from_numpy(tm))
    saved = ref_model.F.dropout

    def run(thr, delay, cap, hook=None):
        ref_model.F.dropout = _MaskedDropout(masks)
        h = model.decoder.gate_layer.register_forward_pre_hook(hook) if hook else None
        dec_mel = []
        h2 = model.postnet.register_forward_pre_hook(lambda m, a: dec_mel.append(a[0].detach().clone()))
        try:
            model.decoder.gate_delay = int(delay)                 # text2speech.py:410
            model.decoder.max_decoder_steps = int(cap)            # :411
            model.decoder.gate_threshold = float(thr)             # :412
            with torch.no_grad():
                out = model.inference(*args)
        finally:
            ref_model.F.dropout = saved
            h2.remove()
            if h:
                h.remove()
        return {k: v.numpy().astype(np.float32) for k, v in out.items()}, dec_mel[0].numpy().astype(np.float32)

    # ---- probe: forced run, capture the gate layer's input [d ; ctx] per step
    feats = []
    run(2.0, 0, TACOTRON_STOP_PROBE, hook=lambda m, a: feats.append(a[0].detach().clone().numpy().astype(np.float64)))
    X = np.stack(feats)                                            # [steps, B, 1280]
    n, H = X.shape[0], hp.second_decoder_rnn_dim
    k = len(TACOTRON_STOP_AMPS)
    hi = np.tanh(1.0)                                              # a step unit's two output levels are -+tanh(1) (sharp limit)
    units = X[:, :, 1:1 + k]
    on = np.arange(n)[:, None, None] >= np.array(TACOTRON_STOP_DRIVE[2])[None, None, :]
    assert ((units > 0) == on).all() and np.abs(np.abs(units) - hi).max() < 0.02, np.abs(np.abs(units) - hi).max()
    ctx = X[:, :, H:].reshape(n * B, -1)
    A = np.concatenate([ctx, np.ones((n * B, 1))], axis=1)
    target = np.tile(-np.array(TACOTRON_STOP_LEVELS), n)
    sol = np.linalg.solve(A @ A.T + 1e-2 * np.eye(n * B), target)
    wb = A.T @ sol
    gate_w = np.zeros((1, X.shape[2]), dtype=np.float32)
    gate_w[0, H:] = wb[:-1]
    gate_w[0, 1:1 + k] = np.array(TACOTRON_STOP_AMPS) / (2 * hi)
    gate_b = np.array([wb[-1] - 1.0 + sum(TACOTRON_STOP_AMPS) / 2], dtype=np.float32)
    fit = (A @ wb).reshape(n, B)
    print(f"[golden] tacotron_stop: context half of the gate weight |w|_2 = {np.linalg.norm(wb[:-1]):.2f}, item-level fit "
          f"residual {np.abs(fit + np.array(TACOTRON_STOP_LEVELS)).max():.4f}")
    assert np.abs(fit + np.array(TACOTRON_STOP_LEVELS)).max() < 0.05
    with torch.no_grad():
        model.decoder.gate_layer.linear_layer.weight.copy_(torch.from_numpy(gate_w))
        model.decoder.gate_layer.linear_layer.bias.copy_(torch.from_numpy(gate_b))

    # ---- the cases: the reference's rule ends the loop
    store = dict(seed=seed, mask_seed=mask_seed, attention_drive=np.array(drive, dtype=np.float32),
                 stop_rate=np.float64(TACOTRON_STOP_DRIVE[0]), stop_sharp=np.float64(TACOTRON_STOP_DRIVE[1]),
                 stop_times=np.array(TACOTRON_STOP_DRIVE[2], dtype=np.int64), text=text, lengths=lengths,
                 speakers=speakers, torchmoji=tm, gate_w=gate_w, gate_b=gate_b, case_names=np.array(list(TACOTRON_STOP_CASES)),
                 case_params=np.array(list(TACOTRON_STOP_CASES.values()), dtype=np.float64))
    t_mel, outs = [], {}
    for name, (thr, delay, cap) in TACOTRON_STOP_CASES.items():
        out, dec_mel = run(thr, delay, cap)
        T = out["pred_mel_postnet"].shape[2]
        assert out["pred_gate"].shape == (B, T) and out["alignments"].shape == (B, T, T_txt) and dec_mel.shape == (B, 80, T)
        t_mel.append(T)
        outs[name] = (out, dec_mel)
        store[f"{name}_pred_mel_postnet"] = out["pred_mel_postnet"]  # the postnet sees T frames: differs near the end per case
        print(f"[golden] tacotron_stop {name}: thr {thr} delay {delay} cap {cap} -> T_mel {T}")
    longest = max(outs, key=lambda c: outs[c][1].shape[2])
    out, dec_mel = outs[longest]
    for name, T in zip(TACOTRON_STOP_CASES, t_mel):                  # the rule does not feed back: every case is a prefix
        o, dm = outs[name]
        assert np.array_equal(dm, dec_mel[:, :, :T]) and np.array_equal(o["alignments"], out["alignments"][:, :T])
        assert np.abs(o["pred_gate"] - out["pred_gate"][:, :T]).max() < 1e-7       # torch.sigmoid over a different length: 1 ulp
    store.update(T_mel=np.array(t_mel, dtype=np.int64), decoder_mel=dec_mel, pred_gate=out["pred_gate"],
                 alignments=out["alignments"])
    sg = out["pred_gate"]
    first = [int(np.argmax(sg[b, 5:] > 0.5) + 5) for b in range(B)]
    assert first == TACOTRON_STOP_CROSS, first
    assert (sg[0, 1:4] > 0.5).all() and (sg[1:, :5] < 0.5).all() and (sg[0, 25:40] < 0.5).all() and (sg[0, 4:23] < 0.5).all()
    assert t_mel == [62, 64, 65, 66, 72, 90, 50, 64, 80], t_mel
    print(f"[golden] tacotron_stop: first crossings {first}, min |sigmoid(gate) - 0.5| = {np.abs(sg - 0.5).min():.3f}, "
          f"max sigmoid {sg.max():.3f}")
    path = os.path.join(HERE, "tacotron_stop.npz")
    np.savez_compressed(path, **store)
    print(f"[golden] tacotron_stop -> {os.path.getsize(path) / 1024:.0f} KiB")


def make_alignment():
    """utils/model/utils.py:47-120 run here on seeded attention maps (data only)."""
    from CookieTTS.utils.model.utils import alignment_metric, get_first_over_thresh
    rng = np.random.default_rng(77)
    out = {}
    for tag, (B, dec, enc) in {"small": (3, 57, 23), "wide": (2, 130, 300)}.items():
        # peaked, roughly monotonic maps with noise: what a decoder run produces
        pos = np.linspace(0, enc - 1, dec)[None, :, None] * rng.uniform(0.6, 1.1, (B, 1, 1))
        e = np.arange(enc)[None, None, :]
        logits = -0.5 * ((e - pos) / rng.uniform(0.5, 2.0, (B, 1, 1))) ** 2 + rng.normal(0, 0.3, (B, dec, enc))
        al = np.exp(logits - logits.max(-1, keepdims=True))
        al = (al / al.sum(-1, keepdims=True)).astype(np.float32)
        in_len = rng.integers(enc // 2, enc + 1, B).astype(np.int32)
        out_len = rng.integers(dec // 2, dec + 1, B).astype(np.int32)
        in_len[0], out_len[0] = enc, dec
        out[f"{tag}_alignments"], out[f"{tag}_in_len"], out[f"{tag}_out_len"] = al, in_len, out_len
        for lens, key in ((True, "lens"), (False, "nolens")):
            r = alignment_metric(torch.from_numpy(al.copy()),
                                 input_lengths=torch.from_numpy(in_len) if lens else None,
                                 output_lengths=torch.from_numpy(out_len) if lens else None)
            for k, v in r.items():
                out[f"{tag}_{key}_{k}"] = v.numpy()
    gate = rng.uniform(0, 0.45, (5, 41)).astype(np.float32)
    gate[0, 17] = 0.9; gate[0, 30] = 0.95          # first crossing wins
    gate[1, 40] = 0.1                               # never crosses -> T-1
    gate[2, 0] = 0.7                                # immediately
    gate[3, 9] = 0.5                                # exactly the threshold counts (arg-max of the clamped row)
    out["gate"], out["gate_threshold"] = gate, np.float32(0.5)
    gate[4] = np.minimum(gate[4], 0.49)
    # shim: utils.py:53 parses torch.__version__ with int(); the "+rocm" local tag of this build breaks that parse
    real_version, torch.__version__ = torch.__version__, torch.__version__.split("+")[0]
    try:
        out["gate_first"] = get_first_over_thresh(torch.from_numpy(gate.copy()), 0.5).numpy()
    finally:
        torch.__version__ = real_version
    path = os.path.join(HERE, "alignment.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if "alignments" in k or k.startswith("gate")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["waveglow", "stft", "waveflow", "waveglow_ax", "tacotron", "alignment"]
    if "alignment" in which:
        make_alignment()
    if "tacotron" in which:
        make_tacotron()
    if "tacotron_small" in which:      # on request only
        make_tacotron_small()
    if "tacotron_long" in which:
        make_tacotron_long()
    if "tacotron_stop" in which:
        make_tacotron_stop()
    if "tacotron_batch8" in which:
        make_tacotron_batch8()
    if "waveflow" in which:
        make_waveflow()
    if "waveglow" in which:
        make_waveglow()
    if "stft" in which:
        make_stft()
    if "waveglow_options" in which or not sys.argv[1:]:
        make_waveglow(options=True)
    if "waveglow_ax" in which:
        make_waveglow_ax()
    if "waveglow_ax_notebook" in which:    # on request only: 272 M parameters
        make_waveglow_ax(full_length=True)
    if "waveglow_ax_untts" in which:       # on request only
        make_waveglow_ax(untts=True)
    if "waveglow_ax_gates" in which:       # on request only
        make_waveglow_ax(gates=True)
    if "waveglow_full_len" in which:       # on request only: minutes of CPU
        make_waveglow(full_length=True)
    if "waveflow_full_len" in which:
        make_waveflow(full_length=True)
