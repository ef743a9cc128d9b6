"""Deterministic synthetic weights / inputs for the mel-to-wave hot path.

There is no network for checkpoints, so benchmarks, parity tests and the golden
vector generator all draw random-init weights from this one recipe.  The recipe
is pure numpy (PCG64 via ``np.random.default_rng``) so it reproduces bit-for-bit
on the GPU box without torch RNG state or the reference being present.

The state dicts use the *reference's checkpoint key names and shapes*
(``/root/reference/CookieTTS/_4_mtw/waveglow/glow.py:110-186,226-265``;
weight-norm parameters stored as ``weight_g`` / ``weight_v``, SURVEY.md §5
"Checkpoint / resume"), so the same dict loads into the reference model (golden
generation) and into ``cookietts_amd.waveglow.WaveGlow`` (product).

The reference zero-initialises ``WN.end`` (glow.py:141-144), which makes every
coupling the identity; the recipe randomises it so parity is not vacuous.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "WAVEGLOW_CONFIGS",
    "WAVEFLOW_CONFIGS",
    "waveflow_config",
    "waveflow_state_dict",
    "waveglow_config",
    "waveglow_flow_channels",
    "waveglow_state_dict",
    "synthetic_mel",
    "synthetic_noise",
    "to_torch",
]


def _wn_config(n_channels, n_layers=8):
    return dict(n_layers=n_layers, n_channels=n_channels, kernel_size=3,
                speaker_embed_dim=0, rezero=False)


def waveglow_config(n_flows, n_channels, n_group=8, n_layers=8, n_early_every=4,
                    n_early_size=2, n_mel_channels=80, win_length=1024, hop_length=256):
    """Constructor kwargs of the reference ``glow.WaveGlow`` (glow.py:226-227)."""
    return dict(yoyo=False, yoyo_WN=False, n_mel_channels=n_mel_channels, n_flows=n_flows,
                n_group=n_group, n_early_every=n_early_every, n_early_size=n_early_size,
                memory_efficient=False, spect_scaling=False, upsample_mode='normal',
                WN_config=_wn_config(n_channels, n_layers),
                win_length=win_length, hop_length=hop_length)


# BASELINE.json configs 1-3 plus toy sizes for fast CPU tests.
WAVEGLOW_CONFIGS = {
    "toy": waveglow_config(n_flows=2, n_channels=128, n_layers=3, n_early_every=4),
    "toy_early": waveglow_config(n_flows=5, n_channels=128, n_layers=2, n_early_every=2),
    "small": waveglow_config(n_flows=4, n_channels=256),     # config 1
    "full": waveglow_config(n_flows=12, n_channels=512),     # configs 2/3
}


def waveglow_flow_channels(cfg):
    """Per-flow (n_remaining_channels, n_half) exactly as glow.py:251-265 derives them."""
    n_half = cfg["n_group"] // 2
    n_rem = cfg["n_group"]
    out = []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def waveflow_config(n_flows=8, n_group=16, n_channels=64, n_layers=8, kernel_size_w=3, kernel_size_h=3,
                    n_mel_channels=80, hop_length=256):
    """Constructor kwargs of the reference ``efficient_model_ax.WaveGlow`` for BASELINE config 4
    (SURVEY.md 8d row 4)."""
    return dict(n_mel_channels=n_mel_channels, n_flows=n_flows, n_group=n_group, n_early_every=100,
                n_early_size=2, memory_efficient=0.0, spect_scaling=False, upsample_mode='normal',
                upsample_first=False, speaker_embed=0, cond_layers=0, cond_hidden_channels=256,
                cond_output_channels=256, cond_kernel_size=1, cond_residual=False, cond_padding_mode='zeros',
                waveflow=True, channel_mixing='permuteheight', mix_first=False, win_length=1024,
                hop_length=hop_length, sampling_rate=22050,
                WN_config=dict(n_layers=n_layers, n_channels=n_channels, kernel_size_w=kernel_size_w,
                               kernel_size_h=kernel_size_h, n_layers_dilations_w=None,
                               n_layers_dilations_h=[1] * n_layers, speaker_embed_dim=0, rezero=False,
                               cond_layers=1, cond_activation_func='none', negative_slope=None,
                               cond_hidden_channels=256, cond_padding_mode='zeros', seperable_conv=False,
                               res_skip=True, merge_res_skip=False, upsample_mode='linear', cond_kernel_size=1))


WAVEFLOW_CONFIGS = {
    "toy": waveflow_config(n_flows=4, n_group=8, n_channels=64, n_layers=3),
    "full": waveflow_config(),                                   # config 4: 8 flows, 64 ch, h = 16
}


def waveflow_state_dict(cfg, seed=1234, end_std=None):
    """Random-init state dict with the reference's keys (SURVEY.md 8a "Checkpoint keys"):
    ``WN.k.WN.{start,in_layers.i,res_skip_layers.i,cond_layers.0}.{bias,weight_g,weight_v}``,
    ``WN.k.WN.end.{weight,bias}`` (PermuteHeight has no parameters)."""
    rng = np.random.default_rng(seed)
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    kh, kw = wn["kernel_size_h"], wn["kernel_size_w"]
    n_mel = cfg["n_mel_channels"]
    if end_std is None:
        end_std = 0.25 / np.sqrt(C)
    sd = {}

    def wn_conv(prefix, shape, fan):
        bound = 1.0 / np.sqrt(fan)
        v = _uniform(rng, shape, bound)
        norm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(axis=1))
        jitter = 0.9 + 0.2 * rng.random((shape[0],), dtype=np.float32)
        sd[prefix + ".weight_v"] = v
        sd[prefix + ".weight_g"] = (norm * jitter).astype(np.float32).reshape((shape[0],) + (1,) * (len(shape) - 1))
        sd[prefix + ".bias"] = _uniform(rng, (shape[0],), bound)

    for k in range(cfg["n_flows"]):
        p = f"WN.{k}.WN"
        wn_conv(p + ".start", (C, 1, 1, 1), 1)
        sd[p + ".end.weight"] = rng.standard_normal((2, C, 1, 1), dtype=np.float32) * np.float32(end_std)
        sd[p + ".end.bias"] = rng.standard_normal((2,), dtype=np.float32) * np.float32(0.02)
        wn_conv(p + ".cond_layers.0", (2 * C * n_layers, n_mel, 1), n_mel * 4)
        for i in range(n_layers):
            wn_conv(f"{p}.in_layers.{i}", (2 * C, C, kh, kw), C * kh * kw)
            rs = 2 * C if i < n_layers - 1 else C
            wn_conv(f"{p}.res_skip_layers.{i}", (rs, C, 1, 1), C)
    return sd


def _uniform(rng, shape, bound):
    return ((rng.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)


def _wn_conv(rng, sd, prefix, out_ch, in_ch, k, gain=1.0):
    """A weight-normed Conv1d: v ~ U(+-1/sqrt(fan_in)), g = |v| * U(0.9,1.1) * gain."""
    bound = 1.0 / np.sqrt(in_ch * k)
    v = _uniform(rng, (out_ch, in_ch, k), bound)
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
    jitter = 0.9 + 0.2 * rng.random((out_ch, 1, 1), dtype=np.float32)
    sd[prefix + ".weight_v"] = v
    sd[prefix + ".weight_g"] = (norm * jitter * gain).astype(np.float32)
    sd[prefix + ".bias"] = _uniform(rng, (out_ch,), bound)


def waveglow_state_dict(cfg, seed=1234, end_std=None, cond_hidden=256):
    """Random-init reference-format state dict (numpy float32 arrays, insertion-ordered)."""
    rng = np.random.default_rng(seed)
    wn = cfg["WN_config"]
    C, n_layers, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    n_mel, G = cfg["n_mel_channels"], cfg["n_group"]
    win, hop = cfg["win_length"], cfg["hop_length"]
    if end_std is None:
        end_std = 0.25 / np.sqrt(C)
    sd = {}
    taps = win // hop
    sd["upsample.weight"] = _uniform(rng, (n_mel, n_mel, win), 1.0 / np.sqrt(n_mel * taps))
    sd["upsample.bias"] = _uniform(rng, (n_mel,), 0.1)
    for k, (n_rem, n_half) in enumerate(waveglow_flow_channels(cfg)):
        p = f"WN.{k}"
        _wn_conv(rng, sd, p + ".start", C, n_half, 1)
        sd[p + ".end.weight"] = (rng.standard_normal((2 * n_half, C, 1), dtype=np.float32)
                                 * np.float32(end_std))
        sd[p + ".end.bias"] = (rng.standard_normal((2 * n_half,), dtype=np.float32)
                               * np.float32(0.02))
        _wn_conv(rng, sd, p + ".cond_layers.0", cond_hidden, n_mel * G, 1)
        _wn_conv(rng, sd, p + ".cond_layers.1", cond_hidden, cond_hidden, 1)
        _wn_conv(rng, sd, p + ".cond_layers.2", 2 * C * n_layers, cond_hidden, 1)
        for i in range(n_layers):
            _wn_conv(rng, sd, f"{p}.in_layers.{i}", 2 * C, C, ks)
            rs = 2 * C if i < n_layers - 1 else C
            _wn_conv(rng, sd, f"{p}.res_skip_layers.{i}", rs, C, 1)
        # invertible 1x1: random orthonormal (QR of a Gaussian) perturbed so W^-1 != W^T
        a = rng.standard_normal((n_rem, n_rem)).astype(np.float64)
        q, _ = np.linalg.qr(a)
        q = q + 0.05 * rng.standard_normal((n_rem, n_rem))
        sd[f"convinv.{k}.conv.weight"] = q.astype(np.float32)[:, :, None]
    return sd


def synthetic_mel(batch, frames, n_mel=80, seed=1234):
    """Log-mel-like input: N(-5, 2^2) clipped to [-11.52, 2]."""
    rng = np.random.default_rng(seed + 7919)
    mel = rng.standard_normal((batch, n_mel, frames), dtype=np.float32) * np.float32(2.0) - np.float32(5.0)
    return np.clip(mel, -11.52, 2.0).astype(np.float32)


def synthetic_noise(batch, n_group, steps, seed=1234):
    """Unit-variance z for every channel the flow stack will ever consume.

    Layout ``[B, n_group, L]``: the last ``n_remaining_channels`` rows are the
    initial latent (glow.py:326); rows above them are the early-output noise
    re-injected at flows k = n_early_every, 2*n_early_every, ... (glow.py:342-347),
    in the order they are prepended.  ``sigma`` is applied by the consumer.
    """
    rng = np.random.default_rng(seed + 104729)
    return rng.standard_normal((batch, n_group, steps), dtype=np.float32)


def to_torch(sd, device=None):
    """numpy state dict -> torch tensors (for ``load_state_dict``)."""
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) if device else torch.from_numpy(np.ascontiguousarray(v))
            for k, v in sd.items()}
