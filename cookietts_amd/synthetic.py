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
    "tacotron_hparams",
    "tacotron_memory_in_dim",
    "tacotron_state_dict",
    "prenet_dropout_masks",
]


def _wn_config(n_channels, n_layers=8, speaker_embed_dim=0, rezero=False):
    return dict(n_layers=n_layers, n_channels=n_channels, kernel_size=3,
                speaker_embed_dim=speaker_embed_dim, rezero=rezero)


def waveglow_config(n_flows, n_channels, n_group=8, n_layers=8, n_early_every=4,
                    n_early_size=2, n_mel_channels=80, win_length=1024, hop_length=256, upsample_mode='normal',
                    speaker_embed_dim=0, rezero=False):
    """Constructor kwargs of the reference ``glow.WaveGlow`` (glow.py:226-227)."""
    return dict(yoyo=False, yoyo_WN=False, n_mel_channels=n_mel_channels, n_flows=n_flows,
                n_group=n_group, n_early_every=n_early_every, n_early_size=n_early_size,
                memory_efficient=False, spect_scaling=False, upsample_mode=upsample_mode,
                WN_config=_wn_config(n_channels, n_layers, speaker_embed_dim, rezero),
                win_length=win_length, hop_length=hop_length)


# BASELINE.json configs 1-3 plus toy sizes for fast CPU tests.
WAVEGLOW_CONFIGS = {
    "toy": waveglow_config(n_flows=2, n_channels=128, n_layers=3, n_early_every=4),
    "toy_early": waveglow_config(n_flows=5, n_channels=128, n_layers=2, n_early_every=2),
    "small": waveglow_config(n_flows=4, n_channels=256),     # config 1
    "full": waveglow_config(n_flows=12, n_channels=512),     # configs 2/3
    # glow.py options: multispeaker (WN speaker embeddings, glow.py:131-133) + ReZero (:127-128), grouped upsampling (:241)
    "toy_spk_rezero": waveglow_config(n_flows=4, n_channels=128, n_layers=3, n_early_every=2, speaker_embed_dim=20,
                                      rezero=True),
    "toy_simple": waveglow_config(n_flows=2, n_channels=128, n_layers=2, n_early_every=4, upsample_mode='simple'),
    "toy_simple_half": waveglow_config(n_flows=2, n_channels=128, n_layers=2, n_early_every=4,
                                       upsample_mode='simple_half'),
    # glow.py:226-265 takes any hop_length / n_group: the widest latent the HIP path builds (16 rows, early outputs of 4
    # rows -> flows of 16 and 12 channels) at hop 512 with two upsampling taps, and n_group 12 at hop 384 (hop / n_group must
    # be a multiple of 4 on the HIP path: latent rows of whole float4s)
    "toy_hop512_g16": waveglow_config(n_flows=4, n_channels=128, n_layers=2, n_group=16, n_early_every=2, n_early_size=4,
                                      win_length=1024, hop_length=512),
    "toy_hop384_g12": waveglow_config(n_flows=3, n_channels=128, n_layers=2, n_group=12, n_early_every=2, n_early_size=4,
                                      win_length=1152, hop_length=384),
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
                    n_mel_channels=80, hop_length=256, win_length=1024, sampling_rate=22050, **over):
    """Constructor kwargs of the reference ``efficient_model_ax.WaveGlow`` for BASELINE config 4
    (SURVEY.md 8d row 4).  ``over``: model-level kwargs to replace; ``WN`` = dict of WN_config entries to replace."""
    wn_over = over.pop("WN", {})
    cfg = dict(n_mel_channels=n_mel_channels, n_flows=n_flows, n_group=n_group, n_early_every=100,
               n_early_size=2, memory_efficient=0.0, spect_scaling=False, upsample_mode='normal',
               upsample_first=False, speaker_embed=0, cond_layers=0, cond_hidden_channels=256,
               cond_output_channels=256, cond_kernel_size=1, cond_residual=False, cond_padding_mode='zeros',
               waveflow=True, channel_mixing='permuteheight', mix_first=False, win_length=win_length,
               hop_length=hop_length, sampling_rate=sampling_rate,
               WN_config=dict(n_layers=n_layers, n_channels=n_channels, kernel_size_w=kernel_size_w,
                              kernel_size_h=kernel_size_h, n_layers_dilations_w=None,
                              n_layers_dilations_h=[1] * n_layers, speaker_embed_dim=0, rezero=False,
                              cond_layers=1, cond_activation_func='none', negative_slope=None,
                              cond_hidden_channels=256, cond_padding_mode='zeros', seperable_conv=False,
                              res_skip=True, merge_res_skip=False, upsample_mode='linear', cond_kernel_size=1))
    cfg.update(over)
    cfg["WN_config"].update(wn_over)
    return cfg


def waveflow_author_config(n_flows=8, n_group=20, n_channels=128, n_layers=8, kernel=7, n_mel_channels=160,
                           hop_length=600, win_length=2400, speaker_embed=96, cond_layers=5, cond_hidden=512,
                           wn_cond_hidden=256):
    """The option set of the author's own WaveFlow checkpoints (printed by
    scripts/"WaveGlow from Ground Truth.ipynb" cell 2): separable 7x7 in-layers, speaker embeddings at model and
    WN level, a 5-layer k=9 (cond_kernel_size 5 -> 2k-1) residual + rezero conditioning stack, a 3-layer 1x1 WN
    conditioning stack with output activation, log-variance mel channels, pre-emphasis 0.9."""
    return waveflow_config(
        n_flows=n_flows, n_group=n_group, n_channels=n_channels, n_layers=n_layers, kernel_size_w=kernel,
        kernel_size_h=kernel, n_mel_channels=n_mel_channels, hop_length=hop_length, win_length=win_length,
        sampling_rate=48000, n_early_every=16, channel_mixing='permute', speaker_embed=speaker_embed,
        cond_layers=cond_layers, cond_activation_func='lrelu', negative_slope=0.25, cond_hidden_channels=cond_hidden,
        cond_output_channels=256, cond_residual=True, cond_res_rezero=True, cond_kernel_size=5,
        shift_spect=0.0, scale_spect=1.0, preceived_vol_scaling=False, preempthasis=0.9, use_logvar_channels=True,
        load_hidden_from_disk=False, iso226_empthasis=False,
        WN=dict(gated_unit='GTU', n_layers_dilations_h=1, speaker_embed_dim=speaker_embed, cond_layers=3,
                cond_activation_func='lrelu', cond_out_activation_func=True, negative_slope=0.5,
                cond_hidden_channels=wn_cond_hidden, cond_kernel_size=1, seperable_conv=True))


def _wn_tconv(wn):
    return bool(wn.get("transposed_conv_scales")) and bool(wn.get("transposed_conv_hidden_dim", 256)) \
        and bool(wn.get("transposed_conv_kernel_size", 4))


def _tconv_params(rng, sd, prefix, c_in, c_out, hid, ksz, scales):
    """TransposedUpsampleNet parameters (glow_ax.py:207-226): t_convs.{0,2,4..} = ConvTranspose1d [in][out][k]
    (a LeakyReLU module sits between them; none after the last one for the WN-level nets)."""
    idx = 0
    for i, sc in enumerate(scales):
        last = i + 1 == len(scales)
        ind, outd = (c_in if i == 0 else hid), (c_out if last else hid)
        kk = ksz[i] if isinstance(ksz, (list, tuple)) else ksz
        bound = 1.0 / np.sqrt(ind * kk / sc)
        sd[f"{prefix}.t_convs.{idx}.weight"] = _uniform(rng, (ind, outd, kk), bound)
        sd[f"{prefix}.t_convs.{idx}.bias"] = _uniform(rng, (outd,), bound)
        idx += 1 if last else 2


def _group_conv_params(rng, sd, cfg, c_wn):
    """``n_flow_group_conv`` (ax:131-134): plain Conv1d(c_wn, out * n_flows, 1, groups = n_flows | 1) -> out."""
    out = cfg.get("group_conv_output_dim")
    if not out:
        return c_wn
    groups = cfg["n_flows"] if cfg.get("group_conv_groupped", True) else 1
    bound = 1.0 / np.sqrt(c_wn // groups)
    sd["n_flow_group_conv.weight"] = _uniform(rng, (out * cfg["n_flows"], c_wn // groups, 1), bound)
    sd["n_flow_group_conv.bias"] = _uniform(rng, (out * cfg["n_flows"],), bound)
    return out


def _with_wn(cfg, **wn_over):
    cfg["WN_config"].update(wn_over)
    return cfg


WAVEFLOW_CONFIGS = {
    "toy": waveflow_config(n_flows=4, n_group=8, n_channels=64, n_layers=3),
    "full": waveflow_config(),                                   # config 4: 8 flows, 64 ch, h = 16
    # SURVEY 8f.4 option set, scaled down / at the author's size
    "author_toy": waveflow_author_config(n_flows=4, n_group=10, n_channels=64, n_layers=3, kernel=5,
                                         n_mel_channels=12, hop_length=40, win_length=160, speaker_embed=8,
                                         cond_layers=2, cond_hidden=32, wn_cond_hidden=24),
    "author": waveflow_author_config(),
    # merge_res_skip and a non-GTU unit on the 2-D core: dense C = 64 (un-fused layer path) and the separable C = 128 path
    "toy_merge": waveflow_config(n_flows=4, n_group=8, n_channels=64, n_layers=3, WN=dict(merge_res_skip=True, gated_unit='GLU')),
    "toy_groupconv": waveflow_config(n_flows=4, n_group=8, n_channels=64, n_layers=2, group_conv_output_dim=16,
                                     group_conv_groupped=True),
    # InvertibleConv1x1 mixing before the coupling with early outputs; PermuteHeight before the coupling with early
    # outputs; 1x1 conv after the coupling (the three orders / kinds config 4 does not use)
    "toy_conv_early": waveflow_config(n_flows=5, n_group=8, n_channels=64, n_layers=2, channel_mixing='1x1conv',
                                      mix_first=True, n_early_every=2, n_early_size=2),
    "toy_permute_mixfirst_early": waveflow_config(n_flows=4, n_group=12, n_channels=64, n_layers=2, hop_length=240,
                                                  win_length=960, channel_mixing='permute', mix_first=True, n_early_every=3,
                                                  n_early_size=4),
    "toy_conv_mixlast": waveflow_config(n_flows=4, n_group=8, n_channels=64, n_layers=2, channel_mixing='1x1conv',
                                        mix_first=False),
    "toy_no_res_skip": waveflow_config(n_flows=2, n_group=8, n_channels=64, n_layers=3,
                                       WN=dict(res_skip=False, merge_res_skip=True)),
    "toy_dilations": waveflow_config(n_flows=2, n_group=8, n_channels=64, n_layers=3, WN=dict(n_layers_dilations_w=[2, 5, 1])),
    # per-layer height dilations (deeper row queues, glow_ax.py:510-512, 597-602): dense C = 64 and separable C = 128
    "toy_dilations_h": waveflow_config(n_flows=2, n_group=10, n_channels=64, n_layers=3, hop_length=40, win_length=160,
                                       n_mel_channels=16, WN=dict(n_layers_dilations_h=[1, 2, 3])),
    "author_toy_dilations_h": _with_wn(waveflow_author_config(n_flows=2, n_group=10, n_channels=128, n_layers=2, kernel=3,
                                                              n_mel_channels=12, hop_length=40, win_length=160, speaker_embed=8,
                                                              cond_layers=2, cond_hidden=32, wn_cond_hidden=24),
                                       n_layers_dilations_h=2),
    # model-level TransposedUpsampleNet (upsample_first=True) in front of the 2-D core
    "toy_upsample_first": waveflow_config(n_flows=2, n_group=8, n_channels=64, n_layers=2, n_mel_channels=16, hop_length=40,
                                          win_length=160, upsample_first=True, transposed_conv_hidden_dim=24,
                                          transposed_conv_kernel_size=[4, 9], transposed_conv_scales=[2, 3],
                                          transposed_conv_output_dim=20, transposed_conv_residual=True,
                                          transposed_conv_residual_linear=True, transposed_conv_res_rezero=False),
    # the 2-D WN's own TransposedUpsampleNet: interpolated (factor 6 vs hop / n_group = 5) and cropped (factor 4 == 32 / 8)
    "toy_wn_tconv": waveflow_config(n_flows=2, n_group=8, n_channels=64, n_layers=2, n_mel_channels=16, hop_length=40,
                                    win_length=160, WN=dict(cond_layers=2, cond_hidden_channels=32,
                                                            transposed_conv_hidden_dim=32, transposed_conv_kernel_size=[4, 9],
                                                            transposed_conv_scales=[2, 3])),
    "toy_wn_tconv_crop": waveflow_config(n_flows=2, n_group=8, n_channels=64, n_layers=2, n_mel_channels=16, hop_length=32,
                                         win_length=128, WN=dict(cond_layers=1, transposed_conv_hidden_dim=24,
                                                                 transposed_conv_kernel_size=4, transposed_conv_scales=[2, 2])),
    "author_toy_gate": _with_wn(waveflow_author_config(n_flows=2, n_group=10, n_channels=128, n_layers=2, kernel=5,
                                                       n_mel_channels=12, hop_length=40, win_length=160, speaker_embed=8,
                                                       cond_layers=2, cond_hidden=32, wn_cond_hidden=24),
                                gated_unit='GSIRRU', merge_res_skip=True),
    # the WaveFlow printed by scripts/"UnTTS Inference.ipynb" (cell output at line 64): the same option family with
    # 12 flows, a 3-layer k=3 cond stack and the mel shifted / scaled on entry (shift_spect 11.52, scale_spect 0.25)
    # The reference's only published numbers for this path ("WaveFlow Inference Times.png") sweep n_group
    # 8 / 12 / 20 / 50 (hop % n_group == 0, efficient_model_ax.py:23), 64 ... 512 channels, dense and separable in-layers:
    # the corners of that sweep at toy depth
    "table_g50_c128": waveflow_config(n_flows=2, n_group=50, n_channels=128, n_layers=3, hop_length=300, win_length=1200),
    "table_g50_c256_sep": waveflow_config(n_flows=2, n_group=50, n_channels=256, n_layers=2, hop_length=300, win_length=1200,
                                          WN=dict(seperable_conv=True)),
    "table_g20_c512": waveflow_config(n_flows=2, n_group=20, n_channels=512, n_layers=2, hop_length=300, win_length=1200),
    "table_g12_c256_sep": waveflow_config(n_flows=2, n_group=12, n_channels=256, n_layers=2, hop_length=300, win_length=1200,
                                          WN=dict(seperable_conv=True)),
    "untts_toy": dict(waveflow_author_config(n_flows=4, n_group=10, n_channels=64, n_layers=3, kernel=5, n_mel_channels=12,
                                             hop_length=40, win_length=160, speaker_embed=8, cond_layers=3, cond_hidden=32,
                                             wn_cond_hidden=24),
                      shift_spect=11.52, scale_spect=0.25, cond_kernel_size=2, negative_slope=0.5, preempthasis=None,
                      use_logvar_channels=False),
}


def waveglow_ax_config(n_flows=4, n_group=8, n_channels=128, n_layers=3, kernel_size_w=3, n_mel_channels=80,
                       hop_length=256, win_length=1024, sampling_rate=22050, channel_mixing='1x1conv', mix_first=True,
                       n_early_every=100, n_early_size=2, **over):
    """Constructor kwargs of the reference ``efficient_model_ax.WaveGlow`` with ``waveflow=False`` (AffineCouplingBlock
    + 1-D ``glow_ax.WN``).  The 1-D WN takes no ``kernel_size_h``.  ``over`` / ``WN`` as in waveflow_config."""
    wn_over = over.pop("WN", {})
    cfg = dict(n_mel_channels=n_mel_channels, n_flows=n_flows, n_group=n_group, n_early_every=n_early_every,
               n_early_size=n_early_size, memory_efficient=0.0, spect_scaling=False, upsample_mode='normal',
               upsample_first=False, speaker_embed=0, cond_layers=0, cond_hidden_channels=256,
               cond_output_channels=256, cond_kernel_size=1, cond_residual=False, cond_padding_mode='zeros',
               waveflow=False, channel_mixing=channel_mixing, mix_first=mix_first, win_length=win_length,
               hop_length=hop_length, sampling_rate=sampling_rate,
               WN_config=dict(n_layers=n_layers, n_channels=n_channels, kernel_size_w=kernel_size_w,
                              n_layers_dilations_w=None, n_layers_dilations_h=1, speaker_embed_dim=0, rezero=False,
                              cond_layers=1, cond_activation_func='none', negative_slope=None,
                              cond_hidden_channels=256, cond_padding_mode='zeros', seperable_conv=False,
                              res_skip=True, merge_res_skip=False, upsample_mode='linear', cond_kernel_size=1))
    cfg.update(over)
    cfg["WN_config"].update(wn_over)
    return cfg


def waveglow_ax_notebook_config(n_flows=48, n_group=24, n_channels=256, n_layers=8, n_mel_channels=160, hop_length=600,
                                win_length=2400, speaker_embed=96, cond_hidden=256, n_early_every=16):
    """The config the reference's only recorded WaveGlow timing is for (BASELINE.md section 1:
    scripts/"WaveGlowFlow Inference Speed Testing.ipynb" cell 2, 4.60x real time at 48 kHz): 48 flows, n_group 24,
    8 x 256 WN, 'permute' mixing after the coupling, early outputs every 16 flows, speaker embeddings at model and WN
    level, a 3-layer k=3 replicate-padded residual + rezero conditioning stack, one 1x1 WN cond layer, linear
    interpolation of the conditioning."""
    return waveglow_ax_config(
        n_flows=n_flows, n_group=n_group, n_channels=n_channels, n_layers=n_layers, n_mel_channels=n_mel_channels,
        hop_length=hop_length, win_length=win_length, sampling_rate=48000, channel_mixing='permute', mix_first=False,
        n_early_every=n_early_every, n_early_size=2, preceived_vol_scaling=False, speaker_embed=speaker_embed,
        cond_layers=3, cond_activation_func='lrelu', negative_slope=0.5, cond_hidden_channels=cond_hidden,
        cond_output_channels=256, cond_residual=True, cond_res_rezero=True, cond_padding_mode='replicate',
        cond_kernel_size=2,
        WN=dict(speaker_embed_dim=speaker_embed, cond_layers=1, cond_activation_func='none', negative_slope=0.5,
                cond_hidden_channels=256, cond_padding_mode='replicate', seperable_conv=0, cond_kernel_size=1))


def waveglow_ax_untts_config(n_flows=24, n_group=24, n_channels=384, n_layers=8, n_mel_channels=256, hop_length=600,
                             win_length=2400, speaker_embed=32, cond_hidden=1024, cond_output=1024, t_hidden=1024,
                             t_kernels=(4, 9, 5), t_scales=(2, 3, 5), t_output=512, shift_spect=0.0, scale_spect=1.0):
    """The vocoder config printed by the reference's ``_2_ttm/untts/inference.ipynb`` (cell output at line 313):
    ``waveflow=False``, InvertibleConv1x1 mixing before the coupling, 24 flows x 8 layers x 384 channels, n_group 24,
    256-channel mel, a 4-layer k=1 conditioning stack with a 1x1-conv residual + rezero, and the conditioning
    upsampled at MODEL level (``upsample_first=True``) by a TransposedUpsampleNet (scales 2*3*5 = 30 != hop / n_group
    = 25, linear residual + rezero) followed by linear interpolation to the latent length."""
    return waveglow_ax_config(
        n_flows=n_flows, n_group=n_group, n_channels=n_channels, n_layers=n_layers, n_mel_channels=n_mel_channels,
        hop_length=hop_length, win_length=win_length, sampling_rate=48000, channel_mixing='1x1conv', mix_first=True,
        n_early_every=n_flows, n_early_size=2, speaker_embed=speaker_embed, shift_spect=shift_spect,
        scale_spect=scale_spect, cond_layers=4 if cond_hidden >= 256 else 2, cond_activation_func='lrelu',
        negative_slope=0.1, cond_hidden_channels=cond_hidden, cond_output_channels=cond_output, cond_residual='1x1conv',
        cond_res_rezero=True, cond_kernel_size=1, cond_padding_mode='zeros', upsample_first=True,
        transposed_conv_hidden_dim=t_hidden, transposed_conv_kernel_size=list(t_kernels),
        transposed_conv_scales=list(t_scales), transposed_conv_output_dim=t_output, transposed_conv_residual=True,
        transposed_conv_residual_linear=True, transposed_conv_res_rezero=True,
        WN=dict(speaker_embed_dim=0, cond_layers=1, cond_activation_func='lrelu', cond_out_activation_func=False,
                negative_slope=0.5, cond_hidden_channels=256, cond_kernel_size=1, cond_padding_mode='zeros',
                transposed_conv_scales=None))


WAVEGLOW_AX_CONFIGS = {
    # InvertibleConv1x1 mixing (the ax constructor's default), mix before the coupling, early outputs
    "toy_conv": waveglow_ax_config(n_flows=5, n_group=8, n_early_every=2),
    # ... mix after the coupling, k = 5 in-layers, 12 latent rows
    "toy_conv_mixlast": waveglow_ax_config(n_flows=4, n_group=12, n_early_every=3, n_early_size=4, mix_first=False,
                                           kernel_size_w=5, hop_length=240, win_length=960),
    # PermuteHeight mixing, both orders
    "toy_permute": waveglow_ax_config(n_flows=4, n_group=8, channel_mixing='permuteheight', mix_first=False,
                                      n_early_every=2),
    "toy_permute_mixfirst": waveglow_ax_config(n_flows=4, n_group=8, channel_mixing='permute', mix_first=True),
    # the notebook's option set scaled down, and at full size
    "notebook_toy": waveglow_ax_notebook_config(n_flows=6, n_group=12, n_channels=128, n_layers=3, n_mel_channels=12,
                                                hop_length=120, win_length=480, speaker_embed=8, cond_hidden=32,
                                                n_early_every=2),
    "notebook": waveglow_ax_notebook_config(),
    # the other gated units of glow_ax.py:45-165 and merge_res_skip (one config per unit family)
    **{f"toy_gate_{g.lower()}": waveglow_ax_config(n_flows=2, n_group=8, n_layers=2, WN=dict(gated_unit=g))
       for g in ("GTRU", "GTLRU", "GLU", "TTU", "STU", "GTSU", "SPTU", "GSIU", "GSIRU", "GTSRU", "GSIRRU", "GSIRLRU", "GSIRRLRU")},
    "toy_merge": waveglow_ax_config(n_flows=4, n_group=8, channel_mixing='permute', mix_first=False,
                                    WN=dict(merge_res_skip=True, gated_unit='GTRU')),
    # the optional per-flow 1x1 conv of the conditioning (ax:131-134), grouped by flow and dense; on top of the untts
    # toy's upsampling in the first case
    "toy_groupconv": dict(waveglow_ax_untts_config(n_flows=4, n_group=8, n_channels=128, n_layers=2, n_mel_channels=16,
                                                   hop_length=40, win_length=160, speaker_embed=8, cond_hidden=48,
                                                   cond_output=48, t_hidden=48, t_kernels=(4, 9), t_scales=(2, 3), t_output=32),
                          group_conv_output_dim=24, group_conv_groupped=True),
    # channel counts that are not a multiple of the GEMM's 128-channel M-block (ragged last block, split row 96 / 160)
    "toy_c96": waveglow_ax_config(n_flows=4, n_group=8, n_channels=96, n_layers=3, channel_mixing='permute', mix_first=False,
                                  n_early_every=2),
    "toy_c160": waveglow_ax_config(n_flows=2, n_group=12, n_channels=160, n_layers=2, kernel_size_w=5, hop_length=240,
                                   win_length=960, WN=dict(gated_unit='GTLRU')),
    # the widest latent the 1-D core takes (n_group = 32 rows in the flow-boundary kernel), with early outputs changing the
    # row count between flows; 32 channels = one 32-channel chunk (seven of the boundary kernel's eight waves idle in `end`)
    "toy_g32": waveglow_ax_config(n_flows=4, n_group=32, n_channels=32, n_layers=2, n_early_every=2, n_early_size=4),
    "toy_g32_permute": waveglow_ax_config(n_flows=4, n_group=32, n_channels=64, n_layers=2, channel_mixing='permute',
                                          mix_first=False, n_early_every=2, n_early_size=2),
    # per-layer width dilations instead of 2^i (a list, and the constant-int form)
    "toy_dilations": waveglow_ax_config(n_flows=2, n_group=8, n_layers=3, kernel_size_w=5, WN=dict(n_layers_dilations_w=[3, 1, 7])),
    "toy_dilations_const": waveglow_ax_config(n_flows=2, n_group=8, n_layers=2, WN=dict(n_layers_dilations_w=2)),
    # res_skip=False: the gated activations are the skip (with merge_res_skip; and the single-layer case without it)
    "toy_no_res_skip": waveglow_ax_config(n_flows=2, n_group=8, n_layers=3, WN=dict(res_skip=False, merge_res_skip=True)),
    "toy_no_res_skip_1layer": waveglow_ax_config(n_flows=2, n_group=8, n_layers=1, WN=dict(res_skip=False)),
    # sigmoid conditioning activations at model and WN level + the "perceived volume" companding of the output
    "toy_sigmoid_vol": waveglow_ax_config(n_flows=2, n_group=8, n_layers=2, n_mel_channels=16, cond_layers=2,
                                          cond_activation_func='sigmoid', cond_hidden_channels=24, cond_output_channels=20,
                                          preceived_vol_scaling=True,
                                          WN=dict(cond_layers=2, cond_hidden_channels=24, cond_activation_func='sigmoid')),
    # the WN's own TransposedUpsampleNet behind its cond stack: factor 6 != hop / n_group = 5 (interpolated), and
    # factor 4 == hop / n_group (centre-cropped: the mel carries one frame more than the latent)
    "toy_wn_tconv": waveglow_ax_config(n_flows=2, n_group=8, n_layers=2, n_mel_channels=16, hop_length=40, win_length=160,
                                       WN=dict(cond_layers=2, cond_hidden_channels=32, cond_activation_func='lrelu',
                                               negative_slope=0.3, transposed_conv_hidden_dim=32,
                                               transposed_conv_kernel_size=[4, 9], transposed_conv_scales=[2, 3])),
    "toy_wn_tconv_crop": waveglow_ax_config(n_flows=2, n_group=8, n_layers=2, n_mel_channels=16, hop_length=32,
                                            win_length=128,
                                            WN=dict(cond_layers=1, transposed_conv_hidden_dim=24,
                                                    transposed_conv_kernel_size=4, transposed_conv_scales=[2, 2])),
    "toy_groupconv_dense": waveglow_ax_config(n_flows=4, n_group=8, n_mel_channels=20, group_conv_output_dim=12,
                                              group_conv_groupped=False, WN=dict(speaker_embed_dim=4)),
    # the untts notebook's vocoder: model-level transposed-conv upsampling, 1x1-conv cond residual; toy and full size
    "untts_toy": waveglow_ax_untts_config(n_flows=4, n_group=8, n_channels=128, n_layers=2, n_mel_channels=16,
                                          hop_length=40, win_length=160, speaker_embed=8, cond_hidden=48, cond_output=48,
                                          t_hidden=48, t_kernels=(4, 9), t_scales=(2, 3), t_output=32, shift_spect=1.5,
                                          scale_spect=0.5),
    "untts": waveglow_ax_untts_config(),
}


def waveglow_ax_flow_channels(cfg):
    """n_remaining_channels per flow (efficient_model_ax.py:170-189)."""
    n_rem, out = cfg["n_group"], []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_rem -= cfg["n_early_size"]
        out.append(n_rem)
    return out


def waveglow_ax_state_dict(cfg, seed=1234, end_std=None):
    """Random-init state dict with the reference's keys for ``waveflow=False``: ``WN.k.WN.{start,in_layers.i,
    res_skip_layers.i,cond_layers.l}.{bias,weight_g,weight_v}`` (3-D conv weights), ``WN.k.WN.end.{weight,bias}``,
    ``WN.k.WN.speaker_embed.weight``, ``convinv.k.weight`` (1x1conv mixing only), ``speaker_embed.weight``, ``alpha``,
    ``cond_layers.l.*``."""
    rng = np.random.default_rng(seed)
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    ks = wn.get("kernel_size_w") or wn.get("kernel_size")
    if end_std is None:
        end_std = 0.25 / np.sqrt(C)
    sd = {}
    c_in, c_wn = waveflow_cond_channels(cfg)
    if cfg["speaker_embed"]:
        sd["speaker_embed.weight"] = rng.standard_normal((512, cfg["speaker_embed"]), dtype=np.float32)
    if cfg.get("cond_res_rezero"):
        sd["alpha"] = np.array([0.3], np.float32)
    if cfg["cond_layers"]:
        k = 2 * cfg["cond_kernel_size"] - 1
        dims = [c_in] + [cfg["cond_hidden_channels"]] * (cfg["cond_layers"] - 1) + [c_wn]
        for l in range(cfg["cond_layers"]):
            _wn_conv(rng, sd, f"cond_layers.{l}", dims[l + 1], dims[l], k)
    if cfg["cond_layers"] and cfg["cond_residual"] == '1x1conv':                 # ax:80-81 (plain Conv1d)
        bound = 1.0 / np.sqrt(c_in)
        sd["res_conv.weight"] = _uniform(rng, (c_wn, c_in, 1), bound)
        sd["res_conv.bias"] = _uniform(rng, (c_wn,), bound)
    if cfg.get("upsample_first") is True:                                        # ax:116-126, glow_ax.py:207-226
        scales = cfg["transposed_conv_scales"]
        ksz = cfg["transposed_conv_kernel_size"]
        hid, t_out = cfg["transposed_conv_hidden_dim"], cfg.get("transposed_conv_output_dim") or c_wn
        if cfg.get("transposed_conv_res_rezero"):
            sd["upsample_net.res_weight"] = np.array([0.4], np.float32)
        idx = 0
        for i, sc in enumerate(scales):
            last = i + 1 == len(scales)
            ind, outd = (c_wn if i == 0 else hid), (t_out if last else hid)
            kk = ksz[i] if isinstance(ksz, (list, tuple)) else ksz
            bound = 1.0 / np.sqrt(ind * kk / sc)
            sd[f"upsample_net.t_convs.{idx}.weight"] = _uniform(rng, (ind, outd, kk), bound)
            sd[f"upsample_net.t_convs.{idx}.bias"] = _uniform(rng, (outd,), bound)
            idx += 2                                                             # a LeakyReLU module follows every conv
        c_wn = t_out
    c_wn = _group_conv_params(rng, sd, cfg, c_wn)
    sdim = wn.get("speaker_embed_dim", 0)
    conv_mix = cfg.get("channel_mixing", '1x1conv').lower() in "1x1convinvertibleconv1x1invconv"
    for k, n_rem in enumerate(waveglow_ax_flow_channels(cfg)):
        p = f"WN.{k}.WN"
        h = n_rem // 2
        _wn_conv(rng, sd, p + ".start", C, h, 1)
        sd[p + ".end.weight"] = rng.standard_normal((2 * h, C, 1), dtype=np.float32) * np.float32(end_std)
        sd[p + ".end.bias"] = rng.standard_normal((2 * h,), dtype=np.float32) * np.float32(0.02)
        if sdim:
            sd[p + ".speaker_embed.weight"] = rng.standard_normal((512, sdim), dtype=np.float32)
        ck = 2 * wn.get("cond_kernel_size", 1) - 1
        cond_out = wn.get("transposed_conv_hidden_dim", 256) if _wn_tconv(wn) else 2 * C * n_layers
        dims = [c_wn + sdim] + [wn["cond_hidden_channels"]] * (wn["cond_layers"] - 1) + [cond_out]
        for l in range(wn["cond_layers"]):
            _wn_conv(rng, sd, f"{p}.cond_layers.{l}", dims[l + 1], dims[l], ck, gain=0.5)
        if _wn_tconv(wn):
            _tconv_params(rng, sd, p + ".upsample_net", cond_out, 2 * C * n_layers, cond_out,
                          wn.get("transposed_conv_kernel_size", 4), wn["transposed_conv_scales"])
        for i in range(n_layers):
            _wn_conv(rng, sd, f"{p}.in_layers.{i}", 2 * C, C, ks)
            if wn.get("res_skip", True):
                _wn_conv(rng, sd, f"{p}.res_skip_layers.{i}", 2 * C if (i < n_layers - 1 and not wn.get("merge_res_skip")) else C, C, 1)
        if conv_mix:
            a = rng.standard_normal((n_rem, n_rem)).astype(np.float64)
            q, _ = np.linalg.qr(a)
            q = q + 0.05 * rng.standard_normal((n_rem, n_rem))
            sd[f"convinv.{k}.weight"] = q.astype(np.float32)[:, :, None]
    return sd


def waveflow_cond_channels(cfg):
    """(model-level cond input channels, channels handed to every WN) - ax:64-66, 73-74, 96."""
    c_in = cfg["n_mel_channels"] * (2 if cfg.get("use_logvar_channels") else 1) + cfg["speaker_embed"]
    c_out = c_in
    if cfg["cond_layers"]:
        c_out = c_in if cfg["cond_residual"] in (True, 1) else cfg["cond_output_channels"]
    return c_in, c_out


def waveflow_state_dict(cfg, seed=1234, end_std=None):
    """Random-init state dict with the reference's keys (SURVEY.md 8a "Checkpoint keys"):
    ``WN.k.WN.{start,in_layers.i,res_skip_layers.i,cond_layers.l}.{bias,weight_g,weight_v}``,
    ``WN.k.WN.end.{weight,bias}`` (PermuteHeight has no parameters); with the 8f.4 options also
    ``speaker_embed.weight``, ``alpha``, ``cond_layers.l.*``, ``WN.k.WN.speaker_embed.weight`` and separable
    in-layers ``WN.k.WN.in_layers.i.{0,1}.*`` (depthwise, pointwise)."""
    rng = np.random.default_rng(seed)
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    kh, kw = wn["kernel_size_h"], wn["kernel_size_w"]
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

    c_in, c_wn = waveflow_cond_channels(cfg)
    if cfg["speaker_embed"]:
        sd["speaker_embed.weight"] = rng.standard_normal((512, cfg["speaker_embed"]), dtype=np.float32)
    if cfg.get("cond_res_rezero"):
        sd["alpha"] = np.array([0.3], np.float32)          # a trained value; the init (0.01..0.03) would hide the stack
    if cfg["cond_layers"]:
        k = 2 * cfg["cond_kernel_size"] - 1
        dims = [c_in] + [cfg["cond_hidden_channels"]] * (cfg["cond_layers"] - 1) + [c_wn]
        for l in range(cfg["cond_layers"]):
            wn_conv(f"cond_layers.{l}", (dims[l + 1], dims[l], k), dims[l] * k)
    if cfg.get("upsample_first") is True:                                        # ax:116-126
        t_out = cfg.get("transposed_conv_output_dim") or c_wn
        if cfg.get("transposed_conv_res_rezero"):
            sd["upsample_net.res_weight"] = np.array([0.4], np.float32)
        hid, ksz, scales = cfg["transposed_conv_hidden_dim"], cfg["transposed_conv_kernel_size"], cfg["transposed_conv_scales"]
        idx = 0
        for i, sc in enumerate(scales):
            last = i + 1 == len(scales)
            ind, outd = (c_wn if i == 0 else hid), (t_out if last else hid)
            kk = ksz[i] if isinstance(ksz, (list, tuple)) else ksz
            bound = 1.0 / np.sqrt(ind * kk / sc)
            sd[f"upsample_net.t_convs.{idx}.weight"] = _uniform(rng, (ind, outd, kk), bound)
            sd[f"upsample_net.t_convs.{idx}.bias"] = _uniform(rng, (outd,), bound)
            idx += 2
        c_wn = t_out
    c_wn = _group_conv_params(rng, sd, cfg, c_wn)
    sdim = wn.get("speaker_embed_dim", 0)
    for k in range(cfg["n_flows"]):
        p = f"WN.{k}.WN"
        wn_conv(p + ".start", (C, 1, 1, 1), 1)
        sd[p + ".end.weight"] = rng.standard_normal((2, C, 1, 1), dtype=np.float32) * np.float32(end_std)
        sd[p + ".end.bias"] = rng.standard_normal((2,), dtype=np.float32) * np.float32(0.02)
        if sdim:
            sd[p + ".speaker_embed.weight"] = rng.standard_normal((512, sdim), dtype=np.float32)
        ck = 2 * wn.get("cond_kernel_size", 1) - 1
        cond_out = wn.get("transposed_conv_hidden_dim", 256) if _wn_tconv(wn) else 2 * C * n_layers
        if _wn_tconv(wn):
            _tconv_params(rng, sd, p + ".upsample_net", cond_out, 2 * C * n_layers, cond_out,
                          wn.get("transposed_conv_kernel_size", 4), wn["transposed_conv_scales"])
        dims = [c_wn + sdim] + [wn["cond_hidden_channels"]] * (wn["cond_layers"] - 1) + [cond_out]
        for l in range(wn["cond_layers"]):
            # (the single-layer recipe keeps its historical fan so the committed config-4 goldens stay valid)
            fan = dims[l] * 4 if wn["cond_layers"] == 1 else dims[l] * ck
            wn_conv(f"{p}.cond_layers.{l}", (dims[l + 1], dims[l], ck), fan)
        for i in range(n_layers):
            if wn.get("seperable_conv") and not (kh == 1 and kw == 1):
                wn_conv(f"{p}.in_layers.{i}.0", (C, 1, kh, kw), kh * kw)
                wn_conv(f"{p}.in_layers.{i}.1", (2 * C, C, 1, 1), C)
            else:
                wn_conv(f"{p}.in_layers.{i}", (2 * C, C, kh, kw), C * kh * kw)
            rs = 2 * C if (i < n_layers - 1 and not wn.get("merge_res_skip")) else C
            if wn.get("res_skip", True):
                wn_conv(f"{p}.res_skip_layers.{i}", (rs, C, 1, 1), C)
    if cfg.get("channel_mixing", '1x1conv').lower() in "1x1convinvertibleconv1x1invconv":      # ax:24-25, 148-149
        for k, n_rem in enumerate(waveglow_ax_flow_channels(cfg)):
            a = rng.standard_normal((n_rem, n_rem)).astype(np.float64)
            q, _ = np.linalg.qr(a)
            q = q + 0.05 * rng.standard_normal((n_rem, n_rem))
            sd[f"convinv.{k}.weight"] = q.astype(np.float32)[:, :, None]
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
    mode = cfg.get("upsample_mode", "normal")
    opg = {"normal": n_mel, "simple": 1, "simple_half": 2}[mode]       # ConvTranspose1d weight [in, out/groups, win]
    sd["upsample.weight"] = _uniform(rng, (n_mel, opg, win), 1.0 / np.sqrt(opg * taps))
    sd["upsample.bias"] = _uniform(rng, (n_mel,), 0.1)
    sdim = wn.get("speaker_embed_dim", 0)
    for k, (n_rem, n_half) in enumerate(waveglow_flow_channels(cfg)):
        p = f"WN.{k}"
        if wn.get("rezero"):
            for i in range(n_layers):                     # trained-looking values (the init is 0.1 +- 0.01, glow.py:182)
                sd[f"{p}.alpha_i.{i}"] = np.array([0.35 + 0.1 * rng.random()], np.float32)
        if sdim:
            sd[p + ".speaker_embed.weight"] = rng.standard_normal((512, sdim), dtype=np.float32) * np.float32(4.0)
        _wn_conv(rng, sd, p + ".start", C, n_half, 1)
        sd[p + ".end.weight"] = (rng.standard_normal((2 * n_half, C, 1), dtype=np.float32)
                                 * np.float32(end_std))
        sd[p + ".end.bias"] = (rng.standard_normal((2 * n_half,), dtype=np.float32)
                               * np.float32(0.02))
        _wn_conv(rng, sd, p + ".cond_layers.0", cond_hidden, n_mel * G + sdim, 1)
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


# ----------------------------------------------------------------------------- Tacotron2-TM ----
def tacotron_hparams(**overrides):
    """The model-shaping defaults of the reference's ``create_hparams()``
    (_2_ttm/tacotron2_tm/hparams.py:139-279) as a plain attribute object, with ``fp16_run=False``.
    (Importing hparams.py itself drags in the text frontend; only these values shape the model.)"""
    from types import SimpleNamespace
    hp = dict(
        fp16_run=False, mask_padding=True, n_mel_channels=80, n_frames_per_step=1, context_frames=1,
        gate_threshold=0.5, gate_delay=10, max_decoder_steps=3000,
        n_symbols=179, symbols_embedding_dim=512,
        encoder_speaker_embed_dim=64, encoder_concat_speaker_embed='before_conv', encoder_kernel_size=5,
        encoder_n_convolutions=3, encoder_conv_hidden_dim=512, encoder_LSTM_dim=1024,
        sylpsnet_layer_dims=[32, 32], emotion_classes=['neutral'] * 16,
        torchMoji_attDim=2304, torchMoji_crushedDim=32, torchMoji_BatchNorm=True,
        n_speakers=512, speaker_embedding_dim=256,
        use_memory_bottleneck=True, memory_bottleneck_dim=512, memory_bottleneck_bias=False,
        hide_startstop_tokens=False,
        prenet_dim=256, prenet_layers=2, prenet_batchnorm=False, prenet_bn_momentum=0.5, p_prenet_dropout=0.5,
        prenet_speaker_embed_dim=0, prenet_noise=0.0, prenet_blur_min=0.0, prenet_blur_max=0.0,
        attention_rnn_dim=1280, AttRNN_hidden_dropout_type='dropout', p_AttRNN_hidden_dropout=0.10,
        AttRNN_extra_decoder_input=True,
        decoder_rnn_dim=768, DecRNN_hidden_dropout_type='dropout', p_DecRNN_hidden_dropout=0.25,
        decoder_residual_connection=False, second_decoder_rnn_dim=768, second_decoder_residual_connection=True,
        attention_type=0, attention_dim=192, windowed_attention_range=16, windowed_att_pos_offset=1.25,
        windowed_att_pos_learned=True, attention_learned_temperature=False,
        attention_location_n_filters=32, attention_location_kernel_size=31, num_att_mixtures=1,
        attention_layers=1, normalize_attention_input=True, normalize_AttRNN_output=False,
        use_postnet=True, postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=6,
        postnet_residual_connections=3, p_teacher_forcing=1.0, teacher_force_till=20, drop_frame_rate=0.0)
    hp.update(overrides)
    return SimpleNamespace(**hp)


def tacotron_memory_in_dim(hp):
    return hp.encoder_LSTM_dim + hp.speaker_embedding_dim + hp.torchMoji_crushedDim + 1


# A checkpoint's own hparams shape the model (the server builds it from checkpoint['hparams'], text2speech.py:299-316):
# every width away from the repo defaults, roughly halved.  Runs on the per-launch decoder (the
# persistent form is built for the default widths only).
TACOTRON_SMALL_OVERRIDES = dict(
    symbols_embedding_dim=256, encoder_speaker_embed_dim=32, encoder_conv_hidden_dim=256, encoder_LSTM_dim=512,
    torchMoji_crushedDim=16, speaker_embedding_dim=128, memory_bottleneck_dim=256, prenet_dim=128, attention_rnn_dim=640,
    decoder_rnn_dim=384, second_decoder_rnn_dim=384, attention_dim=96, windowed_attention_range=8,
    attention_location_n_filters=16, attention_location_kernel_size=15, postnet_embedding_dim=256)


def tacotron_state_dict(hp, seed=1234, shapes=None, attention_drive=None, stop_drive=None):
    """Random-init state dict for the reference's ``Tacotron2`` (keys/shapes of its own
    ``state_dict()``; ``shapes`` = {key: shape} from a constructed module, or None to derive them
    from ``cookietts_amd.tacotron2.Tacotron2(hp)``).  Deterministic numpy recipe:
    weights ~ U(+-1/sqrt(fan_in)), BatchNorm running stats non-trivial, the two zero-initialised
    learnable scalars (decoder.exp_smoothing_factor, attention windowed_att_pos_offset) non-zero,
    and the sylps head forced positive so log(pred_sylps) is finite.

    ``attention_drive=(a, c, gain)`` turns the near-uniform attention of the plain recipe (energies ~ +-0.3) into a peaked,
    monotonically advancing one, the regime a trained model runs in: the energy vector ``v`` is scaled by ``gain``,
    location filter 0 reads the previous weights one and two tokens to the left (taps 13, 14 of channel 0 = ``a``),
    and attention dims 0..3 carry that feature (location_dense rows 0..3 = e_0, v[0..3] = c / 4), so each step the
    peak moves right until the window reaches its right clamp (model.py:131-146).

    ``stop_drive=(rate, sharp, [t_1, ..., t_k])`` builds a step clock into the second decoder LSTM so that a gate layer can be
    given logits that cross the stop threshold at designed steps (tests/golden/make_golden.py tacotron_stop): unit 0 counts
    (forget / input / output gates pinned open by +-12 biases, cell input tanh^-1-free constant ``rate``: h_0(t) =
    tanh((t+1) rate)), unit j = 1..k is a sharp step of that clock (forget gate shut, cell input ``sharp`` x (h_0 - theta_j) read
    through weight_hh, theta_j = tanh((t_j - 1/2) rate)): h_j(t) ~ +-tanh(1) from decoder step t_j on.  All other weights of
    those 1 + k units, and the same units of the first decoder LSTM, are zero, so d_j = dec_h_j + dec2_h_j = h_j exactly."""
    if shapes is None:
        from .tacotron2 import Tacotron2
        shapes = {k: tuple(v.shape) for k, v in Tacotron2(hp).state_dict().items()}
    rng = np.random.default_rng(seed)
    sd = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            sd[key] = np.zeros(shape, dtype=np.int64)
        elif key.endswith("running_var"):
            sd[key] = (0.5 + rng.random(shape, dtype=np.float32)).astype(np.float32)
        elif key.endswith("running_mean"):
            sd[key] = _uniform(rng, shape, 0.1)
        elif key == "decoder.exp_smoothing_factor":
            sd[key] = np.full(shape, 0.3, dtype=np.float32)
        elif key.endswith("windowed_att_pos_offset"):
            sd[key] = np.full(shape, 1.25, dtype=np.float32)
        elif key == "sylps_net.res_weight":
            sd[key] = np.asarray(0.01, dtype=np.float32).reshape(shape)
        elif key == "encoder.sylps_layer.linear_layer.weight":
            sd[key] = np.zeros(shape, dtype=np.float32)
        elif key == "encoder.sylps_layer.linear_layer.bias":
            sd[key] = np.full(shape, 4.0, dtype=np.float32)
        elif ".1.weight" in key and ("convolutions" in key) or key == "tm_bn.weight":      # BatchNorm gamma
            sd[key] = (0.8 + 0.4 * rng.random(shape, dtype=np.float32)).astype(np.float32)
        elif "embedding" in key and key.endswith("weight"):
            sd[key] = _uniform(rng, shape, 0.5)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            sd[key] = _uniform(rng, shape, 1.0 / np.sqrt(fan_in))
        else:
            sd[key] = _uniform(rng, shape, 0.05)
    if attention_drive is not None:
        a, c, gain = (np.float32(x) for x in attention_drive)
        att = "decoder.attention_layer."
        v = sd[att + "v.linear_layer.weight"] * gain
        v[0, :4] = c / np.float32(4)
        sd[att + "v.linear_layer.weight"] = v.astype(np.float32)
        w = sd[att + "location_layer.location_conv.conv.weight"]
        w[0] = 0
        w[0, 0, 13:15] = a
        d = sd[att + "location_layer.location_dense.linear_layer.weight"]
        d[:4] = 0
        d[:4, 0] = 1
    if stop_drive is not None:
        rate, sharp, times = stop_drive
        H = hp.second_decoder_rnn_dim
        units = list(range(1 + len(times)))
        for cell in ("decoder.decoder_rnn", "decoder.second_decoder_rnn"):
            for name in (".weight_ih", ".weight_hh", ".bias_ih", ".bias_hh"):
                for g in range(4):
                    sd[cell + name][[g * H + j for j in units]] = 0
        whh, bih = sd["decoder.second_decoder_rnn.weight_hh"], sd["decoder.second_decoder_rnn.bias_ih"]
        big = np.float32(12.0)
        bih[0 * H + 0], bih[1 * H + 0], bih[3 * H + 0] = big, big, big       # clock: i, f, o open (gate order i, f, g, o)
        bih[2 * H + 0] = np.float32(np.arctanh(rate))                          # cell input tanh(.) = rate per step
        for j, t in enumerate(times, start=1):
            theta = np.tanh((t - 0.5) * rate)
            bih[0 * H + j], bih[1 * H + j], bih[3 * H + j] = big, -big, big    # i open, f shut, o open
            whh[2 * H + j, 0] = np.float32(sharp)
            bih[2 * H + j] = np.float32(-sharp * theta)
    return sd


def prenet_dropout_masks(n_steps, batch, prenet_dim=256, seed=1234):
    """Keep-masks (uint8 0/1) for the prenet's always-on dropout (model.py:189-190): [steps, 2, B, dim]."""
    rng = np.random.default_rng(seed + 15485863)
    return (rng.random((n_steps, 2, batch, prenet_dim)) < 0.5).astype(np.uint8)
