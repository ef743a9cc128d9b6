"""CPU oracle for the "ax" WaveGlow core with ``waveflow=False`` (AffineCouplingBlock + 1-D WN).

TEST INFRASTRUCTURE ONLY (see oracle/waveglow_oracle.py header for the import rule).  numpy fp32 restatement of
the algorithm, sharing the conditioning-stack pieces with oracle/waveflow_oracle.py.
Reference lines followed (relative to /root/reference/CookieTTS/_4_mtw/waveglow/):
  waveglow_ax_inverse  efficient_model_ax.py:279-357: squeeze :310, early-output split :317-320 (the LAST split is
                       the initial latent; early chunks are concatenated back in front at k % n_early_every == 0,
                       :340-341), per flow: [un-mix if not mix_first :324-325] coupling :328, NaN -> 0 :333-334,
                       [un-mix if mix_first :337-338]; un-squeeze :346; de-emphasis :351-355
  coupling             efficient_modules.py:94-105: (log_s, t) = WN(a0); a1 = (a1 - t) / exp(log_s)
  model-level upsampling  efficient_model_ax.py:116-126, 174-186, 318-319 (upsample_first=True): TransposedUpsampleNet
                       glow_ax.py:201-242 then F.interpolate(size = latent length, 'linear', align_corners=True)
  wn1d                 glow_ax.py:375-418: start :376; WN speaker embedding :378-381; cond stack with activation
                       rule :383-387; linear interpolation to the audio length when upsample_first is False :389-390
                       (= _upsample_mels :362-373, align_corners=True); dilated in_layers (kernel_size_w or
                       kernel_size, dilation 2^i, zero pad) + GTU gate (:36-43) :395-399; res/skip with `output`
                       starting from the first skip :401-416; end.chunk(2) :418
  un-mix               InvertibleConv1x1.inverse efficient_modules.py:269-286 (W.float().inverse() then conv1d);
                       PermuteHeight.inverse :400-403 / permute_channels :360-373
Parity pin: tests/golden/waveglow_ax_*.npz = outputs of the reference's own ``infer`` / ``inverse``
(tests/golden/make_golden.py waveglow_ax).
"""
from __future__ import annotations

import numpy as np

from .waveflow_oracle import (F32, _shift, _w, activation, conv1d_same, deemphasis, flow_conds, gated_unit, lerp_align_corners,
                              model_cond, permutation, to_latent_length, transposed_upsample_net, wn_upsample)


def flow_channels(cfg):
    """n_remaining_channels per flow (efficient_model_ax.py:170-189)."""
    n_rem, out = cfg["n_group"], []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_rem -= cfg["n_early_size"]
        out.append(n_rem)
    return out


def mixing_kind(cfg):
    """efficient_model_ax.py:25: substring test against the two option strings."""
    name = cfg.get("channel_mixing", '1x1conv').lower()
    return '1x1conv' if name in "1x1convinvertibleconv1x1invconv" else 'permuteheight'


def wn1d(sd, p, wn, a0, frames, speaker_ids, L, upsample_factor=None):
    """One 1-D WN: a0 [B, h, L], frames [B, c, F] (frame rate) -> (log_s, t), each [B, h, L]."""
    C, n_layers = wn["n_channels"], wn["n_layers"]
    ks = wn.get("kernel_size_w") or wn.get("kernel_size")
    x = (np.matmul(_w(sd, p + ".start")[:, :, 0], a0) + sd[p + ".start.bias"][None, :, None]).astype(F32)
    spect = frames
    if wn.get("speaker_embed_dim", 0) and speaker_ids is not None:
        emb = sd[p + ".speaker_embed.weight"][np.asarray(speaker_ids)]
        spect = np.concatenate([spect, np.repeat(emb[:, :, None], spect.shape[2], axis=2)], axis=1)
    act = activation(wn.get("cond_activation_func", 'none'), wn.get("negative_slope"))
    for l in range(wn["cond_layers"]):
        spect = conv1d_same(spect, _w(sd, f"{p}.cond_layers.{l}"), sd[f"{p}.cond_layers.{l}.bias"],
                            wn.get("cond_padding_mode", 'zeros'))
        if act is not None and (wn.get("cond_out_activation_func", True) or l != wn["cond_layers"] - 1):
            spect = act(spect).astype(F32)
    # upsample_first is False: the WN brings its conditioning to the latent's rate itself (glow_ax.py:389-390);
    # upsample_factor None = it arrives at that rate already (upsample_first=True)
    cond = spect if upsample_factor is None else wn_upsample(sd, p, wn, spect, L, upsample_factor, False)
    out = None
    for i in range(n_layers):
        dl = wn.get("n_layers_dilations_w")                                    # glow_ax.py:328-333
        d = 2 ** i if dl is None else (dl if isinstance(dl, int) else dl[i])
        w = _w(sd, f"{p}.in_layers.{i}")                                   # [2C, C, ks]
        u = sd[f"{p}.in_layers.{i}.bias"][None, :, None] + np.zeros((x.shape[0], 2 * C, L), F32)
        for t in range(ks):
            u = u + np.matmul(np.ascontiguousarray(w[:, :, t]), _shift(x, (t - ks // 2) * d))
        u = (u.astype(F32) + cond[:, 2 * C * i:2 * C * (i + 1)]).astype(F32)   # GTU: in_act + spect, then gate
        g = gated_unit(wn.get("gated_unit", 'GTU'), u, C)
        if wn.get("res_skip", True):
            rs = (np.matmul(_w(sd, f"{p}.res_skip_layers.{i}")[:, :, 0], g)
                  + sd[f"{p}.res_skip_layers.{i}.bias"][None, :, None]).astype(F32)
        else:
            rs = g                                                              # glow_ax.py:401: res_skip_acts = acts
        if i < n_layers - 1 and not wn.get("merge_res_skip", False) and wn.get("res_skip", True):   # glow_ax.py:401-416
            x = (x + rs[:, :C]).astype(F32)
            out = rs[:, C:] if out is None else (out + rs[:, C:]).astype(F32)
        else:
            out = rs if out is None else (out + rs).astype(F32)
    e = (np.matmul(sd[p + ".end.weight"][:, :, 0], out) + sd[p + ".end.bias"][None, :, None]).astype(F32)
    h = e.shape[1] // 2
    return e[:, :h], e[:, h:]


def waveglow_ax_inverse(sd, cfg, z, mel, speaker_ids=None, flow_trace=None):
    """z [B, T] (sigma applied; carries the early-output noise too), mel [B, n_mel(*2), F'] (already padded by
    infer) -> audio [B, T]."""
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
    G, n_flows = cfg["n_group"], cfg["n_flows"]
    every, esize = cfg["n_early_every"], cfg["n_early_size"]
    wn = cfg["WN_config"]
    mix, mix_first = mixing_kind(cfg), cfg.get("mix_first", True)
    z = np.asarray(z, dtype=F32)
    B, T = z.shape
    L = T // G
    a = np.ascontiguousarray(z.reshape(B, L, G).transpose(0, 2, 1))        # a[b, g, l] = z[b, G*l + g]
    chans = flow_channels(cfg)
    n_early = sum(1 for k in range(1, n_flows) if k % every == 0)
    remained = [a[:, i * esize:(i + 1) * esize] for i in range(n_early)]
    zz = a[:, n_early * esize:]
    assert zz.shape[1] == chans[-1]
    frames = model_cond(sd, cfg, mel, speaker_ids)
    if cfg.get("upsample_first") is True:
        # efficient_model_ax.py:318-319 -> _upsample_mels :174-186: TransposedUpsampleNet, then (its length differs
        # from the latent's) linear interpolation with align_corners=True; the WNs then take it as it is
        frames = transposed_upsample_net(sd, "upsample_net", frames, cfg["transposed_conv_scales"],
                                         cfg["transposed_conv_kernel_size"], True, cfg.get("transposed_conv_residual", False),
                                         cfg.get("transposed_conv_residual_linear", False))
        frames = to_latent_length(frames, L, int(np.prod(cfg["transposed_conv_scales"])) != cfg["hop_length"] // G, False)
    frames_k = flow_conds(sd, cfg, frames)

    def unmix(k, v):
        if mix == 'permuteheight':
            return v[:, permutation(k, v.shape[1]), :]
        w_inv = np.linalg.inv(sd[f"convinv.{k}.weight"][:, :, 0].astype(F32)).astype(F32)
        return np.matmul(w_inv, v).astype(F32)

    for k in reversed(range(n_flows)):
        assert zz.shape[1] == chans[k]
        if not mix_first:
            zz = unmix(k, zz)
        h = zz.shape[1] // 2
        log_s, t = wn1d(sd, f"WN.{k}.WN", wn, zz[:, :h], frames_k[k], speaker_ids, L,
                        None if cfg.get("upsample_first") is True else cfg["hop_length"] // G)
        with np.errstate(over="ignore", invalid="ignore"):
            a1 = ((zz[:, h:] - t) / np.exp(log_s)).astype(F32)
        zz = np.concatenate([zz[:, :h], a1], axis=1)
        zz = np.where(np.isnan(zz), F32(0), zz).astype(F32)
        if mix_first:
            zz = unmix(k, zz)
        if k % every == 0 and k:
            zz = np.concatenate([remained.pop(), zz], axis=1)
        if flow_trace is not None:
            flow_trace.append(zz.copy())
    assert not remained
    audio = np.ascontiguousarray(zz.transpose(0, 2, 1)).reshape(B, T)
    if cfg.get("preceived_vol_scaling"):                                   # ax:342-344
        with np.errstate(divide="ignore", invalid="ignore"):
            pos = np.power(F32(10.0), np.log2(np.where(audio > 0, audio, F32(1)))).astype(F32)
            neg = -np.power(F32(10.0), np.log2(np.where(audio < 0, -audio, F32(1)))).astype(F32)
        audio = np.where(audio > 0, pos, np.where(audio < 0, neg, audio)).astype(F32)
    if cfg.get("preempthasis"):
        audio = deemphasis(audio, cfg["preempthasis"])
    return audio


def waveglow_ax_infer(sd, cfg, mel, z, artifact_trimming=1, speaker_ids=None):
    """infer() wrapper (efficient_model_ax.py:359-388): pad zero frames, z [B, samples], drop the last hop samples."""
    mel = np.asarray(mel, dtype=F32)
    hop = cfg["hop_length"]
    melp = np.pad(mel, ((0, 0), (0, 0), (0, artifact_trimming))) if artifact_trimming > 0 else mel
    samples = (melp.shape[2] - 1) * hop
    samples -= samples % cfg["n_group"]
    assert z.shape == (mel.shape[0], samples), (z.shape, samples)
    audio = waveglow_ax_inverse(sd, cfg, z, melp, speaker_ids)
    return audio[:, :-artifact_trimming * hop] if artifact_trimming > 0 else audio
