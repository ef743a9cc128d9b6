"""CPU oracle for the WaveGlow mel->wave path (NVIDIA-style ``glow.py`` topology).

TEST INFRASTRUCTURE ONLY.  This is a numpy fp32 restatement of the reference's
algorithm, written from the equations in SURVEY.md §8(a) "Verified restatement of
rows G2-G8", not from the reference's code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product path (``cookietts_amd``) never does and fails loudly without its HIP
library.

Parity pin: ``tests/golden/waveglow_*.npz`` hold outputs of the reference itself
(``/root/reference/CookieTTS/_4_mtw/waveglow/glow.py`` ``WaveGlow.infer``, imported
and run on CPU by ``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py``
checks this file against them.

Reference lines each function follows (paths relative to /root/reference/CookieTTS):
  fold_weightnorm   torch ``nn.utils.weight_norm`` as applied at _4_mtw/waveglow/glow.py:135-186
  upsample_squeeze  glow.py:318-324 (ConvTranspose1d, trim ``win-hop``, unfold to groups); grouped weights for
                    upsample_mode 'simple' / 'simple_half' (glow.py:238-241: groups = n_mel / n_mel/2)
  wn_forward        glow.py:188-222 (+ fused gate glow.py:34-41); speaker-embedding concat :193-196; ReZero
                    ``res_skip(acts) * alpha_i`` :211-212
  waveglow_infer    glow.py:314-350 (coupling inverse :337-338, inverse 1x1 conv :85-99,
                    early-output re-injection :342-347, un-squeeze :349)
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def fold_weightnorm(g, v):
    """w = v * (g / ||v||), norm over every dim except 0 (per output channel)."""
    v = np.asarray(v, dtype=F32)
    g = np.asarray(g, dtype=F32)
    norm = np.sqrt(np.sum(v * v, axis=tuple(range(1, v.ndim)), keepdims=True, dtype=F32))
    return (v * (g.reshape(norm.shape) / norm)).astype(F32)


def _conv_weight(sd, prefix):
    """Effective conv weight [O, I, K] for a layer stored weight-normed or plain."""
    if prefix + ".weight" in sd:
        return np.asarray(sd[prefix + ".weight"], dtype=F32)
    return fold_weightnorm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"])


def fold_state_dict(sd):
    """Fold every weight_g/weight_v pair once (what ``remove_weightnorm`` would leave behind)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_v"):
            out[k[:-2]] = fold_weightnorm(sd[k[:-2] + "_g"], v)
        elif not k.endswith(".weight_g"):
            out[k] = np.asarray(v, dtype=F32)
    return out


def _conv1x1(w, b, x):
    """w [O, I, 1], b [O], x [B, I, L] -> [B, O, L]."""
    return np.matmul(np.ascontiguousarray(w[:, :, 0]), x) + b[None, :, None]


def _shift(x, s):
    """y[..., l] = x[..., l + s], zero outside [0, L)."""
    if s == 0:
        return x
    y = np.zeros_like(x)
    if s > 0:
        y[..., :-s] = x[..., s:]
    else:
        y[..., -s:] = x[..., :s]
    return y


def upsample_squeeze(mel, w_up, b_up, hop, n_group):
    """mel [B, M, F] -> spect [B, M*n_group, F*hop/n_group].

    y[b,o,t] = bias[o] + sum_{i,f} mel[b,i,f] * w_up[i,o,t - f*hop]   for 0 <= t - f*hop < win
    keep t < F*hop;  spect[b, o*G + g, l] = y[b, o, G*l + g].
    """
    mel = np.asarray(mel, dtype=F32)
    B, M, F = mel.shape
    win = w_up.shape[2]
    taps = win // hop
    assert taps * hop == win
    opg = w_up.shape[1]                                         # output channels per group (ConvTranspose1d weight
    n_out = b_up.shape[0]                                       # is [in, out/groups, win])
    groups = n_out // opg
    ipg = M // groups
    y = np.zeros((B, n_out, F + taps - 1, hop), dtype=F32)
    for gi in range(groups):
        xt = np.ascontiguousarray(mel[:, gi * ipg:(gi + 1) * ipg].transpose(0, 2, 1))     # [B, F, ipg]
        wg = w_up[gi * ipg:(gi + 1) * ipg]
        for j in range(taps):
            wj = wg[:, :, j * hop:(j + 1) * hop].reshape(ipg, -1)                         # [ipg, opg*hop]
            contrib = np.matmul(xt, wj).reshape(B, F, opg, hop)
            y[:, gi * opg:(gi + 1) * opg, j:j + F, :] += contrib.transpose(0, 2, 1, 3)
    y = y[:, :, :F, :].reshape(B, n_out, F * hop) + b_up[None, :, None]
    L = F * hop // n_group
    y = y.reshape(B, y.shape[1], L, n_group)                    # [B, O, L, G]
    return np.ascontiguousarray(y.transpose(0, 1, 3, 2)).reshape(B, -1, L)


def _with_speaker(sd, prefix, spect, speaker_ids):
    """glow.py:193-196: the flow's speaker embedding, repeated over time, concatenated under the spectrogram."""
    if speaker_ids is None or prefix + ".speaker_embed.weight" not in sd:
        return spect
    emb = np.asarray(sd[prefix + ".speaker_embed.weight"], dtype=F32)[np.asarray(speaker_ids)]
    return np.concatenate([spect, np.repeat(emb[:, :, None], spect.shape[2], axis=2)], axis=1)


def wn_forward(sd, prefix, audio0, spect, n_layers, n_channels, trace=None, speaker_ids=None):
    """One WN stack: returns (b, log_s), each [B, n_half, L]."""
    C = n_channels
    x = _conv1x1(_conv_weight(sd, prefix + ".start"), sd[prefix + ".start.bias"], audio0)
    c = _with_speaker(sd, prefix, spect, speaker_ids)
    j = 0
    while f"{prefix}.cond_layers.{j}.bias" in sd:               # no nonlinearity between layers
        c = _conv1x1(_conv_weight(sd, f"{prefix}.cond_layers.{j}"),
                     sd[f"{prefix}.cond_layers.{j}.bias"], c)
        j += 1
    out = np.zeros_like(x)
    for i in range(n_layers):
        d = 2 ** i
        w = _conv_weight(sd, f"{prefix}.in_layers.{i}")          # [2C, C, 3]
        u = sd[f"{prefix}.in_layers.{i}.bias"][None, :, None] + c[:, 2 * C * i:2 * C * (i + 1), :]
        ks = w.shape[2]
        for t in range(ks):
            u = u + np.matmul(np.ascontiguousarray(w[:, :, t]), _shift(x, (t - ks // 2) * d))
        u = u.astype(F32)
        act = np.tanh(u[:, :C]) * (F32(1.0) / (F32(1.0) + np.exp(-u[:, C:])))
        r = _conv1x1(_conv_weight(sd, f"{prefix}.res_skip_layers.{i}"),
                     sd[f"{prefix}.res_skip_layers.{i}.bias"], act.astype(F32))
        if f"{prefix}.alpha_i.{i}" in sd:                        # ReZero (glow.py:211-212)
            r = (r * sd[f"{prefix}.alpha_i.{i}"][0]).astype(F32)
        if i < n_layers - 1:
            x = x + r[:, :C]
            out = out + r[:, C:]
        else:
            out = out + r
        if trace is not None:
            trace.append((x.copy(), out.copy()))
    e = _conv1x1(np.asarray(sd[prefix + ".end.weight"], dtype=F32), sd[prefix + ".end.bias"], out)
    h = e.shape[1] // 2
    return e[:, :h], e[:, h:]


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (the storage rounding of the bf16 HIP variant)."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(F32)


def f16_round(x):
    """Round-to-nearest-even fp32 -> IEEE half -> fp32 (the storage rounding of the f16 HIP variant; numpy's cast rounds to
    nearest even, keeps subnormals and overflows to inf like v_cvt_f16_f32)."""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


BF16_SKIP_GROUP = 4     # layers per skip-sum launch of the bf16 HIP variant (BF_SKIP_GROUP in waveglow_api.hip)


def wn_forward_bf16(sd, prefix, audio0, spect, n_layers, n_channels, speaker_ids=None, rnd=None):
    """bf16-rounded restatement of one WN stack, mirroring the rounding points of the bf16 HIP variant
    (BASELINE config 3): in-layer / cond-layer-2 / res-skip weights and the tensors x, h, act, skip-sum are
    rounded to bf16 where the kernels store them (incl. the squeezed spectrogram and cond layers 0-1); all sums
    are fp32; upsampling, start and end are fp32.  ``rnd``: the storage rounding - ``bf16_round`` (default) or ``f16_round``
    (the IEEE-half variant: the same rounding points, 11-bit significands)."""
    rnd = bf16_round if rnd is None else rnd
    C = n_channels
    x = rnd(_conv1x1(_conv_weight(sd, prefix + ".start"), sd[prefix + ".start.bias"], audio0))
    h = rnd(_with_speaker(sd, prefix, spect, speaker_ids))
    for j in range(2):
        h = rnd(_conv1x1(rnd(_conv_weight(sd, f"{prefix}.cond_layers.{j}")),
                                sd[f"{prefix}.cond_layers.{j}.bias"], h))
    wc2 = rnd(_conv_weight(sd, f"{prefix}.cond_layers.2")[:, :, 0])
    bc2 = sd[f"{prefix}.cond_layers.2.bias"]
    out = None
    skip, bias_total, pending = None, None, []   # fp32 partial skip sum of the current group of BF16_SKIP_GROUP layers
    for i in range(n_layers):
        d = 2 ** i
        w = rnd(_conv_weight(sd, f"{prefix}.in_layers.{i}"))
        bias = (sd[f"{prefix}.in_layers.{i}.bias"] + bc2[2 * C * i:2 * C * (i + 1)]).astype(F32)
        u = np.matmul(np.ascontiguousarray(wc2[2 * C * i:2 * C * (i + 1)]), h)
        ks = w.shape[2]
        for t in range(ks):
            u = u + np.matmul(np.ascontiguousarray(w[:, :, t]), _shift(x, (t - ks // 2) * d))
        u = (u + bias[None, :, None]).astype(F32)
        act = rnd(np.tanh(u[:, :C]) * (F32(1.0) / (F32(1.0) + np.exp(-u[:, C:]))))
        wrs = _conv_weight(sd, f"{prefix}.res_skip_layers.{i}")
        brs = sd[f"{prefix}.res_skip_layers.{i}.bias"]
        if f"{prefix}.alpha_i.{i}" in sd:                        # the kernels fold alpha into weight and bias at pack time
            wrs, brs = (wrs * sd[f"{prefix}.alpha_i.{i}"][0]).astype(F32), (brs * sd[f"{prefix}.alpha_i.{i}"][0]).astype(F32)
        wq = rnd(wrs)
        zero = np.zeros(C, dtype=F32)
        if i < n_layers - 1:
            x = rnd(x + _conv1x1(wq[:C], brs[:C], act).astype(F32))
            wsk, bsk = wq[C:], brs[C:]
        else:
            wsk, bsk = wq, brs
        # the skip rows are one deferred contraction over the stored activations of BF16_SKIP_GROUP layers per launch:
        # fp32 sums inside a launch, the running sum rounded to bf16 once per launch, the summed skip biases of ALL
        # layers added by the first launch
        s_i = _conv1x1(wsk, zero, act).astype(F32)
        skip = s_i if skip is None else skip + s_i
        bias_total = np.asarray(bsk, dtype=F32) if bias_total is None else bias_total + np.asarray(bsk, dtype=F32)
        if (i + 1) % BF16_SKIP_GROUP == 0 or i == n_layers - 1:
            pending.append(skip)
            skip = None
    for gi, part in enumerate(pending):
        out = rnd(part + bias_total[None, :, None]) if gi == 0 else rnd(out + part)
    e = _conv1x1(np.asarray(sd[prefix + ".end.weight"], dtype=F32), sd[prefix + ".end.bias"], out)
    hh = e.shape[1] // 2
    return e[:, :hh], e[:, hh:]


def waveglow_infer(sd, cfg, mel, z_scaled, flow_trace=None, bf16=False, speaker_ids=None, f16=False):
    """mel [B, n_mel, F], z_scaled [B, n_group, L] (sigma already applied) -> wave [B, F*hop].

    ``z_scaled`` rows: the last ``n_remaining_channels`` are the initial latent; the
    ``n_early_size`` rows above are prepended after flow k = n_early_every*(m-1), etc.
    (see cookietts_amd.synthetic.synthetic_noise).
    """
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
    G = cfg["n_group"]
    wn = cfg["WN_config"]
    n_flows, every, esize = cfg["n_flows"], cfg["n_early_every"], cfg["n_early_size"]
    spect = upsample_squeeze(mel, sd["upsample.weight"], sd["upsample.bias"], cfg["hop_length"], G)
    B, _, L = spect.shape
    z_scaled = np.asarray(z_scaled, dtype=F32)
    assert z_scaled.shape == (B, G, L), (z_scaled.shape, (B, G, L))
    n_early = sum(1 for k in range(1, n_flows) if k % every == 0)
    lo = n_early * esize
    audio = z_scaled[:, lo:, :]
    for k in reversed(range(n_flows)):
        h = audio.shape[1] // 2
        a0, a1 = audio[:, :h], audio[:, h:]
        if bf16 or f16:                      # reduced-precision variants: same rounding points, bf16 or IEEE-half storage
            b, s = wn_forward_bf16(sd, f"WN.{k}", a0, spect, wn["n_layers"], wn["n_channels"], speaker_ids=speaker_ids,
                                   rnd=f16_round if f16 else bf16_round)
        else:
            b, s = wn_forward(sd, f"WN.{k}", a0, spect, wn["n_layers"], wn["n_channels"], speaker_ids=speaker_ids)
        a1 = ((a1 - b) / np.exp(s)).astype(F32)
        audio = np.concatenate([a0, a1], axis=1)
        w = sd[f"convinv.{k}.conv.weight"][:, :, 0]
        w_inv = np.linalg.inv(w.astype(F32)).astype(F32)
        audio = np.matmul(w_inv, audio).astype(F32)
        if k % every == 0 and k > 0:
            lo -= esize
            audio = np.concatenate([z_scaled[:, lo:lo + esize, :], audio], axis=1)
        if flow_trace is not None:
            flow_trace.append(audio.copy())
    assert lo == 0
    return np.ascontiguousarray(audio.transpose(0, 2, 1)).reshape(B, -1)
