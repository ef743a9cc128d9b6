"""CPU oracle for the WaveFlow path of the "ax" WaveGlow core (BASELINE config 4).

TEST INFRASTRUCTURE ONLY (see oracle/waveglow_oracle.py header for the import rule).
numpy fp32 restatement written from SURVEY.md §8(a) "Verified restatement of rows W1-W4".
Reference lines followed (relative to /root/reference/CookieTTS/_4_mtw/waveglow/):
  waveflow_inverse   efficient_model_ax.py:279-357 (z.view(B,-1,G).transpose :310, flow loop :325-340,
                     ignore_nan :333-334, un-squeeze :346)
  permutation        efficient_modules.py:360-403 (PermuteHeight: reverse / bipartite reverse)
  coupling           efficient_modules.py:42-65  (row 0 passes, (x - t) / exp(log_s))
  wn2d_row           glow_ax.py:556-635 (start :558, cond layer + linear interp :564-592,545-554,
                     row queue :597-602, Conv2d (k_h,k_w) dil (d_h,2^i) width pad only :518-523, GTU gate
                     :36-43, res/skip :615-626, end :628)
  waveflow_infer     efficient_model_ax.py:359-388 (pad one zero frame, samples, trim hop)

  model-level cond   efficient_model_ax.py:280-306 (shift/scale :280-283, speaker embedding concat :286-291, conv
                     stack + activation after EVERY layer :293-297, rezero alpha :299-300, residual :302-307)
  activation names   efficient_model_ax.py:100-111 = glow_ax.py:493-504: 'lrelu' selects F.relu and 'relu' selects
                     LeakyReLU(negative_slope) - the reference's own (swapped) mapping, restated as is
  WN-level cond      glow_ax.py:566-577 (WN speaker embedding concat, stack, activation after the last layer only if
                     cond_out_activation_func)
  separable in-layer glow_ax.py:525-531 (depthwise Conv2d groups=C, then pointwise 1x1 C->2C, both weight-normed)
  de-emphasis        efficient_model_ax.py:351-355 (scipy.signal.lfilter([1],[1,-p]) in float64, cast back)

Parity pin: tests/golden/waveflow_*.npz = outputs of the reference's own ``inverse(z, cond)`` /
``infer`` (tests/golden/make_golden.py).  Restrictions: waveflow=True, channel_mixing='permuteheight',
mix_first=False, res_skip=True, merge_res_skip=False, GTU gate, n_early_every > n_flows, no transposed-conv
upsampling, height dilation as configured.
"""
from __future__ import annotations

import numpy as np

from .waveglow_oracle import fold_weightnorm

F32 = np.float32


def _w(sd, prefix):
    if prefix + ".weight" in sd:
        return np.asarray(sd[prefix + ".weight"], dtype=F32)
    return fold_weightnorm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"])


def permutation(k, G):
    """Row permutation P_k (its own inverse): new[g] = old[perm[g]]."""
    idx = list(range(G))
    if k % 4 in (2, 3):
        half = G // 2
        return idx[:half][::-1] + idx[half:][::-1]
    return idx[::-1]


def lerp_align_corners(x, out_len):
    """F.interpolate(mode='linear', align_corners=True) on the last axis, fp32 like ATen's CPU kernel:
    scale = (in-1)/(out-1) in fp32, real = scale*dst, i0 = int(real), l1 = real - i0."""
    x = np.asarray(x, dtype=F32)
    n_in = x.shape[-1]
    if out_len == n_in:
        return x.copy()
    scale = F32(n_in - 1) / F32(out_len - 1) if out_len > 1 else F32(0)
    real = (scale * np.arange(out_len, dtype=F32)).astype(F32)
    i0 = real.astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (real - i0.astype(F32)).astype(F32)
    l0 = (F32(1.0) - l1).astype(F32)
    return (l0 * x[..., i0] + l1 * x[..., i1]).astype(F32)


def activation(name, negative_slope):
    """The reference's name -> function table (ax:100-111, gax:493-504), quirk included."""
    name = (name or 'none').lower()
    if name == 'none':
        return None
    if name == 'lrelu':
        return lambda x: np.maximum(x, F32(0))
    if name == 'relu':
        return lambda x: np.where(x >= 0, x, x * F32(negative_slope)).astype(F32)
    if name == 'tanh':
        return np.tanh
    if name == 'sigmoid':
        return lambda x: (F32(1) / (F32(1) + np.exp(-x))).astype(F32)
    raise NotImplementedError(name)


def gated_unit(name, u, C):
    """glow_ax.py:36-165: acts = f(u[:, :C]) * g(u[:, C:]) for the fourteen units of get_gate_func (:168-198)."""
    a, b = u[:, :C].astype(F32), u[:, C:].astype(F32)
    name = (name or 'GTU').upper()
    sig = lambda v: (F32(1.0) / (F32(1.0) + np.exp(-v))).astype(F32)
    lrelu = lambda v, s: np.where(v >= 0, v, v * F32(s)).astype(F32)
    first = {"GLU": lambda v: v, "GTSU": lambda v: v - np.tanh(v), "GTSRU": lambda v: v - np.tanh(v), "GSIU": np.sin,
             "GSIRU": lambda v: np.sin(F32(16.0) * v), "GSIRRU": lambda v: np.sin(F32(16.0) * v),
             "GSIRLRU": lambda v: np.sin(F32(16.0) * v), "GSIRRLRU": lambda v: np.sin(F32(16.0) * v)}.get(name, np.tanh)
    selu = lambda v: (F32(1.0507009873554805) * (np.maximum(v, 0) + np.minimum(0, F32(1.6732632423543772) * np.expm1(v)))).astype(F32)
    second = {"GTRU": lambda v: np.maximum(v, F32(0)), "GTSRU": lambda v: np.maximum(v, F32(0)),
              "GSIRRU": lambda v: np.maximum(v, F32(0)), "GTLRU": lambda v: lrelu(v, 0.01), "GSIRLRU": lambda v: lrelu(v, 0.01),
              "GSIRRLRU": lambda v: lrelu(v, (0.01 + 0.1) / 2),          # F.rrelu outside training: the mean slope
              "TTU": np.tanh, "STU": selu,
              "SPTU": lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, F32(20)))))}.get(name, sig)
    if name not in ("GTU", "GTRU", "GTLRU", "GLU", "TTU", "STU", "GTSU", "SPTU", "GSIU", "GSIRU", "GTSRU", "GSIRRU", "GSIRLRU",
                    "GSIRRLRU"):
        raise Exception("gated_unit is invalid")
    return (first(a).astype(F32) * second(b).astype(F32)).astype(F32)


def conv1d_same(x, w, b, padding_mode='zeros'):
    """Conv1d, odd kernel, padding (k-1)/2 with zeros or edge values ('replicate', the two modes the reference's
    configs use: efficient_model_ax.py:90, glow_ax.py:311): x [B, Cin, T], w [Cout, Cin, k] -> [B, Cout, T]."""
    k = w.shape[2]
    T = x.shape[2]
    h = k // 2
    if padding_mode == 'replicate':
        xp = np.pad(x, ((0, 0), (0, 0), (h, h)), mode='edge')
    else:
        assert padding_mode == 'zeros', padding_mode
        xp = np.pad(x, ((0, 0), (0, 0), (h, h)))
    y = np.zeros((x.shape[0], w.shape[0], T), F32) + b[None, :, None]
    for j in range(k):
        y = y + np.matmul(np.ascontiguousarray(w[:, :, j]), xp[:, :, j:j + T])
    return y.astype(F32)


def model_cond(sd, cfg, mel, speaker_ids=None):
    """efficient_model_ax.py:280-307: what every flow's WN receives as `cond` (frame rate)."""
    cond = np.asarray(mel, dtype=F32)
    if cfg.get("shift_spect", 0.) != 0.:
        cond = cond + F32(cfg["shift_spect"])
    if cfg.get("scale_spect", 1.) != 1.:
        cond = cond * F32(cfg["scale_spect"])
    if cfg["speaker_embed"]:
        emb = sd["speaker_embed.weight"][np.asarray(speaker_ids)]
        cond = np.concatenate([cond, np.repeat(emb[:, :, None], cond.shape[2], axis=2)], axis=1)
    if not cfg["cond_layers"]:
        return cond.astype(F32)                                   # empty stack: cond_res is cond itself
    act = activation(cfg.get("cond_activation_func", 'none'), cfg.get("negative_slope"))
    res = cond
    for l in range(cfg["cond_layers"]):
        res = conv1d_same(res, _w(sd, f"cond_layers.{l}"), sd[f"cond_layers.{l}.bias"], cfg.get("cond_padding_mode", 'zeros'))
        if act is not None:
            res = act(res).astype(F32)
    if "alpha" in sd:
        res = res * sd["alpha"][0]
    if not cfg["cond_residual"]:
        return res.astype(F32)
    if "res_conv.weight" in sd:                                   # cond_residual='1x1conv' (ax:80-81, 303-304)
        cond = (np.matmul(np.asarray(sd["res_conv.weight"], dtype=F32)[:, :, 0], cond) + sd["res_conv.bias"][None, :, None]).astype(F32)
    return (cond + res).astype(F32)


def to_latent_length(cond, L, interpolation_required, two_d):
    """The tail of ``_upsample_mels`` (ax:177-185 = glow_ax.py:365-372 for the 1-D WN; :548-553 for WN_2d): linear
    interpolation with align_corners=True when the transposed convs' factor is not hop // n_group and the lengths
    differ, else a centre crop - ``[pad_l : -pad_r]`` (1-D) or ``[pad : -(pad + pad % 2)]`` (2-D)."""
    n = cond.shape[2]
    if interpolation_required and n != L:
        return lerp_align_corners(cond, L)
    if two_d:
        pad = (n - L) // 2
        out = cond[:, :, pad:n - (pad + pad % 2)]
    else:
        pad_l, pad_r = (n - L) // 2, -((L - n) // 2)
        out = cond[:, :, pad_l:n - pad_r]
    assert out.shape[2] == L, f"crop of {n} columns to {L} fails in the reference as well"
    return np.ascontiguousarray(out)


def wn_upsample(sd, p, wn, spect, L, upsample_factor, two_d):
    """``if self.upsample_first is False: spect = self._upsample_mels(spect, audio.shape)`` (glow_ax.py:389-390 / 576-577):
    the WN's own TransposedUpsampleNet when it has one, then interpolation or crop to the latent's length."""
    required = True
    if f"{p}.upsample_net.t_convs.0.weight" in sd:
        scales = wn["transposed_conv_scales"]
        spect = transposed_upsample_net(sd, p + ".upsample_net", spect, scales, wn.get("transposed_conv_kernel_size", 4),
                                        False, False, False)
        required = int(np.prod(scales)) != upsample_factor
    return to_latent_length(spect, L, required, two_d)


def flow_conds(sd, cfg, cond):
    """efficient_model_ax.py:131-134, 320-321: the optional per-flow 1x1 (grouped) conv of the conditioning,
    ``n_flow_group_conv(cond).chunk(n_flows, dim=1)`` -> list of per-flow tensors; without it every flow gets `cond`."""
    n_flows = cfg["n_flows"]
    if "n_flow_group_conv.weight" not in sd:
        return [cond] * n_flows
    w = np.asarray(sd["n_flow_group_conv.weight"], dtype=F32)[:, :, 0]      # [out * n_flows, cin / groups]
    b = np.asarray(sd["n_flow_group_conv.bias"], dtype=F32)
    out = w.shape[0] // n_flows
    grouped = w.shape[1] != cond.shape[1]
    res = []
    for k in range(n_flows):
        x = cond[:, k * w.shape[1]:(k + 1) * w.shape[1]] if grouped else cond
        res.append((np.matmul(w[k * out:(k + 1) * out], x) + b[None, k * out:(k + 1) * out, None]).astype(F32))
    return res


def conv_transpose1d(x, w, b, stride, padding):
    """nn.ConvTranspose1d: x [B, Cin, T], w [Cin, Cout, k] -> [B, Cout, (T-1)*stride - 2*padding + k]:
    out[n] = b + sum_{t, kk: t*stride - padding + kk = n} w[:, :, kk]^T x[:, t]."""
    B, _, T = x.shape
    k = w.shape[2]
    full = np.zeros((B, w.shape[1], (T - 1) * stride + k), F32)
    for kk in range(k):
        full[:, :, kk:kk + (T - 1) * stride + 1:stride] += np.matmul(np.ascontiguousarray(w[:, :, kk].T), x)
    out = full[:, :, padding:full.shape[2] - padding] + b[None, :, None]
    return out.astype(F32)


def interp_scale(x, scale, linear):
    """F.interpolate(x, scale_factor=scale, mode='linear' (align_corners=False) | 'nearest') on the last axis, the
    arithmetic of ATen's kernels with a given scale factor: src = max((n + 0.5) / scale - 0.5, 0) | floor(n / scale)."""
    x = np.asarray(x, dtype=F32)
    n_in = x.shape[-1]
    n_out = int(np.floor(n_in * scale))
    n = np.arange(n_out, dtype=F32)
    inv = F32(1.0) / F32(scale)
    if not linear:
        return x[..., np.minimum(np.floor(n * inv).astype(np.int64), n_in - 1)].astype(F32)
    real = np.maximum(inv * (n + F32(0.5)) - F32(0.5), F32(0)).astype(F32)
    i0 = real.astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (real - i0.astype(F32)).astype(F32)
    return ((F32(1.0) - l1) * x[..., i0] + l1 * x[..., i1]).astype(F32)


def transposed_upsample_net(sd, prefix, x, scales, kernel_size, last_act, residual, residual_linear):
    """glow_ax.py:201-242 (TransposedUpsampleNet.forward): ConvTranspose1d(stride = scale, padding = (k - scale) // 2)
    + LeakyReLU(0.4) per scale; with `residual`, rezero weight on the result and the interpolated input added to the
    first min(in, out) channels."""
    x = np.asarray(x, dtype=F32)
    xi = interp_scale(x, int(np.prod(scales)), residual_linear) if residual else None
    idx = 0
    for i, sc in enumerate(scales):
        k = kernel_size[i] if isinstance(kernel_size, (list, tuple)) else kernel_size
        x = conv_transpose1d(x, np.asarray(sd[f"{prefix}.t_convs.{idx}.weight"], dtype=F32), sd[f"{prefix}.t_convs.{idx}.bias"],
                             sc, (k - sc) // 2)
        if i + 1 < len(scales) or last_act:
            x = np.where(x >= 0, x, x * F32(0.4)).astype(F32)
        idx += 2
    if residual:
        if f"{prefix}.res_weight" in sd:
            x = (x * sd[f"{prefix}.res_weight"][0]).astype(F32)
        r = min(xi.shape[1], x.shape[1])
        x[:, :r] = x[:, :r] + xi[:, :r]
    return x.astype(F32)


def deemphasis(x, p):
    """y[n] = x[n] + p*y[n-1] in float64 like scipy.signal.lfilter([1],[1,-p]) (ax:351-355)."""
    y = np.empty(x.shape, np.float64)
    acc = np.zeros(x.shape[0], np.float64)
    xd = x.astype(np.float64)
    for n in range(x.shape[1]):
        acc = xd[:, n] + float(p) * acc
        y[:, n] = acc
    return y.astype(F32)


def _shift(x, s):
    if s == 0:
        return x
    y = np.zeros_like(x)
    if s > 0:
        y[..., :-s] = x[..., s:]
    else:
        y[..., -s:] = x[..., :s]
    return y


def waveflow_inverse(sd, cfg, z, mel, speaker_ids=None):
    """z [B, T] (sigma applied), mel [B, n_mel, F'] (already padded by infer) -> audio [B, T]."""
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
    G, n_flows = cfg["n_group"], cfg["n_flows"]
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    kh, kw = wn["kernel_size_h"], wn["kernel_size_w"]
    dhs = wn["n_layers_dilations_h"]
    dhs = [dhs] * n_layers if isinstance(dhs, int) else list(dhs)
    z = np.asarray(z, dtype=F32)
    B, T = z.shape
    L = T // G
    a = np.ascontiguousarray(z.reshape(B, L, G).transpose(0, 2, 1))        # a[b, g, l] = z[b, G*l + g]
    frames = model_cond(sd, cfg, mel, speaker_ids)
    if cfg.get("upsample_first") is True:                                  # ax:318-319 -> _upsample_mels :174-186
        frames = transposed_upsample_net(sd, "upsample_net", frames, cfg["transposed_conv_scales"],
                                         cfg["transposed_conv_kernel_size"], True, cfg.get("transposed_conv_residual", False),
                                         cfg.get("transposed_conv_residual_linear", False))
        frames = to_latent_length(frames, L, int(np.prod(cfg["transposed_conv_scales"])) != cfg["hop_length"] // G, False)
    frames_k = flow_conds(sd, cfg, frames)
    wn_act = activation(wn.get("cond_activation_func", 'none'), wn.get("negative_slope"))
    sep = bool(wn.get("seperable_conv")) and not (kh == 1 and kw == 1)
    # early outputs (ax:311-313): the LAST split is the initial latent, earlier chunks re-join in front (ax:340-341)
    every, esize = cfg.get("n_early_every", 10 ** 9) or 10 ** 9, cfg.get("n_early_size", 0)
    n_early = sum(1 for k in range(1, n_flows) if k % every == 0)
    remained = [a[:, i * esize:(i + 1) * esize] for i in range(n_early)]
    a = a[:, n_early * esize:]
    name = cfg.get("channel_mixing", '1x1conv').lower()
    conv_mix = name in "1x1convinvertibleconv1x1invconv"                       # ax:25
    mix_first = cfg.get("mix_first", True)

    def unmix(k, v):                                                           # PermuteHeight / InvertibleConv1x1 inverse
        if not conv_mix:
            return v[:, permutation(k, v.shape[1]), :]
        w_inv = np.linalg.inv(sd[f"convinv.{k}.weight"][:, :, 0].astype(F32)).astype(F32)
        return np.matmul(w_inv, v).astype(F32)

    for k in reversed(range(n_flows)):
        p = f"WN.{k}.WN"
        G = a.shape[1]                                                         # active rows of this flow
        if not mix_first:
            a = unmix(k, a)                                                    # ax:324-325
        spect = frames_k[k]
        if wn.get("speaker_embed_dim", 0):
            emb = sd[p + ".speaker_embed.weight"][np.asarray(speaker_ids)]
            spect = np.concatenate([spect, np.repeat(emb[:, :, None], spect.shape[2], axis=2)], axis=1)
        for l in range(wn["cond_layers"]):
            spect = conv1d_same(spect, _w(sd, f"{p}.cond_layers.{l}"), sd[f"{p}.cond_layers.{l}.bias"],
                                wn.get("cond_padding_mode", 'zeros'))
            if wn_act is not None and (wn.get("cond_out_activation_func", True) or l != wn["cond_layers"] - 1):
                spect = wn_act(spect).astype(F32)
        # `if not self.upsample_first:` (glow_ax.py:576-577): the WN upsamples unless the model already did
        cond = spect if cfg.get("upsample_first") else wn_upsample(sd, p, wn, spect, L, cfg["hop_length"] // cfg["n_group"], True)
        ws = _w(sd, p + ".start").reshape(C)
        bs = sd[p + ".start.bias"]
        if sep:
            wdw = [_w(sd, f"{p}.in_layers.{i}.0")[:, 0] for i in range(n_layers)]            # [C, kh, kw]
            wpw = [_w(sd, f"{p}.in_layers.{i}.1")[:, :, 0, 0] for i in range(n_layers)]      # [2C, C]
        else:
            win = [_w(sd, f"{p}.in_layers.{i}") for i in range(n_layers)]      # [2C, C, kh, kw]
        has_rs = wn.get("res_skip", True)
        wrs = [_w(sd, f"{p}.res_skip_layers.{i}")[:, :, 0, 0] for i in range(n_layers)] if has_rs else None
        wend = sd[p + ".end.weight"][:, :, 0, 0]
        bend = sd[p + ".end.bias"]
        y = [a[:, 0, :]]
        queues = [np.zeros((B, C, (kh - 1) * dhs[i], L), dtype=F32) for i in range(n_layers)]
        for r in range(G - 1):
            x = ws[None, :, None] * y[r][:, None, :] + bs[None, :, None]       # [B, C, L]
            out = None
            for i in range(n_layers):
                dlw = wn.get("n_layers_dilations_w")                          # glow_ax.py:507-509
                dw, dh = (2 ** i if dlw is None else (dlw if isinstance(dlw, int) else dlw[i])), dhs[i]
                pad = ((kw - 1) * dw) // 2
                Q = np.concatenate([queues[i], x[:, :, None, :]], axis=2)      # [B, C, (kh-1)dh+1, L]
                queues[i] = Q[:, :, 1:, :] if (kh - 1) * dh > 0 else queues[i]
                if sep:
                    d = np.zeros((B, C, L), F32) + sd[f"{p}.in_layers.{i}.0.bias"][None, :, None]
                    for ah in range(kh):
                        row = Q[:, :, ah * dh, :]
                        for j in range(kw):
                            d = d + wdw[i][None, :, ah, j, None] * _shift(row, j * dw - pad)
                    u = (np.matmul(wpw[i], d.astype(F32)) + sd[f"{p}.in_layers.{i}.1.bias"][None, :, None]
                         + cond[:, 2 * C * i:2 * C * (i + 1), :])
                else:
                    u = (sd[f"{p}.in_layers.{i}.bias"][None, :, None] + cond[:, 2 * C * i:2 * C * (i + 1), :]).astype(F32)
                    for ah in range(kh):
                        row = Q[:, :, ah * dh, :]
                        for j in range(kw):
                            u = u + np.matmul(np.ascontiguousarray(win[i][:, :, ah, j]), _shift(row, j * dw - pad))
                u = u.astype(F32)
                act = gated_unit(wn.get("gated_unit", 'GTU'), u, C)
                rs = (np.matmul(wrs[i], act) + sd[f"{p}.res_skip_layers.{i}.bias"][None, :, None]) if has_rs else act   # :609
                if i < n_layers - 1 and not wn.get("merge_res_skip", False) and has_rs:     # glow_ax.py:612-626
                    x = x + rs[:, :C]
                    out = rs[:, C:] if out is None else out + rs[:, C:]
                else:
                    out = rs if out is None else out + rs
            e = np.matmul(wend, out) + bend[None, :, None]                     # [B, 2, L]
            log_s, t = e[:, 0, :], e[:, 1, :]
            with np.errstate(over="ignore", invalid="ignore"):
                y.append(((a[:, r + 1, :] - t) / np.exp(log_s)).astype(F32))
        a = np.stack(y, axis=1)
        a = np.where(np.isnan(a), F32(0), a).astype(F32)
        if mix_first:
            a = unmix(k, a)                                                    # ax:337-338
        if k % every == 0 and k:
            a = np.concatenate([remained.pop(), a], axis=1)
    assert not remained
    audio = np.ascontiguousarray(a.transpose(0, 2, 1)).reshape(B, T)
    if cfg.get("preceived_vol_scaling"):                                   # ax:342-344
        with np.errstate(divide="ignore", invalid="ignore"):
            pos = np.power(F32(10.0), np.log2(np.where(audio > 0, audio, F32(1)))).astype(F32)
            neg = -np.power(F32(10.0), np.log2(np.where(audio < 0, -audio, F32(1)))).astype(F32)
        audio = np.where(audio > 0, pos, np.where(audio < 0, neg, audio)).astype(F32)
    if cfg.get("preempthasis"):
        audio = deemphasis(audio, cfg["preempthasis"])
    return audio


def waveflow_infer(sd, cfg, mel, z, artifact_trimming=1, speaker_ids=None):
    """infer() wrapper: pad `artifact_trimming` zero frames, z [B, F*hop], drop the last hop samples."""
    mel = np.asarray(mel, dtype=F32)
    hop = cfg["hop_length"]
    melp = np.pad(mel, ((0, 0), (0, 0), (0, artifact_trimming))) if artifact_trimming > 0 else mel
    samples = (melp.shape[2] - 1) * hop
    samples -= samples % cfg["n_group"]
    assert z.shape == (mel.shape[0], samples), (z.shape, samples)
    audio = waveflow_inverse(sd, cfg, z, melp, speaker_ids)
    return audio[:, :-artifact_trimming * hop] if artifact_trimming > 0 else audio
