"""WaveFlow / "ax" WaveGlow core on the MI355X HIP path (BASELINE config 4, the ``waveflow=False`` WaveGlow the
reference's own timing notebook runs, and every other config the reference tree prints).

Host-side mirror of ``/root/reference/CookieTTS/_4_mtw/waveglow/efficient_model_ax.py``
``WaveGlow`` (:18-169 constructor, :279-357 ``inverse``, :359-388 ``infer``):

* ``waveflow=True``: ``WaveFlowCoupling`` + ``WN_2d`` (efficient_modules.py:19-65, glow_ax.py:421-635) ->
  ``ctts_waveflow_inverse_f32`` / ``ctts_waveflow_inverse_cond_f32``;
* ``waveflow=False``: ``AffineCouplingBlock`` + the 1-D ``WN`` (efficient_modules.py:68-105, glow_ax.py:245-418) ->
  ``ctts_wgax_inverse_f32``;
* on both: ``PermuteHeight`` or ``InvertibleConv1x1`` mixing in either ``mix_first`` order, early outputs, all
  fourteen gated units, ``merge_res_skip`` / ``res_skip=False``, per-layer dilations, speaker embeddings at model and
  WN level, the model-level conditioning stack (plain / residual / 1x1-conv residual, rezero), multi-layer WN
  conditioning stacks with activations, the grouped per-flow cond conv, model- and WN-level
  ``TransposedUpsampleNet``, spect shift / scale, log-variance mel channels, perceived-volume companding,
  de-emphasis; separable (depthwise + pointwise) in-layers on the 2-D core.

Same constructor kwargs, same ``state_dict`` keys (``WN.k.WN.{start,cond_layers.l,in_layers.i,res_skip_layers.i}.
{weight_g,weight_v,bias}``, ``WN.k.WN.end.{weight,bias}``, ``convinv.k.weight``, ``upsample_net.*``, ...), same
``infer`` / ``inverse`` contracts (output length ``(F-1)*hop`` with the default ``artifact_trimming=1``;
``return_CPU=True`` moves the result to the host like the reference).  The few options left raise
NotImplementedError.  The conditioning stacks are composed on the host from operator-level C entry points
(``ctts_conv1d_f32``, ``ctts_embed_rows_f32``, ``ctts_scale_add_rows_f32``, ``ctts_resample_rows_f32``,
``ctts_interleave_phases_f32``, ...); all arithmetic runs in the C-ABI HIP library; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cache, _lib

__all__ = ["WaveGlow"]


class _WNConv(nn.Module):
    """weight-normed conv parameters (weight_g, weight_v, bias) of arbitrary kernel rank."""

    def __init__(self, shape):
        super().__init__()
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        v = torch.empty(*shape)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fan_in)
        self.bias = nn.Parameter(torch.empty(shape[0]).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(shape[0], *([1] * (len(shape) - 1))).clone())
        self.weight_v = nn.Parameter(v)

    def remove_weight_norm(self):
        if getattr(self, "weight_v", None) is not None:
            v, g = self.weight_v.data, self.weight_g.data
            w = v * (g / v.flatten(1).norm(dim=1).view(g.shape))
            del self._parameters["weight_g"], self._parameters["weight_v"]
            self.weight = nn.Parameter(w)


class _WN2d(nn.Module):
    """Parameter tree of glow_ax.WN_2d (:427-543)."""

    def __init__(self, cond_in, wn):
        super().__init__()
        C_, n_layers = wn['n_channels'], wn['n_layers']
        kh, kw = wn['kernel_size_h'], wn['kernel_size_w']
        self.n_layers, self.n_channels = n_layers, C_
        sdim = wn.get('speaker_embed_dim', 0)
        self.start = _WNConv((C_, 1, 1, 1))
        self.end = nn.Module()
        self.end.weight = nn.Parameter(torch.zeros(2, C_, 1, 1))                # zero-init, glow_ax.py:454-457
        self.end.bias = nn.Parameter(torch.zeros(2))
        if sdim:
            self.speaker_embed = nn.Embedding(512, sdim)                        # glow_ax.py:459-461
        k = 2 * wn.get('cond_kernel_size', 1) - 1                               # glow_ax.py:474
        cond_out = 2 * C_ * n_layers
        if wn.get('transposed_conv_scales') and wn.get('transposed_conv_hidden_dim', 256) and wn.get('transposed_conv_kernel_size', 4):
            # the WN's own TransposedUpsampleNet behind its cond stack (upsample_first is False): the stack ends at the
            # net's hidden width, the net ends at 2C*n_layers (glow_ax.py:286-295 / 462-470)
            hid = wn.get('transposed_conv_hidden_dim', 256)
            self.upsample_net = _TransposedUpsampleNet(hid, cond_out, hid, wn.get('transposed_conv_kernel_size', 4),
                                                       wn['transposed_conv_scales'], False, False, False, False)
            cond_out = hid
        dims = [cond_in + sdim] + [wn['cond_hidden_channels']] * (wn['cond_layers'] - 1) + [cond_out]
        self.cond_layers = nn.ModuleList([_WNConv((dims[l + 1], dims[l], k)) for l in range(wn['cond_layers'])])
        if wn.get('seperable_conv', False) and not (kh == 1 and kw == 1):       # glow_ax.py:521-531
            self.in_layers = nn.ModuleList([nn.ModuleList([_WNConv((C_, 1, kh, kw)), _WNConv((2 * C_, C_, 1, 1))])
                                            for _ in range(n_layers)])
        else:
            self.in_layers = nn.ModuleList([_WNConv((2 * C_, C_, kh, kw)) for _ in range(n_layers)])
        merge = bool(wn.get('merge_res_skip', False))                            # glow_ax.py:535-538
        self.res_skip_layers = nn.ModuleList([
            _WNConv((2 * C_ if (i < n_layers - 1 and not merge) else C_, C_, 1, 1)) for i in range(n_layers)]
            if wn.get('res_skip', True) else [])                                 # res_skip=False: acts are the skip (:609)


class _WN1d(nn.Module):
    """Parameter tree of the 1-D glow_ax.WN (:251-360)."""

    def __init__(self, n_in, cond_in, wn):
        super().__init__()
        C_, n_layers = wn['n_channels'], wn['n_layers']
        ks = wn.get('kernel_size_w') or wn.get('kernel_size')                    # glow_ax.py:256
        assert ks % 2 == 1 and C_ % 2 == 0
        self.n_layers, self.n_channels = n_layers, C_
        sdim = wn.get('speaker_embed_dim', 0)
        self.start = _WNConv((C_, n_in, 1))
        self.end = nn.Module()
        self.end.weight = nn.Parameter(torch.zeros(2 * n_in, C_, 1))            # zero-init, glow_ax.py:278-281
        self.end.bias = nn.Parameter(torch.zeros(2 * n_in))
        if sdim:
            self.speaker_embed = nn.Embedding(512, sdim)                        # glow_ax.py:283-285
        k = 2 * wn.get('cond_kernel_size', 1) - 1                               # glow_ax.py:299
        cond_out = 2 * C_ * n_layers
        if wn.get('transposed_conv_scales') and wn.get('transposed_conv_hidden_dim', 256) and wn.get('transposed_conv_kernel_size', 4):
            # the WN's own TransposedUpsampleNet behind its cond stack (upsample_first is False): the stack ends at the
            # net's hidden width, the net ends at 2C*n_layers (glow_ax.py:286-295 / 462-470)
            hid = wn.get('transposed_conv_hidden_dim', 256)
            self.upsample_net = _TransposedUpsampleNet(hid, cond_out, hid, wn.get('transposed_conv_kernel_size', 4),
                                                       wn['transposed_conv_scales'], False, False, False, False)
            cond_out = hid
        dims = [cond_in + sdim] + [wn['cond_hidden_channels']] * (wn['cond_layers'] - 1) + [cond_out]
        self.cond_layers = nn.ModuleList([_WNConv((dims[l + 1], dims[l], k)) for l in range(wn['cond_layers'])])
        self.in_layers = nn.ModuleList([_WNConv((2 * C_, C_, ks)) for _ in range(n_layers)])
        merge = bool(wn.get('merge_res_skip', False))                            # glow_ax.py:352-355
        self.res_skip_layers = nn.ModuleList([
            _WNConv((2 * C_ if (i < n_layers - 1 and not merge) else C_, C_, 1)) for i in range(n_layers)]
            if wn.get('res_skip', True) else [])                                 # res_skip=False: acts are the skip (:401)


class InvertibleConv1x1(nn.Module):
    """Parameter holder of efficient_modules.InvertibleConv1x1 (:235-253): ``weight`` [c, c, 1], random orthonormal
    with det > 0.  ``W_inverse`` is cached on the module after the first inference like the reference (:271-276)."""

    def __init__(self, c):
        super().__init__()
        W = torch.linalg.qr(torch.randn(c, c))[0]
        if torch.det(W) < 0:
            W[:, 0] = -1 * W[:, 0]
        self.weight = nn.Parameter(W.view(c, c, 1).contiguous())


class _Coupling(nn.Module):
    def __init__(self, wn):
        super().__init__()
        self.WN = wn


PAD = 8          # halo of the frame-rate padded rows (>= k/2 of every cond conv, k <= 11)


def _act_code(name, negative_slope):
    """The reference's activation table (ax:100-111 = glow_ax.py:493-504) -> (ctts_conv1d act, slope).
    As written there, 'lrelu' selects ``F.relu`` and 'relu' selects ``LeakyReLU(negative_slope)``."""
    name = (name or 'none').lower()
    if name == 'none':
        return 0, 0.0
    if name == 'lrelu':
        return 1, 0.0
    if name == 'relu':
        assert negative_slope, "negative_slope not defined in wn_config"
        return 1, float(negative_slope)
    if name == 'tanh':
        return 2, 0.0
    if name == 'sigmoid':
        return 3, 0.0          # composed in _CondConv: sigmoid(x) = (tanh(x / 2) + 1) / 2
    raise NotImplementedError(name)                                            # ax:111


class _CondConv:
    """One ``ctts_conv1d`` operator of a conditioning stack: dense [out, in, k] weight, input channels zero-padded
    to the primitive's multiple of 16."""

    batched = True      # one launch for the whole batch when the buffers allow it (tests turn it off to compare)

    def __init__(self, w, b, act, slope, device, stream, gemm_mode=0):
        lib = _lib.lib()
        out_c, in_c, k = w.shape
        self.c_in, self.c_out = -(-in_c // 16) * 16, out_c
        self.sigmoid = act == 3
        wp = torch.zeros(out_c, self.c_in, k, dtype=torch.float32, device=device)
        wp[:, :in_c] = w
        b = b.detach().float().contiguous()
        if self.sigmoid:       # tanh epilogue on the halved pre-activation, then (t + 1) / 2 with ctts_affine_rows_f32
            wp, b, act = wp * 0.5, b * 0.5, 2
        self.desc = _lib.Conv1dDesc(c_in=self.c_in, c_out=out_c, kernel_size=k, act=act, slope=slope,
                                    f32_gemm_mode=gemm_mode)
        nbytes = lib.ctts_conv1d_packed_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported cond conv: " + lib.ctts_last_error().decode())
        self.blob = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
        _lib.check(lib.ctts_conv1d_pack_f32(C.byref(self.desc), _lib.ptr(wp), _lib.ptr(b), None, None, None, None, 1e-5,
                                           _lib.ptr(self.blob), stream), "ctts_conv1d_pack_f32")
        self._keep = (wp, b)

    def __call__(self, x, y, B, T, ld, stream, padding_mode='zeros'):
        # x [B][>= c_in rows][ld], y [B][>= c_out rows][ld]: the row counts of the buffers are rounded up to 16, so the
        # primitive (dense [batch][c][ld] strides) is driven one utterance at a time
        k = self.desc.kernel_size
        if padding_mode == 'replicate' and k > 1:      # edge values into the halo the taps reach (nn.Conv1d padding_mode)
            _lib.check(_lib.lib().ctts_replicate_halo_f32(_lib.ptr(x), B, x.shape[1], T, ld, PAD, k // 2, stream),
                       "ctts_replicate_halo_f32")
        if (self.batched and not self.sigmoid and x.shape[1] == self.c_in and y.shape[1] == self.c_out
                and tuple(x.stride()) == (self.c_in * ld, ld, 1) and tuple(y.stride()) == (self.c_out * ld, ld, 1)):
            # both buffers have exactly the operator's row counts: the dense [batch][c][ld] strides of the primitive hold
            # for the whole batch, one launch (B x the workgroups) instead of B
            _lib.check(_lib.lib().ctts_conv1d_f32(C.byref(self.desc), _lib.ptr(self.blob), _lib.ptr(x), _lib.ptr(y),
                                                 0, B, T, ld, PAD, stream), "ctts_conv1d_f32")
            return
        for b in range(B):
            _lib.check(_lib.lib().ctts_conv1d_f32(C.byref(self.desc), _lib.ptr(self.blob), _lib.ptr(x[b]), _lib.ptr(y[b]),
                                                 0, 1, T, ld, PAD, stream), "ctts_conv1d_f32")
            if self.sigmoid:
                _lib.check(_lib.lib().ctts_affine_rows_f32(_lib.ptr(y[b]), 1, y.shape[1], self.c_out, T, ld, PAD, 1.0, 0.5,
                                                           stream), "ctts_affine_rows_f32")


def _r16(n):
    return -(-n // 16) * 16


def _ld_for(T):
    """Row stride of a padded [B][rows][ld] buffer with T valid columns (a few spare columns for the transposed-conv
    phases, which run a little past the input length)."""
    return -(-(T + 8) // 128) * 128 + 2 * PAD


class _TransposedConv:
    """``nn.ConvTranspose1d(in, out, k, stride=s, padding=(k - s) // 2)`` (+ LeakyReLU) of ``TransposedUpsampleNet``
    (glow_ax.py:222-226) as ``s`` stride-1 convolutions, one per output residue r = (n + p) mod s:
    out[n] = sum_q W[:, :, r + q s]^T x[(n + p) // s - q], run by ``ctts_conv1d_f32`` over the input positions and
    interleaved by ``ctts_interleave_phases_f32``."""

    def __init__(self, w, b, s, act, slope, device, stream, gemm_mode=0):
        in_c, out_c, k = w.shape                                   # ConvTranspose1d weight layout
        if (k - s) % 2 or k < s:
            raise NotImplementedError(f"transposed conv with kernel {k}, stride {s}: output length is not stride * T")
        Q = -(-k // s)
        wr = torch.zeros(s, out_c, in_c, 2 * Q - 1, dtype=torch.float32, device=device)
        for r in range(s):
            for q in range(Q):
                if r + q * s < k:
                    wr[r, :, :, Q - 1 - q] = w[:, :, r + q * s].t()   # tap offset -q of a 'same' conv with 2Q-1 taps
        self.ops = [_CondConv(wr[r], b, act, slope, device, stream, gemm_mode) for r in range(s)]
        self.s, self.p, self.out_c = s, (k - s) // 2, out_c
        self.extra = (k - self.p - 1) // s                          # phase positions needed past the input length

    def __call__(self, x, B, T, ld, stream):
        dev = x.device
        Tp, T_out = T + self.extra, T * self.s
        assert ld >= PAD + Tp + PAD
        ph = torch.zeros(self.s, B, _r16(self.out_c), ld, dtype=torch.float32, device=dev)
        for r, op in enumerate(self.ops):
            op(x, ph[r], B, Tp, ld, stream)
        ld_out = _ld_for(T_out)
        y = torch.zeros(B, _r16(self.out_c), ld_out, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().ctts_interleave_phases_f32(_lib.ptr(ph), _lib.ptr(y), B, y.shape[1], self.s, self.p, Tp, ld,
                                                         PAD, T_out, ld_out, PAD, stream), "ctts_interleave_phases_f32")
        return y, T_out, ld_out


class _TransposedUpsampleNet(nn.Module):
    """Parameter holder with the reference's ``state_dict`` layout (glow_ax.py:201-242): ``t_convs`` alternates
    ConvTranspose1d and LeakyReLU(0.4) modules, ``res_weight`` exists with rezero."""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, scales, use_last_layer_act_func, residual,
                 residual_linear, rezero):
        super().__init__()
        self.residual, self.residual_linear = bool(residual), bool(residual_linear)
        self.res_weight = nn.Parameter(torch.rand(1) * 0.02 + 0.01) if rezero else None
        self.scales = list(scales)
        self.t_convs = nn.ModuleList()
        self.acts = []
        for i, scale in enumerate(self.scales):
            last = i + 1 == len(self.scales)
            in_dim = in_channels if i == 0 else hidden_channels
            out_dim = out_channels if last else hidden_channels
            k = kernel_size[i] if isinstance(kernel_size, (list, tuple)) else kernel_size
            self.t_convs.append(nn.ConvTranspose1d(in_dim, out_dim, k, scale, padding=(k - scale) // 2))
            act = (not last) or use_last_layer_act_func
            if act:
                self.t_convs.append(nn.LeakyReLU(negative_slope=0.4))
            self.acts.append(act)
        self.res_channels = min(in_channels, out_channels)
        self.out_channels = out_channels

    def convs(self):
        return [m for m in self.t_convs if isinstance(m, nn.ConvTranspose1d)]


class WaveGlow(nn.Module):
    def __init__(self, n_mel_channels, n_flows, n_group, n_early_every, n_early_size, memory_efficient,
                 spect_scaling, upsample_mode, upsample_first, speaker_embed, cond_layers, cond_hidden_channels,
                 cond_output_channels, cond_kernel_size, cond_residual, cond_padding_mode, WN_config, win_length,
                 hop_length, sampling_rate=48000, cond_res_rezero=False, cond_activation_func='none',
                 negative_slope=None, channel_mixing='1x1conv', mix_first=True, preceived_vol_scaling=False,
                 waveflow=True, yoyo='depreciated', yoyo_WN='depreciated', shift_spect=0., scale_spect=1.,
                 preempthasis=None, use_logvar_channels=False, load_hidden_from_disk=False,
                 transposed_conv_hidden_dim=256, transposed_conv_kernel_size=4, transposed_conv_scales=None,
                 transposed_conv_output_dim=256, transposed_conv_residual=False, transposed_conv_residual_linear=False,
                 transposed_conv_res_rezero=False, group_conv_output_dim=None, group_conv_groupped=True, iso226_empthasis=False):
        super().__init__()
        assert n_group % 2 == 0
        assert hop_length % n_group == 0, "hop_length is not int divisible by n_group"
        wn = dict(WN_config)

        def need(cond, what):
            if not cond:
                raise NotImplementedError(f"ax-core option not built on the HIP path yet: {what}")
        assert any(channel_mixing.lower() in x for x in ("1x1convinvertibleconv1x1invconv",
                                                         "waveflowpermuteheightpermutechannelpermute")), \
            "channel_mixing option is invalid. Options are '1x1conv' or 'permuteheight'"          # ax:24
        mixing = '1x1conv' if channel_mixing.lower() in "1x1convinvertibleconv1x1invconv" else 'permuteheight'   # ax:25
        if waveflow:
            need(n_group <= 64, "waveflow=True with n_group > 64")
        else:
            need(not wn.get('seperable_conv', False), "waveflow=False with seperable_conv")
            need(wn['n_channels'] % 32 == 0, "waveflow=False with n_channels not a multiple of 32")
            need(n_group <= 32, "waveflow=False with n_group > 32")
        need(cond_residual in (False, True, 0, 1, '1x1conv'), f"cond_residual={cond_residual!r}")
        use_tconv = bool(transposed_conv_scales) and bool(transposed_conv_hidden_dim) and bool(transposed_conv_kernel_size)
        need(upsample_first is False or upsample_first is True or upsample_first is None or upsample_first == 0,
             f"upsample_first={upsample_first!r}")
        if upsample_first is True:                                # ax:121-126, 174-186: cond upsampled at model level
            need(use_tconv, "upsample_first=True without a TransposedUpsampleNet")
            need(transposed_conv_residual_linear or not transposed_conv_residual,
                 "transposed_conv_residual with 'nearest' interpolation (F.interpolate rejects align_corners there: the "
                 "reference raises ValueError, glow_ax.py:231)")
        else:
            need(not use_tconv, "TransposedUpsampleNet with upsample_first != True")
        wn_tconv = bool(wn.get('transposed_conv_scales')) and bool(wn.get('transposed_conv_hidden_dim', 256)) \
            and bool(wn.get('transposed_conv_kernel_size', 4))
        need(not (wn_tconv and upsample_first is True), "WN-level TransposedUpsampleNet together with upsample_first=True")
        need(not wn_tconv or wn.get('cond_layers', 1) >= 1, "WN-level TransposedUpsampleNet without WN cond layers")
        self._wn_tconv_factor = int(np.prod(wn['transposed_conv_scales'])) if wn_tconv else 0
        need(cond_padding_mode in ('zeros', 'replicate') and wn.get('cond_padding_mode', 'zeros') in ('zeros', 'replicate'),
             "cond_padding_mode other than 'zeros' / 'replicate'")
        need(not iso226_empthasis, "iso226 emphasis")
        need(wn.get('cond_layers', 1) >= 1, "WN without cond layers")
        need(wn.get('upsample_mode', 'linear') == 'linear', "WN upsample_mode != 'linear'")
        if waveflow:                                                            # glow_ax.py:259 / :433
            assert wn.get('res_skip', True) or wn.get('merge_res_skip', False), \
                "Cannot remove res_skip without using merge_res_skip"
        else:
            assert (wn.get('res_skip', True) or wn.get('merge_res_skip', False)) or wn['n_layers'] == 1, \
                "Cannot remove res_skip without using merge_res_skip"
        if str(wn.get('gated_unit', 'GTU')).upper() not in _lib.GATED_UNITS:
            raise Exception("gated_unit is invalid\nOptions are ('GTU','GTRU','GLU').")     # glow_ax.py:198
        assert not wn.get('rezero', False), "WN ReZero is depreciated"                         # glow_ax.py:272, 450
        need(wn['n_layers'] <= 12, "more than 12 WN layers")
        if waveflow:
            dh = wn.get('n_layers_dilations_h', 1)
            dh = [dh] * wn['n_layers'] if isinstance(dh, int) else list(dh)
            need(all(d >= 1 for d in dh), "height dilation < 1")
            need(wn.get('seperable_conv', False) or wn['kernel_size_h'] * wn['kernel_size_w'] <= 11,
                 "dense in-layer kernels with more than 11 taps (use seperable_conv)")
        else:
            need((wn.get('kernel_size_w') or wn.get('kernel_size')) <= 11, "1-D in-layer kernels wider than 11")
        need(2 * cond_kernel_size - 1 <= 11 and 2 * wn.get('cond_kernel_size', 1) - 1 <= 11, "cond kernels wider than 11")
        assert mixing != 'permuteheight' or n_flows % 2 == 0, "PermuteHeight requires even n_flows"

        self.waveflow = bool(waveflow)
        self.n_flows, self.n_group = n_flows, n_group
        self.n_early_every, self.n_early_size = n_early_every, n_early_size
        self.sampling_rate, self.win_size, self.hop_length = sampling_rate, win_length, hop_length
        self.n_mel_channels = n_mel_channels
        self.channel_mixing, self.mix_first = mixing, bool(mix_first)
        self.cond_padding_mode = cond_padding_mode
        self.ignore_nan = True                                                   # ax:50
        self.has_logvar_channels = bool(use_logvar_channels)
        self.preempthasis = preempthasis
        self.speaker_embed_dim = speaker_embed
        self.multispeaker = speaker_embed > 0 or wn.get('speaker_embed_dim', 0) > 0
        self.cond_residual, self.cond_res_rezero = cond_residual, cond_res_rezero
        self.shift_spect, self.scale_spect = float(shift_spect), float(scale_spect)
        self.vol_scaling = bool(preceived_vol_scaling)
        self.use_hidden_cond = load_hidden_from_disk                              # ax:36: a data-loader hint, not used by the model
        self.upsample_early = upsample_first is True
        self.WN_config = wn
        # activation tables (validated now so that an unsupported name fails at construction)
        self._act_model = _act_code(cond_activation_func if cond_layers else 'none', negative_slope)
        self._act_wn = _act_code(wn.get('cond_activation_func', 'none'), wn.get('negative_slope'))

        if speaker_embed:
            self.speaker_embed = nn.Embedding(512, speaker_embed)                # ax:59-61
        self.cond_in_channels = n_mel_channels * (2 if use_logvar_channels else 1) + speaker_embed   # ax:64
        wn_cond = self.cond_in_channels
        if cond_res_rezero:
            self.alpha = nn.Parameter(torch.rand(1) * 0.02 + 0.01)               # ax:75-76
        self.cond_layers = nn.ModuleList()
        if cond_layers:
            out_c = self.cond_in_channels if (cond_residual is True or cond_residual == 1) else cond_output_channels  # ax:72-73
            if cond_residual == '1x1conv':
                self.res_conv = nn.Conv1d(self.cond_in_channels, out_c, 1)       # ax:80-81
            k = 2 * cond_kernel_size - 1                                         # ax:83
            dims = [self.cond_in_channels] + [cond_hidden_channels] * (cond_layers - 1) + [out_c]
            self.cond_layers = nn.ModuleList([_WNConv((dims[l + 1], dims[l], k)) for l in range(cond_layers)])
            wn_cond = out_c
        if self.upsample_early:                                                  # ax:116-126
            t_out = transposed_conv_output_dim if transposed_conv_output_dim is not None else wn_cond
            self.upsample_net = _TransposedUpsampleNet(wn_cond, t_out, transposed_conv_hidden_dim,
                                                       transposed_conv_kernel_size, transposed_conv_scales, True,
                                                       transposed_conv_residual, transposed_conv_residual_linear,
                                                       transposed_conv_res_rezero)
            self.model_cond_channels = wn_cond
            wn_cond = t_out
        self.group_conv_in = 0
        if group_conv_output_dim:                                                # ax:131-134: per-flow 1x1 conv of the cond
            groups = n_flows if group_conv_groupped else 1
            assert wn_cond % groups == 0, "in_channels must be divisible by groups"
            self.n_flow_group_conv = nn.Conv1d(wn_cond, group_conv_output_dim * n_flows, 1, groups=groups)
            self.group_conv_in = wn_cond
            wn_cond = group_conv_output_dim
        self.wn_cond_channels = wn_cond
        if waveflow:                                                             # ax:166-189 (the 2-D WN does not
            self.WN = nn.ModuleList([_Coupling(_WN2d(wn_cond, wn)) for _ in range(n_flows)])   # depend on the row count)
            self.convinv = nn.ModuleList() if mixing == '1x1conv' else []        # PermuteHeight: no parameters (ax:144)
            n_rem = n_group
            self.z_split_sizes = []
            for k in range(n_flows):
                if k % n_early_every == 0 and k > 0:
                    n_rem -= n_early_size
                    self.z_split_sizes.append(n_early_size)
                assert n_rem > 0, "n_remaining_channels is 0. (increase n_group or decrease n_early_every/n_early_size)"
                if mixing == '1x1conv':
                    self.convinv.append(InvertibleConv1x1(n_rem))
            self.z_split_sizes.append(n_rem)
        else:
            self.WN = nn.ModuleList()
            self.convinv = nn.ModuleList() if mixing == '1x1conv' else []
            n_rem = n_group
            self.z_split_sizes = []
            for k in range(n_flows):
                if k % n_early_every == 0 and k > 0:
                    n_rem -= n_early_size
                    self.z_split_sizes.append(n_early_size)
                assert n_rem > 0, "n_remaining_channels is 0. (increase n_group or decrease n_early_every/n_early_size)"
                need(n_rem % 2 == 0, "odd number of remaining channels")
                if mixing == '1x1conv':
                    self.convinv.append(InvertibleConv1x1(n_rem))
                self.WN.append(_Coupling(_WN1d(n_rem // 2, wn_cond, wn)))
            self.z_split_sizes.append(n_rem)
        # one k=1 linear WN cond layer on the bare mel commutes with the interpolation: folded into the in-layer GEMM
        self._folded = (bool(waveflow) and not cond_layers and not speaker_embed and not wn.get('speaker_embed_dim', 0)
                        and not group_conv_output_dim and not wn_tconv and not self.upsample_early
                        and self.shift_spect == 0. and self.scale_spect == 1.
                        and wn.get('cond_layers', 1) == 1 and wn.get('cond_kernel_size', 1) == 1
                        and self._act_wn[0] == 0)
        self._packed = None
        self._ws = {}
        self._f32_gemm_mode = None
        _cache.hook_invalidate(self)

    # ------------------------------------------------------------------ plumbing ----
    def c_config(self):
        wn = self.WN_config
        return _lib.WaveFlowConfig(n_mel_channels=self.cond_in_channels, n_flows=self.n_flows, n_group=self.n_group,
                                   n_layers=wn['n_layers'], n_channels=wn['n_channels'],
                                   kernel_size_w=wn['kernel_size_w'], kernel_size_h=wn['kernel_size_h'], dilation_h=1,
                                   seperable_conv=1 if wn.get('seperable_conv', False) else 0,
                                   gated_unit=_lib.GATED_UNITS[str(wn.get('gated_unit', 'GTU')).upper()],
                                   merge_res_skip=1 if (wn.get('merge_res_skip', False) or not wn.get('res_skip', True)) else 0,
                                   n_early_every=self.n_early_every, n_early_size=self.n_early_size,
                                   mixing=_lib.MIX_CONV1X1 if self.channel_mixing == '1x1conv' else _lib.MIX_PERMUTE,
                                   mix_first=1 if self.mix_first else 0,
                                   dilation_w=_lib.dilation_array(wn.get('n_layers_dilations_w'), wn['n_layers']),
                                   dilation_h_l=_lib.dilation_array(wn.get('n_layers_dilations_h', 1), wn['n_layers']),
                                   cond_precomputed=0 if self._folded else 1,
                                   f32_gemm_mode=_lib.model_gemm_mode(self._f32_gemm_mode))

    def set_f32_gemm_mode(self, mode):
        """Main loop of THIS model's GEMMs (both cores, and the conditioning operators in front of them): ``"f32"``,
        ``"bf16x3"``, ``"bf16x6"`` or ``None`` / ``"default"`` (the library default).  See ``waveglow.WaveGlow.set_f32_gemm_mode``."""
        _lib.model_gemm_mode(mode)
        self._f32_gemm_mode = mode
        self._invalidate()            # the conditioning operators carry the mode in their packed descriptors
        return self

    def c_config_1d(self):
        wn = self.WN_config
        return _lib.WgaxConfig(n_flows=self.n_flows, n_group=self.n_group, n_early_every=self.n_early_every,
                               n_early_size=self.n_early_size, n_layers=wn['n_layers'], n_channels=wn['n_channels'],
                               kernel_size=wn.get('kernel_size_w') or wn.get('kernel_size'),
                               mixing=_lib.MIX_CONV1X1 if self.channel_mixing == '1x1conv' else _lib.MIX_PERMUTE,
                               mix_first=1 if self.mix_first else 0, ignore_nan=1 if self.ignore_nan else 0,
                               gated_unit=_lib.GATED_UNITS[str(wn.get('gated_unit', 'GTU')).upper()],
                               merge_res_skip=1 if (wn.get('merge_res_skip', False) or not wn.get('res_skip', True)) else 0,
                               dilation_w=_lib.dilation_array(wn.get('n_layers_dilations_w'), wn['n_layers']),
                               f32_gemm_mode=_lib.model_gemm_mode(self._f32_gemm_mode))

    def _invalidate(self):
        self._packed, self._ws = None, {}
        for m in (self.convinv if isinstance(self.convinv, nn.ModuleList) else []):
            if hasattr(m, 'W_inverse'):
                del m.W_inverse

    def _apply(self, fn, *a, **kw):
        self._invalidate()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self._invalidate()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def repack(self):
        self._invalidate()

    def remove_weightnorm(self):
        for m in self.modules():
            if isinstance(m, _WNConv):
                m.remove_weight_norm()
        self._invalidate()

    def _dense(self, layer, stream, keep):
        lib = _lib.lib()
        if getattr(layer, 'weight_v', None) is not None:
            v = layer.weight_v.detach().float().contiguous()
            g = layer.weight_g.detach().float().contiguous()
            w = torch.empty_like(v)
            _lib.check(lib.ctts_fold_weightnorm_f32(_lib.ptr(v), _lib.ptr(g), _lib.ptr(w), v.shape[0], v[0].numel(),
                                                   stream), "ctts_fold_weightnorm_f32")
            keep += [v, g, w]
            return w
        w = layer.weight.detach().float().contiguous()
        keep.append(w)
        return w

    def _ensure_packed(self, device):
        """-> (blob, cond_ops): packed WaveFlow weights, and (unless the cond layer is folded) the conv operators
        of the conditioning stacks {'model': [...], 'wn': [[...] per flow]}."""
        key = _cache.param_key(self)
        if self._packed is not None and self._packed[0] == device and self._packed[3] == key:
            return self._packed[1], self._packed[2]
        if device.type != 'cuda':
            raise _lib.HipLibraryError("WaveFlow HIP path needs the model on a GPU (no CPU fallback)")
        lib = _lib.lib()
        wn_cfg = self.WN_config
        n_layers = wn_cfg['n_layers']
        if self.waveflow:
            cfg = self.c_config()
            nbytes = lib.ctts_waveflow_packed_bytes(C.byref(cfg))
        else:
            cfg = self.c_config_1d()
            nbytes = lib.ctts_wgax_packed_bytes(C.byref(cfg))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported ax-core config: " + lib.ctts_last_error().decode())
        sep = self.waveflow and isinstance(self.WN[0].WN.in_layers[0], nn.ModuleList)
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            blob = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            keep = []

            def dev(t):
                t = t.detach().float().contiguous()
                keep.append(t)
                return t.data_ptr()

            def arr(fn):
                a = (C.c_void_p * n_layers)()
                for i in range(n_layers):
                    a[i] = fn(i)
                return a
            # res_skip=False (glow_ax.py:401 / :609: `res_skip_acts = acts`): the same launch with an identity weight
            no_rs = not wn_cfg.get('res_skip', True)
            if no_rs:
                eye = torch.eye(wn_cfg['n_channels'], dtype=torch.float32, device=device).contiguous()
                zero_b = torch.zeros(wn_cfg['n_channels'], dtype=torch.float32, device=device)
                keep += [eye, zero_b]
            for k in range(self.n_flows):
                wn = self.WN[k].WN
                if not self.waveflow:
                    fw = _lib.WgaxFlowWeights()
                    fw.start_w = self._dense(wn.start, stream, keep).data_ptr()
                    fw.start_b = dev(wn.start.bias)
                    fw.in_w = arr(lambda i: self._dense(wn.in_layers[i], stream, keep).data_ptr())
                    fw.in_b = arr(lambda i: dev(wn.in_layers[i].bias))
                    fw.rs_w = arr(lambda i: eye.data_ptr() if no_rs else self._dense(wn.res_skip_layers[i], stream, keep).data_ptr())
                    fw.rs_b = arr(lambda i: zero_b.data_ptr() if no_rs else dev(wn.res_skip_layers[i].bias))
                    fw.end_w = dev(wn.end.weight)
                    fw.end_b = dev(wn.end.bias)
                    if self.channel_mixing == '1x1conv':
                        # efficient_modules.py:271-276: W.float().inverse(), cached on the module as W_inverse
                        W = self.convinv[k].weight.detach().squeeze(-1)
                        W_inverse = W.float().cpu().inverse().to(device).contiguous()
                        self.convinv[k].W_inverse = W_inverse[..., None]
                        keep.append(W_inverse)
                        fw.w_inverse = W_inverse.data_ptr()
                    _lib.check(lib.ctts_wgax_pack_flow(C.byref(cfg), k, C.byref(fw), _lib.ptr(blob), stream),
                               f"ctts_wgax_pack_flow({k})")
                    continue
                fw = _lib.WaveFlowFlowWeights()
                fw.start_w = self._dense(wn.start, stream, keep).data_ptr()
                fw.start_b = dev(wn.start.bias)
                if self._folded:
                    fw.cond_w = self._dense(wn.cond_layers[0], stream, keep).data_ptr()
                    fw.cond_b = dev(wn.cond_layers[0].bias)
                gemm = (lambda i: wn.in_layers[i][1]) if sep else (lambda i: wn.in_layers[i])
                fw.in_w = arr(lambda i: self._dense(gemm(i), stream, keep).data_ptr())
                fw.in_b = arr(lambda i: dev(gemm(i).bias))
                if sep:
                    fw.dw_w = arr(lambda i: self._dense(wn.in_layers[i][0], stream, keep).data_ptr())
                    fw.dw_b = arr(lambda i: dev(wn.in_layers[i][0].bias))
                fw.rs_w = arr(lambda i: eye.data_ptr() if no_rs else self._dense(wn.res_skip_layers[i], stream, keep).data_ptr())
                fw.rs_b = arr(lambda i: zero_b.data_ptr() if no_rs else dev(wn.res_skip_layers[i].bias))
                fw.end_w = dev(wn.end.weight)
                fw.end_b = dev(wn.end.bias)
                if self.channel_mixing == '1x1conv':                            # efficient_modules.py:271-276
                    W = self.convinv[k].weight.detach().squeeze(-1)
                    W_inverse = W.float().cpu().inverse().to(device).contiguous()
                    self.convinv[k].W_inverse = W_inverse[..., None]
                    keep.append(W_inverse)
                    fw.w_inverse = W_inverse.data_ptr()
                _lib.check(lib.ctts_waveflow_pack_flow(C.byref(cfg), k, C.byref(fw), _lib.ptr(blob), stream),
                           f"ctts_waveflow_pack_flow({k})")
            ops = None
            gm = _lib.model_gemm_mode(self._f32_gemm_mode)
            if not self._folded:
                def stack(layers, act, act_last):
                    out = []
                    for l, layer in enumerate(layers):
                        a = act if (act_last or l != len(layers) - 1) else (0, 0.0)
                        out.append(_CondConv(self._dense(layer, stream, keep), layer.bias, a[0], a[1], device, stream, gm))
                    return out
                ops = {'model': stack(self.cond_layers, self._act_model, True),          # ax:293-297: every layer
                       'res_conv': (_CondConv(self.res_conv.weight.detach().float(), self.res_conv.bias, 0, 0.0, device, stream, gm)
                                    if hasattr(self, 'res_conv') else None),            # ax:303-304
                       'group': self._group_conv_ops(device, stream),                  # ax:131-134, 320-321
                       'wn_tconv': ([[_TransposedConv(m.weight.detach().float(), m.bias, sc, 1 if act else 0, 0.4, device, stream, gm)
                                      for m, sc, act in zip(c.WN.upsample_net.convs(), c.WN.upsample_net.scales,
                                                            c.WN.upsample_net.acts)] for c in self.WN]
                                    if self._wn_tconv_factor else None),                # glow_ax.py:362-373 / 545-554
                       'tconv': ([_TransposedConv(m.weight.detach().float(), m.bias, sc, 1 if act else 0, 0.4, device, stream, gm)
                                  for m, sc, act in zip(self.upsample_net.convs(), self.upsample_net.scales,
                                                        self.upsample_net.acts)]
                                 if self.upsample_early else None),                     # glow_ax.py:214-226
                       'wn': [stack(c.WN.cond_layers, self._act_wn, wn_cfg.get('cond_out_activation_func', True))
                              for c in self.WN]}                                       # glow_ax.py:573-577
            torch.cuda.current_stream(device).synchronize()
        self._packed = (device, blob, ops, key)
        return blob, ops

    def _to_latent_length(self, h, B, rows_n, hT, hld, T, ld, dst, interpolate, stream):
        """``_upsample_mels`` after the transposed convs (ax:177-185, glow_ax.py:365-372 / 548-553): linear interpolation
        (align_corners=True) to the latent's length when the net's factor is not hop // n_group, else a centre crop -
        the 1-D rule ``[pad_l : -pad_r]`` or the 2-D WN's ``[pad : -(pad + pad % 2)]``."""
        lib = _lib.lib()
        if interpolate:
            for b in range(B):
                _lib.check(lib.ctts_resample_rows_f32(_lib.ptr(h[b]), _lib.ptr(dst[b]), 1, rows_n, hT, hld, PAD, T, ld, PAD, 0,
                                                      0.0, stream), "ctts_resample_rows_f32")
            return
        if self.waveflow:
            pad = (hT - T) // 2
            lo, hi = pad, pad + pad % 2
        else:
            lo, hi = (hT - T) // 2, -((T - hT) // 2)
        if hT - lo - hi != T or lo < 0:
            raise RuntimeError(f"upsampled conditioning of length {hT} cannot be cropped to the latent's {T} columns "
                               f"(the reference fails on the same shapes)")
        for b in range(B):      # shifted copy: 'nearest' at scale 1 from the source advanced by `lo` columns
            _lib.check(lib.ctts_resample_rows_f32(h[b].data_ptr() + 4 * lo, _lib.ptr(dst[b]), 1, rows_n, hT - lo, hld, PAD, T,
                                                  ld, PAD, 2, 1.0, stream), "ctts_resample_rows_f32")

    def _group_conv_ops(self, device, stream):
        """One dense 1x1 operator per flow out of ``n_flow_group_conv``: flow k owns output rows [k*out, (k+1)*out) and,
        when grouped, reads only its own slice of the input channels -> [(op, first input row)] or None."""
        if not self.group_conv_in:
            return None
        conv = self.n_flow_group_conv
        w, b = conv.weight.detach().float(), conv.bias.detach().float()
        out = w.shape[0] // self.n_flows
        cin_g = w.shape[1]
        ops = []
        for k in range(self.n_flows):
            row0 = k * cin_g if conv.groups > 1 else 0
            ops.append((_CondConv(w[k * out:(k + 1) * out].contiguous(), b[k * out:(k + 1) * out].contiguous(), 0, 0.0,
                                  device, stream, _lib.model_gemm_mode(self._f32_gemm_mode)), row0))
        return ops

    def _cond_frames(self, ops, cond, speaker_ids, stream, out_steps=None):
        """ax:281-307 + glow_ax.py:566-577 -> ([n_flows][B][2C*n_layers][ld] padded rows, ld, T): at frame rate
        (T = frames), or with ``upsample_first=True`` at the rate of the latent (T = out_steps)."""
        lib = _lib.lib()
        dev = cond.device
        B, Cm, Fr = cond.shape
        ld = _ld_for(Fr)
        if self.multispeaker:
            if speaker_ids is None:
                raise Exception("This WaveFlow/WaveGlow model requires speaker ids or speaker embeddings.")   # ax:288
            ids = speaker_ids.to(device=dev, dtype=torch.int64).contiguous()

        def rows(c, ld_=None):
            return torch.zeros(B, _r16(c), ld_ or ld, dtype=torch.float32, device=dev)
        # model-level input: [mel (+logvar), shifted and scaled | speaker embedding]
        x0 = rows(self.cond_in_channels)
        for b in range(B):
            _lib.check(lib.ctts_pad_rows_f32(_lib.ptr(cond[b]), 0, Fr, _lib.ptr(x0[b]), 1, Cm, Fr, ld, PAD, stream),
                       "ctts_pad_rows_f32")
        if self.shift_spect != 0. or self.scale_spect != 1.:                    # ax:281-284
            _lib.check(lib.ctts_affine_rows_f32(_lib.ptr(x0), B, x0.shape[1], Cm, Fr, ld, PAD, self.shift_spect,
                                                self.scale_spect, stream), "ctts_affine_rows_f32")
        if self.speaker_embed_dim:
            tab = self.speaker_embed.weight.detach().float().contiguous()
            _lib.check(lib.ctts_embed_rows_f32(_lib.ptr(tab), _lib.ptr(ids), _lib.ptr(x0), Cm, self.speaker_embed_dim, B,
                                               x0.shape[1], Fr, ld, PAD, stream), "ctts_embed_rows_f32")
        # conv stack, rezero, residual (identity or 1x1 conv of the input, ax:299-307)
        sdim = self.WN_config.get('speaker_embed_dim', 0)
        grp = ops.get('group')
        c_wn_in = self.group_conv_in if grp else self.wn_cond_channels       # channels of the tensor the flows share
        c_model = self.model_cond_channels if self.upsample_early else c_wn_in
        # (+16 spare rows: a grouped per-flow conv reads its input slice rounded up to the primitive's 16 channels)
        xw = rows(c_model + (0 if (self.upsample_early or grp) else sdim) + (16 if grp else 0))
        h = x0
        for op in ops['model']:
            y = rows(op.c_out)
            op(h, y, B, Fr, ld, stream, self.cond_padding_mode)
            h = y
        alpha = self.alpha.detach().float().contiguous() if (self.cond_res_rezero and len(ops['model'])) else None
        resid = None
        if self.cond_residual and len(ops['model']):
            resid = x0
            if ops['res_conv'] is not None:
                resid = rows(ops['res_conv'].c_out)
                ops['res_conv'](x0, resid, B, Fr, ld, stream)
        for b in range(B):     # row counts differ between the buffers: one utterance per call
            _lib.check(lib.ctts_scale_add_rows_f32(_lib.ptr(h[b]), _lib.ptr(alpha), _lib.ptr(None if resid is None else resid[b]),
                                                   _lib.ptr(xw[b]), 1, c_model, Fr, ld, PAD, stream),
                       "ctts_scale_add_rows_f32")
        T = Fr
        if self.upsample_early:                                                  # ax:174-186 with TransposedUpsampleNet
            net = self.upsample_net
            factor = int(np.prod(net.scales))
            h, hT, hld = xw, Fr, ld
            for tc in ops['tconv']:
                h, hT, hld = tc(h, B, hT, hld, stream)
            if net.residual:                                                     # glow_ax.py:229-241
                xi = rows(net.res_channels, hld)
                for b in range(B):
                    _lib.check(lib.ctts_resample_rows_f32(_lib.ptr(xw[b]), _lib.ptr(xi[b]), 1, net.res_channels, Fr, ld, PAD,
                                                          hT, hld, PAD, 1 if net.residual_linear else 2, float(factor),
                                                          stream), "ctts_resample_rows_f32")
                rw = net.res_weight.detach().float().contiguous() if net.res_weight is not None else None
                for b in range(B):
                    _lib.check(lib.ctts_scale_add_rows_f32(_lib.ptr(h[b]), _lib.ptr(rw), _lib.ptr(xi[b]), _lib.ptr(h[b]), 1,
                                                           net.res_channels, hT, hld, PAD, stream), "ctts_scale_add_rows_f32")
                    if net.out_channels > net.res_channels and rw is not None:
                        tail = h[b][net.res_channels:]
                        _lib.check(lib.ctts_scale_add_rows_f32(_lib.ptr(tail), _lib.ptr(rw), None, _lib.ptr(tail), 1,
                                                               net.out_channels - net.res_channels, hT, hld, PAD, stream),
                                   "ctts_scale_add_rows_f32")
            # to the length of the latent (ax:177-178: 'linear', align_corners=True)
            T = out_steps
            ld = _ld_for(T)
            xw = rows(c_wn_in + (0 if grp else sdim) + (16 if grp else 0))
            self._to_latent_length(h, B, c_wn_in, hT, hld, T, ld, xw, factor != self.hop_length // self.n_group, stream)
        # per flow: WN speaker embedding, conv stack -> 2C*n_layers rows
        C2L = 2 * self.WN_config['n_channels'] * self.WN_config['n_layers']
        wn_up = ops.get('wn_tconv')                 # per-flow TransposedUpsampleNet behind the WN cond stack
        T_out, ld_out = (out_steps, _ld_for(out_steps)) if wn_up else (T, ld)
        frames = torch.zeros(self.n_flows, B, C2L, ld_out, dtype=torch.float32, device=dev)
        xk = rows(self.wn_cond_channels + sdim) if grp else xw
        for k in range(self.n_flows):
            if grp:                                                              # this flow's chunk of the group conv
                op, row0 = grp[k]
                op(xw[:, row0:], xk, B, T, ld, stream)
            if sdim:
                tab = self.WN[k].WN.speaker_embed.weight.detach().float().contiguous()
                _lib.check(lib.ctts_embed_rows_f32(_lib.ptr(tab), _lib.ptr(ids), _lib.ptr(xk), self.wn_cond_channels, sdim,
                                                   B, xk.shape[1], T, ld, PAD, stream), "ctts_embed_rows_f32")
            h = xk
            for l, op in enumerate(ops['wn'][k]):
                y = frames[k] if (l == len(ops['wn'][k]) - 1 and not wn_up) else rows(op.c_out)
                op(h, y, B, T, ld, stream, self.WN_config.get('cond_padding_mode', 'zeros'))
                h = y
            if wn_up:                                                            # glow_ax.py:389-390 -> :362-373
                hT, hld = T, ld
                for tc in wn_up[k]:
                    h, hT, hld = tc(h, B, hT, hld, stream)
                self._to_latent_length(h, B, C2L, hT, hld, T_out, ld_out, frames[k],
                                       self._wn_tconv_factor != self.hop_length // self.n_group, stream)
        return frames, ld_out, T_out

    # --------------------------------------------------------------------- the path ----
    def inverse(self, z, cond, speaker_ids=None, return_CPU=True):
        """efficient_model_ax.py:279-357: z [B, T] (noise, sigma applied), cond [B, n_mel(*2), frames]."""
        if not self.waveflow:
            return self._inverse_1d(z, cond, speaker_ids, return_CPU)
        device = cond.device
        blob, ops = self._ensure_packed(device)
        lib = _lib.lib()
        cfg = self.c_config()
        mel = cond.detach().float().contiguous()
        zz = z.detach().to(device=device, dtype=torch.float32).contiguous()
        B, T = zz.shape
        assert mel.shape[0] == B and mel.shape[1] == self.n_mel_channels * (2 if self.has_logvar_channels else 1)
        key = (device, B, T)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib.ctts_waveflow_workspace_bytes(C.byref(cfg), B, T)
            if nbytes == 0:
                raise _lib.HipLibraryError("WaveFlow workspace query failed: " + lib.ctts_last_error().decode())
            self._drop_workspaces()
            ws = self._ws.setdefault(key, torch.zeros(nbytes // 4, dtype=torch.float32, device=device))
        audio = torch.empty(B, T, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            if self._folded:
                _lib.check(lib.ctts_waveflow_inverse_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(zz), _lib.ptr(mel),
                                                        _lib.ptr(audio), B, T, mel.shape[2], _lib.ptr(ws),
                                                        ws.numel() * 4, stream), "ctts_waveflow_inverse_f32")
            else:
                frames, ld, n_cond = self._cond_frames(ops, mel, speaker_ids, stream, out_steps=T // self.n_group)
                _lib.check(lib.ctts_waveflow_inverse_cond_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(zz), _lib.ptr(frames),
                                                             ld, PAD, _lib.ptr(audio), B, T, n_cond, _lib.ptr(ws),
                                                             ws.numel() * 4, stream), "ctts_waveflow_inverse_cond_f32")
            if self.vol_scaling:       # ax:342-344
                _lib.check(lib.ctts_vol_unscale_f32(_lib.ptr(audio), audio.numel(), stream), "ctts_vol_unscale_f32")
            if self.preempthasis:      # ax:351-355 (scipy lfilter on the host there; here on the device, in place)
                _lib.check(lib.ctts_deemphasis_f32(_lib.ptr(audio), _lib.ptr(audio), B, T, float(self.preempthasis),
                                                  stream), "ctts_deemphasis_f32")
        if return_CPU:
            audio = audio.cpu()
            # the copy synchronised the stream: a row-queue abort of THIS call is known now - raise instead of handing out
            # its NaN audio (a call that stays on the device reports it through the next call on this workspace: CTTS_E_ABORT)
            with torch.cuda.device(device):
                _lib.check(lib.ctts_waveflow_abort_status(C.byref(cfg), B, T, _lib.ptr(ws), ws.numel() * 4, stream),
                           "WaveFlow.inverse")
        return audio, None

    def _drop_workspaces(self):
        """Forget the cached WaveFlow workspaces (another shape is coming) - but not an abort one of them still holds: a
        row-queue call that stayed on the device (``return_CPU=False``) reports a given-up wait through the workspace's status
        word, and dropping the workspace would drop the report.  Raises HipLibraryError (CTTS_E_ABORT) then."""
        old, self._ws = self._ws, {}
        if not self.waveflow:
            return
        lib = _lib.lib()
        cfg = self.c_config()
        for (device, B, T), ws in old.items():
            with torch.cuda.device(device):
                stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                _lib.check(lib.ctts_waveflow_abort_status(C.byref(cfg), B, T, _lib.ptr(ws), ws.numel() * 4, stream),
                           "WaveFlow.inverse (an earlier call on a workspace that is being replaced)")

    def _inverse_1d(self, z, cond, speaker_ids, return_CPU):
        """waveflow=False: efficient_model_ax.py:279-357 with AffineCouplingBlock + 1-D WN (ctts_wgax_inverse_f32)."""
        device = cond.device
        blob, ops = self._ensure_packed(device)
        lib = _lib.lib()
        cfg = self.c_config_1d()
        mel = cond.detach().float().contiguous()
        zz = z.detach().to(device=device, dtype=torch.float32).contiguous()
        B, T = zz.shape
        assert mel.shape[0] == B and mel.shape[1] == self.n_mel_channels * (2 if self.has_logvar_channels else 1)
        assert T % self.n_group == 0, "z length is not a multiple of n_group"
        key = (device, B, T)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib.ctts_wgax_workspace_bytes(C.byref(cfg), B, T)
            if nbytes == 0:
                raise _lib.HipLibraryError("ax WaveGlow workspace query failed: " + lib.ctts_last_error().decode())
            self._ws = {}
            ws = self._ws.setdefault(key, torch.zeros(nbytes // 4, dtype=torch.float32, device=device))
        audio = torch.empty(B, T, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            frames, ld, n_cond = self._cond_frames(ops, mel, speaker_ids, stream, out_steps=T // self.n_group)
            _lib.check(lib.ctts_wgax_inverse_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(zz), _lib.ptr(frames), ld, PAD,
                                                 n_cond, _lib.ptr(audio), B, T, _lib.ptr(ws), ws.numel() * 4,
                                                 stream), "ctts_wgax_inverse_f32")
            if self.vol_scaling:
                _lib.check(lib.ctts_vol_unscale_f32(_lib.ptr(audio), audio.numel(), stream), "ctts_vol_unscale_f32")
            if self.preempthasis:
                _lib.check(lib.ctts_deemphasis_f32(_lib.ptr(audio), _lib.ptr(audio), B, T, float(self.preempthasis),
                                                  stream), "ctts_deemphasis_f32")
        if return_CPU:
            audio = audio.cpu()
        return audio, None

    def _prep_spect(self, spect, artifact_trimming):
        p = next(self.parameters())
        spect = spect.to(p.device, p.dtype)
        if spect.dim() == 2:
            spect = spect[None, ...]
        if artifact_trimming > 0:
            spect = F.pad(spect, (0, artifact_trimming), value=0.0)
        return spect

    @torch.no_grad()
    def infer(self, spect, speaker_ids=None, artifact_trimming=1, sigma=1., t_scaler=1.0, return_CPU=True):
        """efficient_model_ax.py:359-388."""
        input_dtype = spect.dtype
        spect = self._prep_spect(spect, artifact_trimming)
        batch_dim, _, steps = spect.shape
        samples = (steps - 1) * self.hop_length * t_scaler
        samples = int(samples - (samples % self.n_group))
        z = spect.new_empty((batch_dim, samples))
        if sigma > 0:
            z.normal_(std=sigma)
        else:
            z.zero_()
        audio, _ = self.inverse(z, spect, speaker_ids, return_CPU=return_CPU)
        if artifact_trimming > 0:
            audio = audio[:, :-artifact_trimming * self.hop_length]
        return audio.to(input_dtype)

    @torch.no_grad()
    def infer_from_noise(self, spect, z_scaled, speaker_ids=None, artifact_trimming=1, return_CPU=True):
        """Deterministic entry: ``infer`` (efficient_model_ax.py:359-388) with the latent it would draw at :376-378 given
        by the caller - ``z_scaled`` [B, (F + artifact_trimming - 1) * hop - remainder], sigma already applied."""
        input_dtype = spect.dtype
        spect = self._prep_spect(spect, artifact_trimming)
        samples = (spect.shape[2] - 1) * self.hop_length
        samples -= samples % self.n_group
        if tuple(z_scaled.shape) != (spect.shape[0], samples):
            raise ValueError(f"z_scaled is {tuple(z_scaled.shape)}, this mel needs {(spect.shape[0], samples)}")
        audio, _ = self.inverse(z_scaled, spect, speaker_ids, return_CPU=return_CPU)
        if artifact_trimming > 0:
            audio = audio[:, :-artifact_trimming * self.hop_length]
        return audio.to(input_dtype)

    def forward(self, *a, **kw):
        raise NotImplementedError("training direction is outside the inference hot path")
