"""WaveFlow / "ax" WaveGlow core on the MI355X HIP path (BASELINE config 4).

Host-side mirror of ``/root/reference/CookieTTS/_4_mtw/waveglow/efficient_model_ax.py``
``WaveGlow`` (:18-169 constructor, :279-357 ``inverse``, :359-388 ``infer``) for the option subset
of BASELINE config 4: ``waveflow=True`` (``WaveFlowCoupling`` + ``WN_2d``, efficient_modules.py:19-65,
glow_ax.py:421-635), ``channel_mixing='permuteheight'``, ``mix_first=False``, no model-level cond
layers, no speaker embedding, linear-interpolated conditioning.  Same constructor kwargs, same
``state_dict`` keys (``WN.k.WN.{start,cond_layers.0,in_layers.i,res_skip_layers.i}.{weight_g,weight_v,
bias}``, ``WN.k.WN.end.{weight,bias}``), same ``infer`` / ``inverse`` contracts (output length
``(F-1)*hop`` with the default ``artifact_trimming=1``; ``return_CPU=True`` moves the result to
the host like the reference).  Everything else the constructor accepts raises NotImplementedError.
All arithmetic runs in the C-ABI HIP library (``ctts_waveflow_inverse_f32``); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

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
    def __init__(self, n_mel, n_layers, n_channels, kh, kw):
        super().__init__()
        self.n_layers, self.n_channels = n_layers, n_channels
        self.start = _WNConv((n_channels, 1, 1, 1))
        self.end = nn.Module()
        self.end.weight = nn.Parameter(torch.zeros(2, n_channels, 1, 1))        # zero-init, glow_ax.py:454-457
        self.end.bias = nn.Parameter(torch.zeros(2))
        self.cond_layers = nn.ModuleList([_WNConv((2 * n_channels * n_layers, n_mel, 1))])
        self.in_layers = nn.ModuleList([_WNConv((2 * n_channels, n_channels, kh, kw)) for _ in range(n_layers)])
        self.res_skip_layers = nn.ModuleList([
            _WNConv((2 * n_channels if i < n_layers - 1 else n_channels, n_channels, 1, 1)) for i in range(n_layers)])


class _Coupling(nn.Module):
    def __init__(self, wn):
        super().__init__()
        self.WN = wn


class WaveGlow(nn.Module):
    def __init__(self, n_mel_channels, n_flows, n_group, n_early_every, n_early_size, memory_efficient,
                 spect_scaling, upsample_mode, upsample_first, speaker_embed, cond_layers, cond_hidden_channels,
                 cond_output_channels, cond_kernel_size, cond_residual, cond_padding_mode, WN_config, win_length,
                 hop_length, sampling_rate=48000, cond_res_rezero=False, cond_activation_func='none',
                 negative_slope=None, channel_mixing='1x1conv', mix_first=True, preceived_vol_scaling=False,
                 waveflow=True, yoyo='depreciated', yoyo_WN='depreciated', shift_spect=0., scale_spect=1.,
                 preempthasis=None, use_logvar_channels=False, load_hidden_from_disk=False, **unsupported):
        super().__init__()
        assert n_group % 2 == 0
        assert hop_length % n_group == 0, "hop_length is not int divisible by n_group"
        wn = dict(WN_config)

        def need(cond, what):
            if not cond:
                raise NotImplementedError(f"ax-core option not built on the HIP path yet: {what}")
        need(waveflow, "waveflow=False (AffineCouplingBlock + 1-D WN of the ax core)")
        need(channel_mixing.lower() in "waveflowpermuteheightpermutechannelpermute", "channel_mixing='1x1conv'")
        need(not mix_first, "mix_first=True")
        need(cond_layers == 0 and not cond_residual, "model-level cond_layers / cond_residual")
        need(speaker_embed == 0 and wn.get('speaker_embed_dim', 0) == 0, "speaker embeddings")
        need(not upsample_first, "upsample_first")
        need(n_early_every > n_flows, "early outputs (n_early_every <= n_flows)")
        need(shift_spect == 0. and scale_spect == 1. and not preceived_vol_scaling, "spect shift/scale, vol scaling")
        need(not preempthasis and not use_logvar_channels and not load_hidden_from_disk, "preempthasis / logvar / hidden cond")
        need(not unsupported.get('iso226_empthasis', False) and not unsupported.get('transposed_conv_scales'),
             "iso226 emphasis / transposed-conv upsampling")
        need(wn.get('cond_layers', 1) == 1 and wn.get('cond_kernel_size', 1) == 1
             and wn.get('cond_activation_func', 'none') == 'none', "WN cond stack other than one k=1 layer")
        need(wn.get('upsample_mode', 'linear') == 'linear', "WN upsample_mode != 'linear'")
        need(not wn.get('seperable_conv', False) and wn.get('res_skip', True) and not wn.get('merge_res_skip', False),
             "seperable_conv / merge_res_skip")
        need(wn.get('gated_unit', 'GTU') == 'GTU' and not wn.get('rezero', False), "gate other than GTU / rezero")
        need(wn.get('n_layers_dilations_w') is None, "custom width dilations")
        dh = wn.get('n_layers_dilations_h', 1)
        dh = [dh] * wn['n_layers'] if isinstance(dh, int) else list(dh)
        need(all(d == 1 for d in dh), "height dilation != 1")
        assert n_flows % 2 == 0, "PermuteHeight requires even n_flows"

        self.n_flows, self.n_group = n_flows, n_group
        self.n_early_every, self.n_early_size = n_early_every, n_early_size
        self.sampling_rate, self.win_size, self.hop_length = sampling_rate, win_length, hop_length
        self.n_mel_channels = n_mel_channels
        self.channel_mixing, self.mix_first = 'permuteheight', mix_first
        self.has_logvar_channels = False
        self.multispeaker = False
        self.WN_config = wn
        self.WN = nn.ModuleList([
            _Coupling(_WN2d(n_mel_channels, wn['n_layers'], wn['n_channels'], wn['kernel_size_h'], wn['kernel_size_w']))
            for _ in range(n_flows)])
        self._packed = None
        self._ws = {}

    # ------------------------------------------------------------------ plumbing ----
    def c_config(self):
        wn = self.WN_config
        return _lib.WaveFlowConfig(n_mel_channels=self.n_mel_channels, n_flows=self.n_flows, n_group=self.n_group,
                                   n_layers=wn['n_layers'], n_channels=wn['n_channels'],
                                   kernel_size_w=wn['kernel_size_w'], kernel_size_h=wn['kernel_size_h'], dilation_h=1)

    def _invalidate(self):
        self._packed, self._ws = None, {}

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
        if self._packed is not None and self._packed[0] == device:
            return self._packed[1]
        if device.type != 'cuda':
            raise _lib.HipLibraryError("WaveFlow HIP path needs the model on a GPU (no CPU fallback)")
        lib = _lib.lib()
        cfg = self.c_config()
        nbytes = lib.ctts_waveflow_packed_bytes(C.byref(cfg))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported WaveFlow config: " + lib.ctts_last_error().decode())
        n_layers = self.WN_config['n_layers']
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            blob = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            keep = []

            def dev(t):
                t = t.detach().float().contiguous()
                keep.append(t)
                return t.data_ptr()
            for k in range(self.n_flows):
                wn = self.WN[k].WN
                fw = _lib.WaveFlowFlowWeights()
                fw.start_w = self._dense(wn.start, stream, keep).data_ptr()
                fw.start_b = dev(wn.start.bias)
                fw.cond_w = self._dense(wn.cond_layers[0], stream, keep).data_ptr()
                fw.cond_b = dev(wn.cond_layers[0].bias)
                arrs = {}
                for name, layers in (("in", wn.in_layers), ("rs", wn.res_skip_layers)):
                    wa, ba = (C.c_void_p * n_layers)(), (C.c_void_p * n_layers)()
                    for i in range(n_layers):
                        wa[i] = self._dense(layers[i], stream, keep).data_ptr()
                        ba[i] = dev(layers[i].bias)
                    arrs[name] = (wa, ba)
                fw.in_w, fw.in_b = arrs["in"]
                fw.rs_w, fw.rs_b = arrs["rs"]
                fw.end_w = dev(wn.end.weight)
                fw.end_b = dev(wn.end.bias)
                _lib.check(lib.ctts_waveflow_pack_flow(C.byref(cfg), k, C.byref(fw), _lib.ptr(blob), stream),
                           f"ctts_waveflow_pack_flow({k})")
            torch.cuda.current_stream(device).synchronize()
        self._packed = (device, blob)
        return blob

    # --------------------------------------------------------------------- the path ----
    def inverse(self, z, cond, speaker_ids=None, return_CPU=True):
        """efficient_model_ax.py:279-357: z [B, T] (noise, sigma applied), cond [B, n_mel, frames]."""
        device = cond.device
        blob = self._ensure_packed(device)
        lib = _lib.lib()
        cfg = self.c_config()
        mel = cond.detach().float().contiguous()
        zz = z.detach().to(device=device, dtype=torch.float32).contiguous()
        B, T = zz.shape
        assert mel.shape[0] == B and mel.shape[1] == self.n_mel_channels
        key = (device, B, T)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib.ctts_waveflow_workspace_bytes(C.byref(cfg), B, T)
            if nbytes == 0:
                raise _lib.HipLibraryError("WaveFlow workspace query failed: " + lib.ctts_last_error().decode())
            self._ws = {}
            ws = self._ws.setdefault(key, torch.zeros(nbytes // 4, dtype=torch.float32, device=device))
        audio = torch.empty(B, T, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            _lib.check(lib.ctts_waveflow_inverse_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(zz), _lib.ptr(mel),
                                                    _lib.ptr(audio), B, T, mel.shape[2], _lib.ptr(ws),
                                                    ws.numel() * 4, stream), "ctts_waveflow_inverse_f32")
        if return_CPU:
            audio = audio.cpu()
        return audio, None

    @torch.no_grad()
    def infer(self, spect, speaker_ids=None, artifact_trimming=1, sigma=1., t_scaler=1.0, return_CPU=True):
        """efficient_model_ax.py:359-388."""
        input_dtype = spect.dtype
        p = next(self.parameters())
        spect = spect.to(p.device, p.dtype)
        if spect.dim() == 2:
            spect = spect[None, ...]
        if artifact_trimming > 0:
            spect = F.pad(spect, (0, artifact_trimming), value=0.0)
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

    def forward(self, *a, **kw):
        raise NotImplementedError("training direction is outside the inference hot path")
