"""WaveGlow vocoder (NVIDIA-style topology) on the MI355X HIP path.

Host-side mirror of the reference's ``WaveGlow`` ``nn.Module``
(``/root/reference/CookieTTS/_4_mtw/waveglow/glow.py:225-350``): same constructor
kwargs, same ``state_dict`` keys (``upsample.*``, ``WN.{k}.{start,end,cond_layers.j,
in_layers.i,res_skip_layers.i}.*`` with ``weight_g``/``weight_v``, ``convinv.{k}.conv.weight``),
same ``infer(spect, speaker_id=None, sigma=1.0)`` contract, so reference checkpoints
and callers drop in.  All arithmetic of ``infer`` runs in the C-ABI HIP library
(``include/cookietts_hip.h``); PyTorch only owns device memory, the stream and the RNG
draw.  There is no CPU fallback: without the library, or with CPU tensors, it raises.

Deliberate deviations from reference quirks (SURVEY.md §8a "Quirks"):
  * early-output noise (glow.py:342-347) is drawn with the device generator for any
    device instead of the CUDA-only ``torch.cuda.FloatTensor`` constructor, and all noise
    is drawn up front (``infer_from_noise`` takes it explicitly for deterministic parity);
  * ``spect_scaling=True`` references parameters the reference never creates
    (glow.py:233-235 vs :315-316) -> ``NotImplementedError`` here;
  * ``remove_weightnorm`` works (the reference's raises AttributeError, glow.py:358).
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _cache, _lib

__all__ = ["WaveGlow", "Invertible1x1Conv", "WN"]


class _Conv1dParams(nn.Module):
    """Parameter holder with nn.Conv1d's names/shapes/default init (no forward)."""

    def __init__(self, in_ch, out_ch, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_ch * k)
            self.bias = nn.Parameter(torch.empty(out_ch).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class _WeightNormConv1d(nn.Module):
    """``nn.utils.weight_norm(nn.Conv1d(...))`` as stored in checkpoints: weight_g, weight_v, bias."""

    def __init__(self, in_ch, out_ch, k):
        super().__init__()
        plain = _Conv1dParams(in_ch, out_ch, k)
        v = plain.weight.data
        self.bias = plain.bias
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(out_ch, 1, 1).clone())
        self.weight_v = nn.Parameter(v.clone())

    def remove_weight_norm(self):
        if hasattr(self, "weight_v") and self.weight_v is not None:
            v, g = self.weight_v.data, self.weight_g.data
            w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
            del self._parameters["weight_g"], self._parameters["weight_v"]
            self.weight = nn.Parameter(w)


class Invertible1x1Conv(nn.Module):
    """Parameter holder for glow.py:65-83 (random orthonormal init, det > 0)."""

    def __init__(self, c):
        super().__init__()
        self.conv = _Conv1dParams(c, c, 1, bias=False)
        W = torch.linalg.qr(torch.randn(c, c))[0]
        if torch.det(W) < 0:
            W[:, 0] = -1 * W[:, 0]
        self.conv.weight.data = W.view(c, c, 1).contiguous()


class WN(nn.Module):
    """Parameter holder for glow.py:110-186."""

    def __init__(self, n_in_channels, n_mel_channels, n_layers, n_channels, kernel_size,
                 speaker_embed_dim, rezero):
        super().__init__()
        assert kernel_size % 2 == 1
        assert n_channels % 2 == 0
        self.n_layers = n_layers
        self.n_channels = n_channels
        self.speaker_embed_dim = speaker_embed_dim
        if rezero:                                                       # glow.py:127-128, 181-183
            self.alpha_i = nn.ParameterList([nn.Parameter(torch.rand(1) * 0.02 + 0.09) for _ in range(n_layers)])
        if speaker_embed_dim:                                            # glow.py:130-133
            self.speaker_embed = nn.Embedding(_lib.N_SPEAKERS, speaker_embed_dim)
            self.speaker_embed.weight.data.mul_(0.05)
        hidden = 256  # glow.py:153
        self.start = _WeightNormConv1d(n_in_channels, n_channels, 1)
        self.end = _Conv1dParams(n_channels, 2 * n_in_channels, 1)
        self.end.weight.data.zero_()
        self.end.bias.data.zero_()
        self.cond_layers = nn.ModuleList([
            _WeightNormConv1d(n_mel_channels + speaker_embed_dim, hidden, 1),
            _WeightNormConv1d(hidden, hidden, 1),
            _WeightNormConv1d(hidden, 2 * n_channels * n_layers, 1)])
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        for i in range(n_layers):
            self.in_layers.append(_WeightNormConv1d(n_channels, 2 * n_channels, kernel_size))
            rs = 2 * n_channels if i < n_layers - 1 else n_channels
            self.res_skip_layers.append(_WeightNormConv1d(n_channels, rs, 1))


class WaveGlow(nn.Module):
    def __init__(self, yoyo, yoyo_WN, n_mel_channels, n_flows, n_group, n_early_every, n_early_size,
                 memory_efficient, spect_scaling, upsample_mode, WN_config, win_length, hop_length):
        super().__init__()
        if upsample_mode not in ('normal', 'simple', 'simple_half'):
            raise ValueError(f"upsample_mode = {upsample_mode} invalid")            # glow.py:241
        if memory_efficient:
            raise NotImplementedError("memory_efficient builds no layers in the reference (glow.py:263)")
        assert n_group % 2 == 0
        assert hop_length % n_group == 0 and win_length % hop_length == 0

        # What glow.py:226-265 accepts and the HIP path does not build is refused HERE, where the user can see it (the
        # library's own plan check, waveglow_api.hip make_plan, restates these and would otherwise speak at the first infer)
        def need(cond, what):
            if not cond:
                raise NotImplementedError(f"cookietts_amd.WaveGlow (glow.py topology): {what} is not built on the HIP path")
        wn = WN_config
        need(n_group in (4, 8, 12, 16), f"n_group={n_group} (4, 8, 12 or 16: the un-squeeze writes whole float4s, the flow-boundary "
             "kernels hold <= 8 half-channels)")
        need((hop_length // n_group) % 4 == 0, f"hop_length / n_group = {hop_length // n_group} (a multiple of 4: latent rows of any "
             "number of frames are then whole float4s)")
        need(wn['kernel_size'] == 3, f"WN kernel_size={wn['kernel_size']} (3: the dilated in-layers are three K segments of one GEMM)")
        need(wn['n_channels'] >= 128 and wn['n_channels'] % 128 == 0, f"WN n_channels={wn['n_channels']} (a multiple of 128: the GEMM's m-block)")
        need(1 <= wn['n_layers'] <= 12, f"WN n_layers={wn['n_layers']} (1..12: dilation 2^11 is the largest halo)")
        need((n_mel_channels * n_group) % 16 == 0, f"n_mel_channels * n_group = {n_mel_channels * n_group} (a multiple of 16: one K chunk)")
        need(not wn['speaker_embed_dim'] or (n_mel_channels * n_group) % 32 == 0,
             "a speaker embedding with n_mel_channels * n_group not a multiple of 32")
        self.spect_scaling = spect_scaling
        self.multispeaker = WN_config['speaker_embed_dim'] > 0
        self.n_mel_channels = n_mel_channels
        self.n_flows = n_flows
        self.n_group = n_group
        self.n_early_every = n_early_every
        self.n_early_size = n_early_size
        self.win_length = win_length
        self.hop_length = hop_length
        self.WN_config = dict(WN_config)

        # nn.ConvTranspose1d(n_mel, n_mel, win, stride=hop, groups): weight [in, out/groups, win] (glow.py:238-241).
        # 'simple' = one filter per mel channel; 'simple_half' = pairs of channels (the reference passes the float
        # n_mel/2 as `groups`, which nn.ConvTranspose1d rejects under torch 2.x; the integer it stands for is used here)
        self.upsample_mode = upsample_mode
        self.upsample_groups = {'normal': 1, 'simple': n_mel_channels, 'simple_half': n_mel_channels // 2}[upsample_mode]
        assert n_mel_channels % self.upsample_groups == 0
        self.upsample = nn.Module()
        opg = n_mel_channels // self.upsample_groups
        w = torch.empty(n_mel_channels, opg, win_length)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(opg * win_length)
        self.upsample.weight = nn.Parameter(w)
        self.upsample.bias = nn.Parameter(torch.empty(n_mel_channels).uniform_(-bound, bound))

        self.WN = nn.ModuleList()
        self.convinv = nn.ModuleList()
        n_half = n_group // 2
        n_remaining_channels = n_group
        for k in range(n_flows):
            if k % n_early_every == 0 and k > 0:
                n_half = n_half - n_early_size // 2
                n_remaining_channels = n_remaining_channels - n_early_size
            self.convinv.append(Invertible1x1Conv(n_remaining_channels))
            self.WN.append(WN(n_half, n_mel_channels * n_group, **WN_config))
        self.n_remaining_channels = n_remaining_channels

        self._packed = None          # (device, fp32 blob, bf16 blob or None, param key)
        _cache.hook_invalidate(self)
        self._compute_dtype = torch.float32
        self._f32_gemm_mode = None   # None: the library default; see set_f32_gemm_mode
        self._workspaces = {}        # (device, B, F) -> zero-initialised workspace tensor

    # ------------------------------------------------------------------ plumbing ----
    def c_config(self):
        wn = self.WN_config
        return _lib.WaveGlowConfig(
            n_mel_channels=self.n_mel_channels, n_group=self.n_group, n_flows=self.n_flows,
            n_early_every=self.n_early_every, n_early_size=self.n_early_size,
            win_length=self.win_length, hop_length=self.hop_length, n_layers=wn['n_layers'],
            n_channels=wn['n_channels'], kernel_size=wn['kernel_size'], cond_hidden=256,
            speaker_embed_dim=wn.get('speaker_embed_dim', 0), f32_gemm_mode=_lib.model_gemm_mode(self._f32_gemm_mode))

    def set_f32_gemm_mode(self, mode):
        """Main loop of THIS model's fp32 GEMMs: ``"f32"`` (fp32 MFMA), ``"bf16x3"`` (split bf16: fp32 tensors, three
        bf16 MFMA products per operand pair, 16 mantissa bits per operand), ``"bf16x6"`` (three-way split = all 24
        mantissa bits, the six products >= 2^-16: fp32-grade error at 6/16 of the fp32 matrix-pipe cost) or ``None`` /
        ``"default"`` (fp32 MFMA).  Travels in the config struct: other models are not affected; there is no process-wide
        default (removed with ABI 6)."""
        _lib.model_gemm_mode(mode)
        self._f32_gemm_mode = mode
        return self

    def _invalidate(self):
        self._packed = None
        self._workspaces = {}
        for m in self.convinv:
            if hasattr(m, 'W_inverse'):
                del m.W_inverse

    def _apply(self, fn, *a, **kw):
        self._invalidate()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self._invalidate()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def set_compute_dtype(self, dtype):
        """torch.float32 (default: exact fp32 MFMA path), torch.bfloat16 (BASELINE config 3: WN GEMMs on
        bf16 MFMA with fp32 accumulation, WN activations stored bf16; parameters stay fp32 masters and are
        rounded to bf16 once, after weight-norm folding; ``model.bfloat16()`` selects bf16 as well), or the string
        ``"bf16x3"``: split bf16 - weights and activations carried as hi + lo bf16 pairs (16 mantissa bits), every
        contraction as three bf16 MFMA products with fp32 accumulation (see ``ctts_waveglow_infer_spk_bf16x3``); or
        torch.float16: the bf16 path's layouts and kernels with IEEE-half storage and ``v_mfma_f32_32x32x16_f16`` - the
        reference's own reduced-precision mode (glow.py:343), same speed as bf16, ~8x closer to the fp32 reference
        (inside the 1e-3 waveform bound).  Explicit only: ``model.half()`` keeps computing in fp32 MFMA on the fp16-rounded
        weights."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16, "bf16x3"):
            raise NotImplementedError(f"compute dtype {dtype} is not built (float32, bfloat16, float16 or 'bf16x3')")
        if dtype != torch.float32 and self.n_group > 8:
            raise NotImplementedError(f"reduced-precision compute with n_group={self.n_group}: the bf16 / f16 flow-boundary kernels hold "
                                      "<= 8 channels (fp32 takes 16)")
        self._compute_dtype = dtype
        self._invalidate()
        return self

    def _use_bf16(self):
        """0 = fp32 MFMA, 1 = bf16, 3 = split bf16 (the number of bf16 products per contraction), 16 = IEEE half."""
        if self._compute_dtype == "bf16x3":
            return 3
        if self._compute_dtype == torch.float16:
            return 16
        return 1 if (self._compute_dtype == torch.bfloat16 or next(self.parameters()).dtype == torch.bfloat16) else 0

    def repack(self):
        """Call after modifying parameters in place; the next infer re-ingests the weights."""
        self._invalidate()

    @staticmethod
    def remove_weightnorm(model):
        for wn in model.WN:
            wn.start.remove_weight_norm()
            for layer in list(wn.in_layers) + list(wn.cond_layers) + list(wn.res_skip_layers):
                layer.remove_weight_norm()
        model._invalidate()
        return model

    def _dense_weight(self, layer, stream, keep):
        """fp32 contiguous folded conv weight on the model's device (fold runs in the HIP library)."""
        lib = _lib.lib()
        if getattr(layer, 'weight_v', None) is not None:
            v = layer.weight_v.detach().float().contiguous()
            g = layer.weight_g.detach().float().contiguous()
            w = torch.empty_like(v)
            _lib.check(lib.ctts_fold_weightnorm_f32(_lib.ptr(v), _lib.ptr(g), _lib.ptr(w), v.shape[0],
                                                   v[0].numel(), stream), "ctts_fold_weightnorm_f32")
            keep += [v, g, w]
            return w
        w = layer.weight.detach().float().contiguous()
        keep.append(w)
        return w

    @staticmethod
    def _scaled(t, alpha, stream, keep):
        """alpha * t on the device (alpha: 1-element parameter, read by the kernel), through ctts_scale_add_rows_f32."""
        a = alpha.detach().float().contiguous()
        y = torch.empty_like(t)
        rows, cols = (t.shape[0], t[0].numel()) if t.dim() > 1 else (1, t.numel())
        _lib.check(_lib.lib().ctts_scale_add_rows_f32(_lib.ptr(t), _lib.ptr(a), None, _lib.ptr(y), 1, rows, cols, cols, 0,
                                                      stream), "ctts_scale_add_rows_f32")
        keep += [a, y, t]
        return y

    def _ensure_packed(self, device):
        key = _cache.param_key(self)
        if self._packed is not None and self._packed[0] == device and self._packed[3] == key:
            return self._packed[1], self._packed[2]
        if self._packed is not None:
            self._invalidate()
        if device.type != 'cuda':
            raise _lib.HipLibraryError("WaveGlow HIP path needs the model on a GPU (no CPU fallback)")
        use_bf16 = self._use_bf16()
        lib = _lib.lib()
        for p in self.parameters():
            if p.device != device:
                raise RuntimeError(f"parameter on {p.device}, input on {device}: move the model first")
        cfg = self.c_config()
        nbytes = lib.ctts_waveglow_packed_bytes(C.byref(cfg))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported WaveGlow config: " + lib.ctts_last_error().decode())
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            blob = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            bblob = None
            if use_bf16:
                nb = (lib.ctts_waveglow_packed_bf16x3_bytes if use_bf16 == 3 else lib.ctts_waveglow_packed_bf16_bytes)(C.byref(cfg))
                if nb == 0:
                    raise _lib.HipLibraryError("unsupported bf16 WaveGlow config: " + lib.ctts_last_error().decode())
                bblob = torch.zeros(nb // 2, dtype=torch.int16, device=device)
            keep = []
            up_w = self.upsample.weight.detach().float().contiguous()
            if self.upsample_groups > 1:
                # grouped transposed conv = the dense one with zeros outside each group's block: input channel i feeds
                # outputs [g*opg, (g+1)*opg) of its group g = i // (n_mel / groups); x * 0 adds exactly 0
                n_mel, opg = self.n_mel_channels, self.n_mel_channels // self.upsample_groups
                ipg = n_mel // self.upsample_groups
                dense = torch.zeros(n_mel, n_mel, self.win_length, dtype=torch.float32, device=device)
                for gi in range(self.upsample_groups):
                    dense[gi * ipg:(gi + 1) * ipg, gi * opg:(gi + 1) * opg] = up_w[gi * ipg:(gi + 1) * ipg]
                up_w = dense
            up_b = self.upsample.bias.detach().float().contiguous()
            keep += [up_w, up_b]
            _lib.check(lib.ctts_waveglow_pack_upsample(C.byref(cfg), _lib.ptr(up_w), _lib.ptr(up_b),
                                                      _lib.ptr(blob), stream), "ctts_waveglow_pack_upsample")
            n_layers = self.WN_config['n_layers']
            for k in range(self.n_flows):
                wn = self.WN[k]
                fw = _lib.WaveGlowFlowWeights()

                def dev(t):
                    t = t.detach().float().contiguous()
                    keep.append(t)
                    return t.data_ptr()

                fw.start_w = self._dense_weight(wn.start, stream, keep).data_ptr()
                fw.start_b = dev(wn.start.bias)
                for j in range(3):
                    fw.cond_w[j] = self._dense_weight(wn.cond_layers[j], stream, keep).data_ptr()
                    fw.cond_b[j] = dev(wn.cond_layers[j].bias)
                arrs = {}
                for name, layers in (("in", wn.in_layers), ("rs", wn.res_skip_layers)):
                    wa = (C.c_void_p * n_layers)()
                    ba = (C.c_void_p * n_layers)()
                    for i in range(n_layers):
                        wt = self._dense_weight(layers[i], stream, keep)
                        bt = layers[i].bias.detach().float().contiguous()
                        if name == "rs" and hasattr(wn, 'alpha_i'):
                            # ReZero (glow.py:211-212): res_skip(acts) * alpha_i == (alpha*W) acts + alpha*b, folded once
                            wt, bt = self._scaled(wt, wn.alpha_i[i], stream, keep), self._scaled(bt, wn.alpha_i[i], stream, keep)
                        keep.append(bt)
                        wa[i], ba[i] = wt.data_ptr(), bt.data_ptr()
                    arrs[name] = (wa, ba)
                fw.in_w, fw.in_b = arrs["in"]
                fw.rs_w, fw.rs_b = arrs["rs"]
                fw.end_w = dev(wn.end.weight)
                fw.end_b = dev(wn.end.bias)
                # glow.py:90-99: W.float().inverse(), cached on the module as W_inverse
                W = self.convinv[k].conv.weight.detach().squeeze(-1)
                W_inverse = W.float().cpu().inverse().to(device).contiguous()   # host fp32 inverse
                self.convinv[k].W_inverse = W_inverse[..., None]
                keep.append(W_inverse)
                fw.w_inverse = W_inverse.data_ptr()
                if wn.speaker_embed_dim:
                    fw.speaker_embed = dev(wn.speaker_embed.weight)
                _lib.check(lib.ctts_waveglow_pack_flow(C.byref(cfg), k, C.byref(fw), _lib.ptr(blob), stream),
                           f"ctts_waveglow_pack_flow({k})")
                if bblob is not None:
                    pack16 = {1: lib.ctts_waveglow_pack_flow_bf16, 3: lib.ctts_waveglow_pack_flow_bf16x3,
                          16: lib.ctts_waveglow_pack_flow_f16}[use_bf16]
                    _lib.check(pack16(C.byref(cfg), k, C.byref(fw), _lib.ptr(bblob), stream),
                               f"ctts_waveglow_pack_flow_bf16({k})")
            torch.cuda.current_stream(device).synchronize()   # dense temporaries may now be freed
        self._packed = (device, blob, bblob, key)
        return blob, bblob

    def _workspace(self, device, B, F, bf16=False):
        key = (device, B, F, bf16)
        ws = self._workspaces.get(key)
        if ws is None:
            lib = _lib.lib()
            cfg = self.c_config()
            query = {0: lib.ctts_waveglow_workspace_bytes, 1: lib.ctts_waveglow_workspace_bf16_bytes,
                     3: lib.ctts_waveglow_workspace_bf16x3_bytes, 16: lib.ctts_waveglow_workspace_bf16_bytes}[int(bf16)]
            nbytes = query(C.byref(cfg), B, F)
            if nbytes == 0:
                raise _lib.HipLibraryError("workspace query failed: " + lib.ctts_last_error().decode())
            self._workspaces.clear()     # one live geometry at a time
            ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            self._workspaces[key] = ws
        return ws

    def steps_for(self, frames):
        return frames * self.hop_length // self.n_group

    # --------------------------------------------------------------------- the path ----
    def infer_from_noise(self, spect, z_scaled, speaker_id=None):
        """Deterministic entry: ``z_scaled`` [B, n_group, L] already multiplied by sigma.

        Rows: the last ``n_remaining_channels`` are the initial latent (glow.py:326), the
        rows above are the early-output noise in prepend order (glow.py:342-347).
        """
        if self.spect_scaling:
            raise NotImplementedError("spect_scaling=True is broken in the reference (glow.py:233-235)")
        if spect.dim() == 2:
            spect = spect.unsqueeze(0)
        device = spect.device
        blob, bblob = self._ensure_packed(device)
        lib = _lib.lib()
        B, M, F = spect.shape
        assert M == self.n_mel_channels, (M, self.n_mel_channels)
        L = self.steps_for(F)
        assert tuple(z_scaled.shape) == (B, self.n_group, L), (tuple(z_scaled.shape), (B, self.n_group, L))
        mel = spect.detach().float().contiguous()
        z = z_scaled.detach().to(device=device, dtype=torch.float32).contiguous()
        ids = None
        if self.multispeaker:
            if speaker_id is None:      # the reference would feed cond_layers[0] too few channels and crash (glow.py:193-198)
                raise RuntimeError("this WaveGlow is multispeaker (speaker_embed_dim > 0): pass speaker_id")
            # range check like nn.Embedding's, where it costs no device sync (host ids); ids already on the device are
            # checked by the kernel, which turns an out-of-range id into a NaN utterance instead of an out-of-bounds read
            if not speaker_id.is_cuda and speaker_id.numel() and \
                    (int(speaker_id.min()) < 0 or int(speaker_id.max()) >= _lib.N_SPEAKERS):
                raise IndexError("speaker id out of range of the embedding table")
            ids = speaker_id.detach().to(device=device, dtype=torch.int64).reshape(-1).contiguous()
            assert ids.shape[0] == B, (tuple(ids.shape), B)
        mode = self._use_bf16()
        ws = self._workspace(device, B, F, bf16=mode)
        wave = torch.empty(B, L * self.n_group, dtype=torch.float32, device=device)
        cfg = self.c_config()
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            if bblob is not None:
                name16 = {1: "ctts_waveglow_infer_spk_bf16", 3: "ctts_waveglow_infer_spk_bf16x3", 16: "ctts_waveglow_infer_spk_f16"}[mode]
                _lib.check(getattr(lib, name16)(C.byref(cfg), _lib.ptr(blob), _lib.ptr(bblob), _lib.ptr(mel), _lib.ptr(z),
                                                _lib.ptr(ids), _lib.ptr(wave), B, F, _lib.ptr(ws), ws.numel() * 4, stream), name16)
            else:
                _lib.check(lib.ctts_waveglow_infer_spk_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(mel), _lib.ptr(z),
                                                          _lib.ptr(ids), _lib.ptr(wave), B, F, _lib.ptr(ws),
                                                          ws.numel() * 4, stream), "ctts_waveglow_infer_spk_f32")
        return wave.to(spect.dtype)

    def infer(self, spect, speaker_id=None, sigma=1.0):
        """``glow.WaveGlow.infer``: spect [B, n_mel, F] -> audio [B, F*hop] on spect's device."""
        if spect.dim() == 2:
            spect = spect.unsqueeze(0)
        B, _, F = spect.shape
        z = torch.randn(B, self.n_group, self.steps_for(F), device=spect.device, dtype=torch.float32)
        return self.infer_from_noise(spect, z * sigma, speaker_id=speaker_id)

    def forward(self, spect, audio=None, speaker_id=None):
        raise NotImplementedError("training direction (glow.py:267-312) is outside the inference hot path")
