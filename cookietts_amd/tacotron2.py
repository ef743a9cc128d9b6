"""Tacotron2-TM (text -> mel) with the decoder loop on the MI355X HIP path (BASELINE config 5).

Host-side mirror of ``/root/reference/CookieTTS/_2_ttm/tacotron2_tm/model.py``: ``Tacotron2(hparams)``
(:920-957), ``Tacotron2.inference`` (:1044-1080), ``Decoder.inference`` (:851-916), ``load_model`` (:22-33),
with the reference's module tree so ``state_dict`` keys/shapes are identical and reference checkpoints
(``checkpoint['state_dict']``, train.py:255-279) load unchanged.

What runs where:
  * hot loop - ``Decoder.inference``: memory bottleneck + processed memory (T4), then per step prenet (T5),
    attention LSTM, location-sensitive windowed attention, two decoder LSTMs, gate/mel projection (T6-T8):
    hand-written HIP behind ``ctts_taco_decoder_*`` (``csrc/tacotron_decoder.hip``).  The stop rule (T9,
    model.py:898-904) is evaluated on the host every ``STOP_CHECK_EVERY`` steps instead of after every
    step (the reference syncs the device each step); the result is trimmed to the exact stop step, so
    outputs are identical.
  * one-shot stages - embedding, encoder (T2), memory assembly (T3), postnet (T10): NOT yet native HIP;
    they run as PyTorch-ROCm library ops on the GPU (marked "next-tier" in SURVEY.md §2.2, <1 % of the
    time).  No CPU fallback: CPU tensors raise.

Reference quirks handled (SURVEY.md §8a "Quirks"): prenet dropout is ALWAYS on (model.py:189-190) - masks
are drawn on the device per call, or passed explicitly (``keep_masks``) for deterministic parity;
``gt_sylps or pred_sylps`` (:1058) is evaluated as "use gt_sylps if given"; ``MaskedBatchNorm1d``'s hidden
call counter (untts/model.py:326,333-336) is not reproduced (plain eval-mode batch norm).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["Tacotron2", "Decoder", "load_model", "stop_step"]

STOP_CHECK_EVERY = 32
drop_rate = 0.5


class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x):
        return self.linear_layer(x)


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain='linear'):
        super().__init__()
        if padding is None:
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x):
        return self.conv(x)


class LocationLayer(nn.Module):
    def __init__(self, n_filters, kernel_size, attention_dim):
        super().__init__()
        self.location_conv = ConvNorm(2, n_filters, kernel_size=kernel_size, padding=int((kernel_size - 1) / 2),
                                      bias=False)
        self.location_dense = LinearNorm(n_filters, attention_dim, bias=False, w_init_gain='tanh')


class Attention(nn.Module):
    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, n_filters, kernel_size, window_range,
                 pos_learned, pos_offset):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(n_filters, kernel_size, attention_dim)
        self.windowed_attention_range = window_range
        if pos_learned is True:
            self.windowed_att_pos_offset = nn.Parameter(torch.zeros(1))
        else:
            self.windowed_att_pos_offset = pos_offset


class Prenet(nn.Module):
    def __init__(self, in_dim, sizes, p_prenet_dropout):
        super().__init__()
        in_sizes = [in_dim] + sizes[:-1]
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip(in_sizes, sizes)])
        self.p_prenet_dropout = p_prenet_dropout


class MemoryBottleneck(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        in_dim = hparams.encoder_LSTM_dim + hparams.speaker_embedding_dim + hparams.torchMoji_crushedDim + 1
        self.bottleneck = LinearNorm(in_dim, hparams.memory_bottleneck_dim, bias=hparams.memory_bottleneck_bias,
                                     w_init_gain='tanh')


def stop_step(gate_logits, gate_threshold, gate_delay, max_decoder_steps, state=None):
    """The reference's stop rule (model.py:879-904) over a [B, n] block of gate logits (host tensors).
    ``state`` = (sig_max [B], break_point, first_step) carried between blocks.  Returns (n_total or None, state)."""
    B, n = gate_logits.shape
    sig_max, break_point, i0 = state if state is not None else (torch.zeros(B), max_decoder_steps, 0)
    for j in range(n):
        i = i0 + j
        if i > 4:
            sig_max = torch.max(torch.sigmoid(gate_logits[:, j].float()), sig_max)
        if sig_max.min() > gate_threshold:
            break_point = min(break_point, i + gate_delay)
        if i >= break_point:
            return i + 1, (sig_max, break_point, i + 1)
    return None, (sig_max, break_point, i0 + n)


class Decoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        hp = hparams

        def need(cond, what):
            if not cond:
                raise NotImplementedError(f"Tacotron2 decoder option not built on the HIP path yet: {what}")
        need(hp.attention_type == 0, "attention_type != 0")
        need(hp.use_memory_bottleneck and not hp.memory_bottleneck_bias, "use_memory_bottleneck=False / bias")
        need(hp.AttRNN_extra_decoder_input, "AttRNN_extra_decoder_input=False")
        need(hp.second_decoder_rnn_dim > 0 and hp.second_decoder_residual_connection
             and hp.second_decoder_rnn_dim == hp.decoder_rnn_dim, "second decoder RNN without residual")
        need(not hp.decoder_residual_connection, "decoder_residual_connection")
        need(hp.prenet_layers == 2 and not getattr(hp, 'prenet_batchnorm', False) and hp.p_prenet_dropout == 0.5
             and not getattr(hp, 'prenet_speaker_embed_dim', 0), "prenet other than 2 layers / p=0.5 / no BN")
        need(getattr(hp, 'windowed_attention_range', 0) > 0, "windowed_attention_range=0")
        need(not hp.attention_learned_temperature and not getattr(hp, 'use_cum_attention_scaler', False),
             "learned temperature / cum attention scaler")
        need(hp.n_frames_per_step == 1 and hp.context_frames == 1 and not hp.hide_startstop_tokens,
             "n_frames_per_step/context_frames != 1, hide_startstop_tokens")
        self.n_mel_channels = hp.n_mel_channels
        self.memory_dim = hp.memory_bottleneck_dim
        self.attention_rnn_dim = hp.attention_rnn_dim
        self.decoder_rnn_dim = hp.decoder_rnn_dim
        self.second_decoder_rnn_dim = hp.second_decoder_rnn_dim
        self.prenet_dim = hp.prenet_dim
        self.max_decoder_steps = hp.max_decoder_steps
        self.gate_threshold = hp.gate_threshold
        self.gate_delay = 0                                   # model.py:432 (the server overrides it)
        self.windowed_attention_range = hp.windowed_attention_range
        self.exp_smoothing_factor = nn.Parameter(torch.ones(1) * 0.0)
        self.memory_bottleneck = MemoryBottleneck(hp)
        self.prenet = Prenet(hp.n_mel_channels, [hp.prenet_dim] * hp.prenet_layers, hp.p_prenet_dropout)
        self.attention_rnn = nn.LSTMCell(hp.prenet_dim + self.memory_dim + hp.decoder_rnn_dim, hp.attention_rnn_dim)
        self.attention_layer = Attention(hp.attention_rnn_dim, self.memory_dim, hp.attention_dim,
                                         hp.attention_location_n_filters, hp.attention_location_kernel_size,
                                         hp.windowed_attention_range, hp.windowed_att_pos_learned,
                                         hp.windowed_att_pos_offset)
        self.decoder_rnn = nn.LSTMCell(hp.attention_rnn_dim + self.memory_dim, hp.decoder_rnn_dim)
        self.second_decoder_rnn = nn.LSTMCell(hp.decoder_rnn_dim, hp.second_decoder_rnn_dim)
        self.linear_projection = LinearNorm(hp.second_decoder_rnn_dim + self.memory_dim, hp.n_mel_channels)
        self.gate_layer = LinearNorm(hp.second_decoder_rnn_dim + self.memory_dim, 1, bias=True, w_init_gain='sigmoid')
        self._attn_cfg = (hp.attention_dim, hp.attention_location_n_filters, hp.attention_location_kernel_size)
        self._memory_in_dim = hp.encoder_LSTM_dim + hp.speaker_embedding_dim + hp.torchMoji_crushedDim + 1
        self._packed = None
        self._ws = {}

    # ------------------------------------------------------------------ plumbing ----
    def c_config(self):
        A, Fl, K = self._attn_cfg
        return _lib.TacoDecoderConfig(
            n_mel_channels=self.n_mel_channels, memory_in_dim=self._memory_in_dim, memory_dim=self.memory_dim,
            attention_dim=A, attention_rnn_dim=self.attention_rnn_dim, decoder_rnn_dim=self.decoder_rnn_dim,
            second_decoder_rnn_dim=self.second_decoder_rnn_dim, prenet_dim=self.prenet_dim, location_n_filters=Fl,
            location_kernel_size=K, window_range=self.windowed_attention_range)

    def _invalidate(self):
        self._packed, self._ws = None, {}

    def _apply(self, fn, *a, **kw):
        self._invalidate()
        return super()._apply(fn, *a, **kw)

    def _ensure_packed(self, device):
        if self._packed is not None and self._packed[0] == device:
            return self._packed[1]
        if device.type != 'cuda':
            raise _lib.HipLibraryError("Tacotron2 decoder HIP path needs the model on a GPU (no CPU fallback)")
        lib = _lib.lib()
        cfg = self.c_config()
        nbytes = lib.ctts_taco_decoder_packed_bytes(C.byref(cfg))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported decoder config: " + lib.ctts_last_error().decode())
        keep = []

        def dev(t):
            t = t.detach().float().contiguous()
            if t.device != device:
                raise RuntimeError(f"parameter on {t.device}, input on {device}: move the model first")
            keep.append(t)
            return t.data_ptr()

        def lstm(cell):
            return _lib.LstmWeights(dev(cell.weight_ih), dev(cell.weight_hh), dev(cell.bias_ih), dev(cell.bias_hh))
        al = self.attention_layer
        w = _lib.TacoDecoderWeights()
        w.bottleneck_w = dev(self.memory_bottleneck.bottleneck.linear_layer.weight)
        w.memory_layer_w = dev(al.memory_layer.linear_layer.weight)
        w.query_w = dev(al.query_layer.linear_layer.weight)
        w.v_w = dev(al.v.linear_layer.weight)
        w.loc_conv_w = dev(al.location_layer.location_conv.conv.weight)
        w.loc_dense_w = dev(al.location_layer.location_dense.linear_layer.weight)
        w.prenet_w1 = dev(self.prenet.layers[0].linear_layer.weight)
        w.prenet_w2 = dev(self.prenet.layers[1].linear_layer.weight)
        w.att_rnn, w.dec_rnn, w.dec2_rnn = lstm(self.attention_rnn), lstm(self.decoder_rnn), lstm(self.second_decoder_rnn)
        w.proj_w = dev(self.linear_projection.linear_layer.weight)
        w.proj_b = dev(self.linear_projection.linear_layer.bias)
        w.gate_w = dev(self.gate_layer.linear_layer.weight)
        w.gate_b = dev(self.gate_layer.linear_layer.bias)
        off = al.windowed_att_pos_offset
        w.windowed_att_pos_offset = float(off.detach().reshape(-1)[0]) if torch.is_tensor(off) else float(off)
        w.exp_smoothing_factor = float(self.exp_smoothing_factor.detach().reshape(-1)[0])
        with torch.cuda.device(device):
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            blob = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            _lib.check(lib.ctts_taco_decoder_pack(C.byref(cfg), C.byref(w), _lib.ptr(blob), stream),
                       "ctts_taco_decoder_pack")
            torch.cuda.current_stream(device).synchronize()
        self._packed = (device, blob)
        return blob

    # --------------------------------------------------------------------- the path ----
    @torch.no_grad()
    def inference(self, memory, memory_lengths, return_hidden_state=False, keep_masks=None, fixed_steps=None):
        """model.py:851-916.  memory [B, txt_T, memory_in_dim], memory_lengths [B].
        Returns (mel [B, n_mel, T], gate (sigmoid) [B, T], alignments [B, T, txt_T], None).
        ``keep_masks`` [>=T, 2, B, prenet_dim] uint8 overrides the random prenet dropout;
        ``fixed_steps`` runs exactly that many steps with the stop rule disabled (benchmarks)."""
        if return_hidden_state:
            raise NotImplementedError("return_hidden_state is not built")
        device = memory.device
        blob = self._ensure_packed(device)
        lib = _lib.lib()
        cfg = self.c_config()
        B, T, D = memory.shape
        assert D == self._memory_in_dim, (D, self._memory_in_dim)
        max_steps = int(fixed_steps) if fixed_steps is not None else int(self.max_decoder_steps)
        mem = memory.detach().float().contiguous()
        lens = memory_lengths.detach().to(device=device, dtype=torch.int32).contiguous()
        if keep_masks is None:
            keep_masks = (torch.rand(max_steps, 2, B, self.prenet_dim, device=device) < 0.5).to(torch.uint8)
        else:
            keep_masks = torch.as_tensor(keep_masks).to(device=device, dtype=torch.uint8).contiguous()
            assert keep_masks.shape[0] >= max_steps and tuple(keep_masks.shape[1:]) == (2, B, self.prenet_dim)
            keep_masks = keep_masks[:max_steps].contiguous()
        key = (device, B, T)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib.ctts_taco_decoder_workspace_bytes(C.byref(cfg), B, T)
            if nbytes == 0:
                raise _lib.HipLibraryError("decoder workspace query failed: " + lib.ctts_last_error().decode())
            self._ws = {key: torch.empty(nbytes // 4, dtype=torch.float32, device=device)}
            ws = self._ws[key]
        mel = torch.zeros(B, self.n_mel_channels, max_steps, dtype=torch.float32, device=device)
        gate = torch.zeros(B, max_steps, dtype=torch.float32, device=device)
        align = torch.zeros(B, max_steps, T, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream_obj = torch.cuda.current_stream(device)
            stream = C.c_void_p(stream_obj.cuda_stream)
            _lib.check(lib.ctts_taco_decoder_init_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(mem), _lib.ptr(lens), B, T,
                                                     _lib.ptr(ws), ws.numel() * 4, stream), "ctts_taco_decoder_init_f32")
            done, n_total, state = 0, None, None
            while done < max_steps and n_total is None:
                n = min(STOP_CHECK_EVERY if fixed_steps is None else max_steps, max_steps - done)
                _lib.check(lib.ctts_taco_decoder_steps_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(keep_masks),
                                                          _lib.ptr(mel), _lib.ptr(gate), _lib.ptr(align), B, T, done, n,
                                                          max_steps, _lib.ptr(ws), stream), "ctts_taco_decoder_steps_f32")
                if fixed_steps is None:
                    n_total, state = stop_step(gate[:, done:done + n].cpu(), self.gate_threshold, self.gate_delay,
                                               max_steps, state)
                done += n
            if n_total is None:
                n_total = max_steps
                if fixed_steps is None:
                    print("Warning! Reached max decoder steps")
        return (mel[:, :, :n_total].contiguous(), torch.sigmoid(gate[:, :n_total]), align[:, :n_total].contiguous(), None)


class Postnet(nn.Module):
    """model.py:196-228 (one-shot stage, PyTorch-ROCm library ops for now)."""

    def __init__(self, hparams):
        super().__init__()
        self.b_res = getattr(hparams, 'postnet_residual_connections', False)
        self.convolutions = nn.ModuleList()
        prev_output_layer = True
        n = hparams.postnet_n_convolutions
        for i in range(n):
            is_out = (bool(self.b_res) and bool(i % self.b_res == 0)) or (i + 1 == n)
            layers = [ConvNorm(hparams.n_mel_channels if prev_output_layer else hparams.postnet_embedding_dim,
                               hparams.n_mel_channels if is_out else hparams.postnet_embedding_dim,
                               kernel_size=hparams.postnet_kernel_size, stride=1,
                               padding=int((hparams.postnet_kernel_size - 1) / 2), dilation=1,
                               w_init_gain='linear' if is_out else 'tanh')]
            if not is_out:
                layers.append(nn.BatchNorm1d(hparams.postnet_embedding_dim))
            prev_output_layer = is_out
            self.convolutions.append(nn.Sequential(*layers))

    def forward(self, x):
        x_orig = x.clone()
        n = len(self.convolutions)
        for i, conv in enumerate(self.convolutions):
            if (bool(self.b_res) and bool(i % self.b_res == 0)) or (i + 1 == n):
                x_orig = x_orig + conv(x)
                x = x_orig
            else:
                x = F.dropout(torch.tanh(conv(x)), drop_rate, self.training)
        return x_orig


class Encoder(nn.Module):
    """model.py:231-316 (one-shot stage, PyTorch-ROCm library ops for now)."""

    def __init__(self, hparams):
        super().__init__()
        hp = hparams
        if hp.encoder_concat_speaker_embed != 'before_conv':
            raise NotImplementedError("encoder_concat_speaker_embed != 'before_conv'")
        self.encoder_speaker_embed_dim = hp.encoder_speaker_embed_dim
        if self.encoder_speaker_embed_dim:
            self.encoder_speaker_embedding = nn.Embedding(hp.n_speakers, self.encoder_speaker_embed_dim)
        convs = []
        for i in range(hp.encoder_n_convolutions):
            in_dim = hp.symbols_embedding_dim + self.encoder_speaker_embed_dim if i == 0 else hp.encoder_conv_hidden_dim
            out_dim = hp.encoder_LSTM_dim if i == hp.encoder_n_convolutions - 1 else hp.encoder_conv_hidden_dim
            convs.append(nn.Sequential(ConvNorm(in_dim, out_dim, kernel_size=hp.encoder_kernel_size, stride=1,
                                                padding=int((hp.encoder_kernel_size - 1) / 2), dilation=1,
                                                w_init_gain='relu'), nn.BatchNorm1d(out_dim)))
        self.convolutions = nn.ModuleList(convs)
        self.lstm = nn.LSTM(hp.encoder_LSTM_dim, int(hp.encoder_LSTM_dim / 2), 1, batch_first=True, bidirectional=True)
        self.LReLU = nn.LeakyReLU(negative_slope=0.01)
        self.sylps_layer = LinearNorm(hp.encoder_LSTM_dim, 1)

    def forward(self, text, text_lengths=None, speaker_ids=None):
        if self.encoder_speaker_embed_dim:
            emb = self.encoder_speaker_embedding(speaker_ids)[:, None].transpose(1, 2)
            text = torch.cat((text, emb.repeat(1, 1, text.size(2))), dim=1)
        for conv in self.convolutions:
            text = F.dropout(self.LReLU(conv(text)), drop_rate, self.training)
        text = text.transpose(1, 2)
        if text_lengths is not None:
            text = nn.utils.rnn.pack_padded_sequence(text, text_lengths.cpu().numpy(), batch_first=True,
                                                     enforce_sorted=False)
        self.lstm.flatten_parameters()
        outputs, (hidden_state, _) = self.lstm(text)
        if text_lengths is not None:
            outputs, _ = nn.utils.rnn.pad_packed_sequence(outputs, batch_first=True)
        hidden_state = hidden_state.transpose(0, 1).contiguous().view(hidden_state.shape[1], -1)
        return outputs, hidden_state, self.sylps_layer(hidden_state)


class SylpsNet(nn.Module):
    """tacotron2_ssvae/nets/SylpsNet.py:7-61 (infer_auto only)."""

    def __init__(self, hparams):
        super().__init__()
        layers = []
        dims = hparams.sylpsnet_layer_dims
        for i, dim in enumerate(dims):
            last = i + 1 == len(dims)
            layers.append(LinearNorm(2 if i == 0 else dim, 1 if last else dim))
            if not last:
                layers.append(nn.LeakyReLU(negative_slope=0.05, inplace=True))
        self.seq_layers = nn.Sequential(*layers)
        self.res_weight = nn.Parameter(torch.tensor(0.01))

    def infer_auto(self, sylps, rand_sampling=False):
        sylps_cat = torch.cat((sylps, sylps.log()), dim=1)
        syl_params = sylps_cat + self.res_weight * self.seq_layers(sylps_cat)
        return syl_params[:, 0][:, None]


class Tacotron2(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        hp = hparams
        self.mask_padding = hp.mask_padding
        self.fp16_run = hp.fp16_run
        self.n_mel_channels = hp.n_mel_channels
        self.n_frames_per_step = hp.n_frames_per_step
        self.embedding = nn.Embedding(hp.n_symbols, hp.symbols_embedding_dim)
        val = np.sqrt(3.0) * np.sqrt(2.0 / (hp.n_symbols + hp.symbols_embedding_dim))
        self.embedding.weight.data.uniform_(-val, val)
        self.speaker_embedding_dim = hp.speaker_embedding_dim
        if self.speaker_embedding_dim:
            self.speaker_embedding = nn.Embedding(hp.n_speakers, self.speaker_embedding_dim)
        self.encoder = Encoder(hp)
        self.decoder = Decoder(hp)
        if getattr(hp, 'use_postnet', True):
            self.postnet = Postnet(hp)
        self.sylps_net = SylpsNet(hp)
        self.tm_linear = nn.Linear(hp.torchMoji_attDim, hp.torchMoji_crushedDim)
        if hp.torchMoji_BatchNorm:
            self.tm_bn = nn.BatchNorm1d(hp.torchMoji_attDim, momentum=0.05)

    @torch.no_grad()
    def inference(self, text_seq, text_lengths, speaker_id, torchmoji_hdn, gt_sylps=None, return_hidden_state=False,
                  keep_masks=None, fixed_steps=None):
        """model.py:1044-1080.  Returns the reference's dict (pred_mel_postnet, pred_gate, alignments, pred_sylps)."""
        if self.training:
            raise RuntimeError("call .eval() first: inference uses eval-mode batch norm / no dropout but the prenet's")
        memory = []
        embedded_text = self.embedding(text_seq).transpose(1, 2)
        encoder_outputs, _, pred_sylps = self.encoder(embedded_text, text_lengths, speaker_ids=speaker_id)
        memory.append(encoder_outputs)
        txt_T = encoder_outputs.size(1)
        memory.append(self.speaker_embedding(speaker_id)[:, None].repeat(1, txt_T, 1))
        sylzu = self.sylps_net.infer_auto(gt_sylps if gt_sylps is not None else pred_sylps, rand_sampling=False)
        memory.append(sylzu[:, None].repeat(1, txt_T, 1))
        tm = self.tm_bn(torchmoji_hdn).to(sylzu) if hasattr(self, 'tm_bn') else torchmoji_hdn
        memory.append(self.tm_linear(tm)[:, None].repeat(1, txt_T, 1))
        memory = torch.cat(memory, dim=2)
        pred_mel, pred_gate, alignments, _ = self.decoder.inference(memory, memory_lengths=text_lengths,
                                                                    keep_masks=keep_masks, fixed_steps=fixed_steps)
        pred_mel_postnet = self.postnet(pred_mel) if hasattr(self, 'postnet') else pred_mel
        return {"pred_mel_postnet": pred_mel_postnet, "pred_gate": pred_gate, "alignments": alignments,
                "pred_sylps": pred_sylps, "pred_mel": pred_mel}


def load_model(hparams):
    """model.py:22-33."""
    model = Tacotron2(hparams)
    if torch.cuda.is_available():
        model = model.cuda()
    if hparams.fp16_run:
        raise NotImplementedError("fp16_run is not built on the HIP path (fp32 only)")
    return model
