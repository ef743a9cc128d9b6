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
  * one-shot stages - embedding, encoder (T2), memory assembly (T3), postnet (T10): composed here, in the
    reference's own order, from operator-level HIP primitives of the same library: ``ctts_conv1d_f32`` (same-padded
    Conv1d with the eval-mode BatchNorm folded in and LeakyReLU/tanh fused, on the fp32 MFMA conv-GEMM),
    ``ctts_lstm_seq_f32`` (packed-sequence LSTM, one direction), ``ctts_taco_embed_f32``, ``ctts_taco_memory_f32``.
    PyTorch only allocates the buffers.  No CPU fallback: CPU tensors raise.

Reference quirks handled (SURVEY.md §8a "Quirks"): prenet dropout is ALWAYS on (model.py:189-190) - masks
are drawn on the device per call, or passed explicitly (``keep_masks``) for deterministic parity;
``gt_sylps or pred_sylps`` (:1058) is evaluated as "use gt_sylps if given"; ``MaskedBatchNorm1d``'s hidden
call counter (untts/model.py:326,333-336) is not reproduced (plain eval-mode batch norm).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cache, _lib

__all__ = ["Tacotron2", "Decoder", "load_model", "stop_step"]

STOP_CHECK_EVERY = 32
PINNED_SLOTS = 4            # pinned verdict slots per inference() call (two blocks are in flight at most)
PERSIST_CTL_WORDS = 8       # the LAST 8 uint64 words (64 bytes) of a persistent-decoder exchange buffer are its control words
# environment knobs, read once at import: CTTS_TACO_NO_PERSIST=1 keeps the six-launches-per-step decoder,
# CTTS_TACO_CHUNK=n overrides the steps per launch block (profiling)
_PERSIST_ON = not os.environ.get("CTTS_TACO_NO_PERSIST")
_CHUNK = int(os.environ.get("CTTS_TACO_CHUNK", 0))
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
    """Host restatement of the block-fed stop rule, kept for the tests that pin ``ctts_taco_stop_rule_f32`` (the
    product path evaluates the rule on the device, see Decoder.inference).
    The reference's stop rule (model.py:879-904) over a [B, n] block of gate logits (host tensors).
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


PERSIST_REPROBE_AFTER = 16     # inference() calls a decoder spends on the per-launch form before it tries the persistent form again


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
        self._xchg = {}
        self.use_persistent = _PERSIST_ON      # False: the six-launches-per-step decoder (CTTS_TACO_NO_PERSIST=1 at import)
        # per-decoder state of the persistent form: "unprobed" (its next launch is checked synchronously), "ok", or
        # "disabled" (a probe found the 256 workgroups not co-resident; re-probed after PERSIST_REPROBE_AFTER calls)
        self._persist = "unprobed"
        self._persist_fallback_calls = 0
        _cache.hook_invalidate(self)

    @property
    def persistent_state(self):
        """Which form of the decoder loop the next ``inference`` call of THIS decoder takes: "off" (``use_persistent``
        False), "unprobed" / "ok" (persistent kernel), or "disabled" (per-launch form after a failed probe)."""
        return self._persist if self.use_persistent else "off"

    def reprobe_persistent(self):
        """Forget a failed probe now (it is forgotten by itself after PERSIST_REPROBE_AFTER calls): the next call launches
        the persistent kernel again and checks it synchronously."""
        self._persist, self._persist_fallback_calls = "unprobed", 0
        self._xchg = {}

    # ------------------------------------------------------------------ plumbing ----
    def c_config(self):
        A, Fl, K = self._attn_cfg
        return _lib.TacoDecoderConfig(
            n_mel_channels=self.n_mel_channels, memory_in_dim=self._memory_in_dim, memory_dim=self.memory_dim,
            attention_dim=A, attention_rnn_dim=self.attention_rnn_dim, decoder_rnn_dim=self.decoder_rnn_dim,
            second_decoder_rnn_dim=self.second_decoder_rnn_dim, prenet_dim=self.prenet_dim, location_n_filters=Fl,
            location_kernel_size=K, window_range=self.windowed_attention_range)

    def _invalidate(self):
        self._packed, self._ws, self._xchg = None, {}, {}

    def _apply(self, fn, *a, **kw):
        self._invalidate()
        return super()._apply(fn, *a, **kw)

    def _ensure_packed(self, device):
        key = _cache.param_key(self)
        if self._packed is not None and self._packed[0] == device and self._packed[2] == key:
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
        self._packed = (device, blob, key)
        return blob

    # --------------------------------------------------------------------- the path ----
    @torch.no_grad()
    def inference(self, memory, memory_lengths, return_hidden_state=False, keep_masks=None, fixed_steps=None):
        """model.py:851-916.  memory [B, txt_T, memory_in_dim], memory_lengths [B].
        Returns (mel [B, n_mel, T], gate (sigmoid) [B, T], alignments [B, T, txt_T], None).
        ``keep_masks`` [>=T, 2, B, prenet_dim] uint8 overrides the random prenet dropout;
        ``fixed_steps`` runs exactly that many steps with the stop rule disabled (benchmarks)."""
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
        # One workspace holds the whole batch where the library builds the batched MFMA form for this shape (up to 256 rows:
        # what the server's simultaneous_texts x batch_size_per_text asks for, text2speech.py:418-424, 537, 554); batches of
        # <= MAX_GROUP take the weight-resident persistent kernel.  Anything larger than the library's limit runs as groups in
        # lockstep (same chunk of steps for every group, then ONE stop-rule evaluation over the whole batch, model.py:898-904)
        per_ws = MAX_GROUP
        if B >= BATCHED_FROM:
            per_ws = max(MAX_GROUP, int(lib.ctts_taco_decoder_max_batch(C.byref(cfg))))
        groups = [(g0, min(g0 + per_ws, B)) for g0 in range(0, B, per_ws)]
        key = (device, B, T)
        wss = self._ws.get(key)
        if wss is None:
            wss = []
            for g0, g1 in groups:
                nbytes = lib.ctts_taco_decoder_workspace_bytes(C.byref(cfg), g1 - g0, T)
                if nbytes == 0:
                    raise _lib.HipLibraryError("decoder workspace query failed: " + lib.ctts_last_error().decode())
                wss.append(torch.empty(nbytes // 4, dtype=torch.float32, device=device))
            self._ws = {key: wss}
        masks = [keep_masks if len(groups) == 1 else keep_masks[:, :, g0:g1].contiguous() for g0, g1 in groups]
        # persistent form (one launch per block of steps, weight-stationary fresh columns, granule all-gathers) where the
        # library builds it for this shape and device (0 bytes otherwise: the six-launches-per-step form).  The exchange
        # buffers are zero-filled once and kept with the workspaces; their last 64 bytes are the sticky control words.
        if self._persist == "disabled" and self.use_persistent:
            self._persist_fallback_calls += 1
            if self._persist_fallback_calls > PERSIST_REPROBE_AFTER:
                self.reprobe_persistent()
        # (return_hidden_state: the per-launch forms record [dec_h + d2_h | context] per step, model.py:762, 888-889; the persistent
        #  kernel has no such output)
        persist = bool(self.use_persistent) and self._persist != "disabled" and not return_hidden_state
        xchg = self._xchg.get(key) if persist else [None] * len(groups)
        if xchg is None:
            xchg = []
            for g0, g1 in groups:
                nb = lib.ctts_taco_decoder_persistent_bytes(C.byref(cfg), g1 - g0, T)
                xchg.append(torch.zeros(nb // 8, dtype=torch.int64, device=device) if nb else None)
            self._xchg = {key: xchg}
        xchg = list(xchg)

        def ctl_words(xb):
            return xb[-PERSIST_CTL_WORDS:].view(torch.int32)[:4]

        def raise_if_aborted(words):
            if int(words[0]) != 0:
                self._xchg = {}       # the control words are sticky: drop the buffers so a retry starts clean
                raise _lib.HipLibraryError(f"persistent decoder gave up waiting (workgroup {int(words[1])}, "
                                           f"phase {int(words[2])}, step {int(words[3])})")

        def check_persistent():
            for xb in xchg:
                if xb is not None:
                    raise_if_aborted(ctl_words(xb).cpu())
        mel = torch.zeros(B, self.n_mel_channels, max_steps, dtype=torch.float32, device=device)
        gate = torch.zeros(B, max_steps, dtype=torch.float32, device=device)
        align = torch.zeros(B, max_steps, T, dtype=torch.float32, device=device)
        hidden = torch.zeros(B, self.second_decoder_rnn_dim + self.memory_dim, max_steps, dtype=torch.float32, device=device) \
            if return_hidden_state else None
        with torch.cuda.device(device):
            stream_obj = torch.cuda.current_stream(device)
            stream = C.c_void_p(stream_obj.cuda_stream)
            for (g0, g1), ws in zip(groups, wss):
                _lib.check(lib.ctts_taco_decoder_init_f32(C.byref(cfg), _lib.ptr(blob), _lib.ptr(mem[g0:g1]), _lib.ptr(lens[g0:g1]),
                                                         g1 - g0, T, _lib.ptr(ws), ws.numel() * 4, stream),
                           "ctts_taco_decoder_init_f32")
            # stop rule (model.py:879-904) on the device: after every block of steps one small kernel folds the block's
            # gate logits into the rule's state and its verdict (n_total or -1) is copied to pinned host memory.  The
            # host looks at block k-1's verdict only after block k is enqueued, so the GPU never waits for the host;
            # steps run past the stop are trimmed below (at most 2 * STOP_CHECK_EVERY of them).
            done, n_total = 0, None
            chunk = _CHUNK or (STOP_CHECK_EVERY if fixed_steps is None else max_steps)
            if fixed_steps is None:
                state = torch.empty(lib.ctts_taco_stop_state_bytes(B) // 4, dtype=torch.float32, device=device)
                _lib.check(lib.ctts_taco_stop_reset(_lib.ptr(state), B, max_steps, stream), "ctts_taco_stop_reset")
                verdict_dev = state[B + 1:B + 2].view(torch.int32)
                # per enqueued block one pinned slot [verdict, ctl word 0 of every group] and one event; two blocks are in
                # flight at most, so a ring of PINNED_SLOTS slots allocated once per call is never overwritten while pending
                pinned = torch.zeros(PINNED_SLOTS, 1 + len(groups), dtype=torch.int32).pin_memory()
                events = [torch.cuda.Event() for _ in range(PINNED_SLOTS)]
                pending, n_blocks = [], 0                      # slots of enqueued, not yet examined blocks
            while done < max_steps and n_total is None:
                n = min(chunk, max_steps - done)
                for gi, ((g0, g1), ws, km) in enumerate(zip(groups, wss, masks)):
                    xb = xchg[gi]
                    if xb is not None:
                        _lib.check(lib.ctts_taco_decoder_steps_persistent_f32(
                            C.byref(cfg), _lib.ptr(blob), _lib.ptr(km), _lib.ptr(mel[g0:g1]), _lib.ptr(gate[g0:g1]),
                            _lib.ptr(align[g0:g1]), g1 - g0, T, done, n, max_steps, _lib.ptr(ws), _lib.ptr(xb),
                            xb.numel() * 8, stream), "ctts_taco_decoder_steps_persistent_f32")
                        if self._persist == "unprobed":
                            # first persistent launch of this decoder (or the first after a re-probe): make sure all 256
                            # workgroups really were resident together (a bounded wait gives up otherwise and leaves the
                            # decoder state untouched), else fall back to the per-launch form for the next
                            # PERSIST_REPROBE_AFTER calls of THIS decoder and redo this block with it
                            stream_obj.synchronize()
                            self._persist = "ok"
                            words = ctl_words(xb).cpu()
                            if int(words[0]) != 0:
                                import warnings
                                warnings.warn(f"persistent decoder kernel could not run (workgroup {int(words[1])} gave up in "
                                              f"phase {int(words[2])} of step {int(words[3])}: workgroups not co-resident?); this "
                                              f"decoder uses the per-launch form for its next {PERSIST_REPROBE_AFTER} calls")
                                self._persist, self._persist_fallback_calls = "disabled", 0
                                xchg = [None] * len(groups)      # every remaining group of this call too
                                self._xchg = {}
                                xb = None
                    if xb is None:
                        _lib.check(lib.ctts_taco_decoder_steps_hidden_f32(
                            C.byref(cfg), _lib.ptr(blob), _lib.ptr(km), _lib.ptr(mel[g0:g1]), _lib.ptr(gate[g0:g1]),
                            _lib.ptr(align[g0:g1]), None if hidden is None else _lib.ptr(hidden[g0:g1]), g1 - g0, T, done, n,
                            max_steps, _lib.ptr(ws), stream), "ctts_taco_decoder_steps_hidden_f32")
                done += n
                if fixed_steps is None:
                    _lib.check(lib.ctts_taco_stop_rule_f32(_lib.ptr(gate), B, max_steps, done - n, n,
                                                           float(self.gate_threshold), int(self.gate_delay), _lib.ptr(state),
                                                           stream), "ctts_taco_stop_rule_f32")
                    slot = n_blocks % PINNED_SLOTS
                    n_blocks += 1
                    pinned[slot, 0:1].copy_(verdict_dev, non_blocking=True)
                    for gi, xb in enumerate(xchg):              # a kernel abort ends the loop at the next block
                        if xb is not None:
                            pinned[slot, 1 + gi:2 + gi].copy_(ctl_words(xb)[0:1], non_blocking=True)
                        else:
                            pinned[slot, 1 + gi] = 0
                    events[slot].record(stream_obj)
                    pending.append(slot)
                    while len(pending) > (1 if done < max_steps else 0) and n_total is None:
                        slot0 = pending.pop(0)
                        events[slot0].synchronize()
                        if bool((pinned[slot0, 1:] != 0).any()):
                            check_persistent()                  # raises with the recorded (workgroup, phase, step)
                        if int(pinned[slot0, 0]) >= 0:
                            n_total = int(pinned[slot0, 0])
            if any(x is not None for x in xchg):
                stream_obj.synchronize()
                check_persistent()
            if n_total is None:
                n_total = max_steps
                if fixed_steps is None:
                    print("Warning! Reached max decoder steps")
        return (mel[:, :, :n_total].contiguous(), torch.sigmoid(gate[:, :n_total]), align[:, :n_total].contiguous(),
                None if hidden is None else hidden[:, :, :n_total].contiguous())


MAX_GROUP = 4      # utterances per persistent-decoder / packed-LSTM workspace (ctts_taco_decoder_steps_persistent_f32, ctts_lstm_seq_*: batch <= 4)
# batches from this size on decode in ONE workspace on the batched MFMA form (ctts_taco_decoder_max_batch rows at most); below it,
# groups of MAX_GROUP on the persistent kernel.  Measured (profiles/r6_05): the batched step costs 63-64 us from 5 to 16 rows, a
# persistent group of <= 4 rows 22 us - two groups (<= 8 rows) are ahead of one batched call, three are not
BATCHED_FROM = int(os.environ.get("CTTS_TACO_BATCHED_FROM", "9"))
PAD = 8            # halo of the padded [B][C][ld] layout used by the conv primitives (>= kernel_size // 2)


def _ld_for(T):
    return (T + 127) // 128 * 128 + 2 * PAD


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _HipConv1d:
    """One packed ``ctts_conv1d`` operator: Conv1d (+ eval BatchNorm1d folded) + activation."""

    def __init__(self, conv, bn, act, slope, device):
        lib = _lib.lib()
        w = conv.weight.detach().float().contiguous()
        b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        self.desc = _lib.Conv1dDesc(c_in=w.shape[1], c_out=w.shape[0], kernel_size=w.shape[2], act=act, slope=slope)
        nbytes = lib.ctts_conv1d_packed_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise _lib.HipLibraryError("unsupported conv1d: " + lib.ctts_last_error().decode())
        self.blob = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
        bn_t = [None] * 4
        eps = 1e-5
        if bn is not None:
            bn_t = [t.detach().float().contiguous() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
            eps = bn.eps
        with torch.cuda.device(device):
            _lib.check(lib.ctts_conv1d_pack_f32(C.byref(self.desc), _lib.ptr(w), _lib.ptr(b), *[_lib.ptr(t) for t in bn_t],
                                               eps, _lib.ptr(self.blob), _stream(device)), "ctts_conv1d_pack_f32")
            torch.cuda.current_stream(device).synchronize()

    def __call__(self, x, y, accumulate, B, T, ld):
        _lib.check(_lib.lib().ctts_conv1d_f32(C.byref(self.desc), _lib.ptr(self.blob), _lib.ptr(x), _lib.ptr(y),
                                             1 if accumulate else 0, B, T, ld, PAD, _stream(x.device)), "ctts_conv1d_f32")


class Postnet(nn.Module):
    """model.py:196-228; ``forward`` runs the 6 convs (+BN+tanh, residual every 3) as ``ctts_conv1d`` operators."""

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

    def _invalidate(self):
        self._hip_ops = None

    def _ops(self, device):
        key = _cache.param_key(self)
        if getattr(self, "_hip_ops", None) is None or self._hip_ops[0] != device or self._hip_ops[-1] != key:
            n = len(self.convolutions)
            ops = []
            for i, seq in enumerate(self.convolutions):
                is_out = (bool(self.b_res) and bool(i % self.b_res == 0)) or (i + 1 == n)
                ops.append(_HipConv1d(seq[0].conv, None if is_out else seq[1], 0 if is_out else 2, 0.0, device))
            self._hip_ops = (device, ops, key)
        return self._hip_ops[1]

    def _apply(self, fn, *a, **kw):
        self._hip_ops = None
        return super()._apply(fn, *a, **kw)

    @torch.no_grad()
    def forward(self, x):
        """x [B, n_mel, T] (device) -> x + postnet residuals, eval mode."""
        if self.training:
            raise RuntimeError("Postnet HIP path is inference-only: call .eval()")
        device = x.device
        if device.type != 'cuda':
            raise _lib.HipLibraryError("Tacotron2 HIP path needs GPU tensors (no CPU fallback)")
        ops = self._ops(device)
        lib = _lib.lib()
        B, M, T = x.shape
        ld = _ld_for(T)
        xd = x.detach().float().contiguous()
        n = len(self.convolutions)
        with torch.cuda.device(device):
            x_orig = torch.zeros(B, M, ld, dtype=torch.float32, device=device)
            _lib.check(lib.ctts_pad_rows_f32(_lib.ptr(xd), M * T, T, _lib.ptr(x_orig), B, M, T, ld, PAD, _stream(device)),
                       "ctts_pad_rows_f32")
            cur = x_orig
            for i, op in enumerate(ops):
                if (bool(self.b_res) and bool(i % self.b_res == 0)) or (i + 1 == n):
                    y = x_orig.clone()                     # x_orig + conv(cur), without aliasing the conv's input halo
                    op(cur, y, True, B, T, ld)
                    x_orig = cur = y
                else:
                    y = torch.zeros(B, op.desc.c_out, ld, dtype=torch.float32, device=device)
                    op(cur, y, False, B, T, ld)
                    cur = y
            out = torch.empty(B, M, T, dtype=torch.float32, device=device)
            _lib.check(lib.ctts_unpad_rows_f32(_lib.ptr(x_orig), _lib.ptr(out), M * T, T, B, M, T, ld, PAD, _stream(device)),
                       "ctts_unpad_rows_f32")
        return out.to(x.dtype)


class Encoder(nn.Module):
    """model.py:231-316; ``forward`` runs embedding gather, 3 x (conv + BN + LeakyReLU) and the packed BiLSTM as
    HIP operators and returns the memory tensor's encoder part in place."""

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

    def _apply(self, fn, *a, **kw):
        self._hip_ops = None
        return super()._apply(fn, *a, **kw)

    def _invalidate(self):
        self._hip_ops = None

    def _ops(self, device):
        key = _cache.param_key(self)
        if getattr(self, "_hip_ops", None) is None or self._hip_ops[0] != device or self._hip_ops[-1] != key:
            lib = _lib.lib()
            convs = [_HipConv1d(seq[0].conv, seq[1], 1, 0.01, device) for seq in self.convolutions]
            I, H = self.lstm.input_size, self.lstm.hidden_size
            packs = []
            with torch.cuda.device(device):
                for sfx in ("", "_reverse"):
                    ts = [getattr(self.lstm, n + "_l0" + sfx).detach().float().contiguous()
                          for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                    w = _lib.LstmWeights(*[t.data_ptr() for t in ts])
                    blob = torch.zeros(lib.ctts_lstm_seq_packed_bytes(I, H) // 4, dtype=torch.float32, device=device)
                    _lib.check(lib.ctts_lstm_seq_pack_f32(C.byref(w), I, H, _lib.ptr(blob), _stream(device)),
                               "ctts_lstm_seq_pack_f32")
                    torch.cuda.current_stream(device).synchronize()
                    packs.append(blob)
            self._hip_ops = (device, convs, packs, key)
        return self._hip_ops[1], self._hip_ops[2]

    @torch.no_grad()
    def forward_into_memory(self, embedding_weight, text_seq, text_lengths, speaker_ids, memory_in, hn):
        """Writes encoder outputs into memory_in[:, :, :encoder_LSTM_dim] (zeros beyond each length, as
        pad_packed_sequence leaves them) and the final hidden states [fwd | bwd] into hn."""
        if self.training:
            raise RuntimeError("Encoder HIP path is inference-only: call .eval()")
        device = text_seq.device
        convs, packs = self._ops(device)
        lib = _lib.lib()
        B, T = text_seq.shape
        ld = _ld_for(T)
        E = embedding_weight.shape[1]
        S = self.encoder_speaker_embed_dim
        I, H = self.lstm.input_size, self.lstm.hidden_size
        with torch.cuda.device(device):
            st = _stream(device)
            x = torch.zeros(B, E + S, ld, dtype=torch.float32, device=device)
            spk_w = self.encoder_speaker_embedding.weight.detach().float().contiguous() if S else None
            emb_w = embedding_weight.detach().float().contiguous()
            text64 = text_seq.to(torch.int64).contiguous()
            spk64 = speaker_ids.to(torch.int64).contiguous()
            _lib.check(lib.ctts_taco_embed_f32(_lib.ptr(emb_w), _lib.ptr(spk_w), _lib.ptr(text64), _lib.ptr(spk64), _lib.ptr(x),
                                              B, T, E, S, ld, PAD, st), "ctts_taco_embed_f32")
            for op in convs:
                y = torch.zeros(B, op.desc.c_out, ld, dtype=torch.float32, device=device)
                op(x, y, False, B, T, ld)
                x = y
            lens = text_lengths.to(device=device, dtype=torch.int32).contiguous()
            row = memory_in.shape[2]
            # the packed-sequence LSTM takes <= MAX_GROUP utterances per call on its VALU step kernels, up to 256 where the
            # library runs the recurrent product as an MFMA GEMM over the batch (hidden size a multiple of 64)
            per = MAX_GROUP
            if B > MAX_GROUP and lib.ctts_lstm_seq_workspace_bytes(H, min(B, 256), ld) > 0:
                per = 256
            for g0 in range(0, B, per):
                g1 = min(g0 + per, B)
                nbytes = lib.ctts_lstm_seq_workspace_bytes(H, g1 - g0, ld)
                if nbytes == 0:
                    raise _lib.HipLibraryError("lstm_seq workspace query failed: " + lib.ctts_last_error().decode())
                # both directions in lockstep: one launch per time step instead of two (the layer is launch-bound)
                ws = torch.empty(2, nbytes // 4, dtype=torch.float32, device=device)
                _lib.check(lib.ctts_lstm_biseq_f32(_lib.ptr(packs[0]), _lib.ptr(packs[1]), _lib.ptr(x[g0:g1]),
                                                   _lib.ptr(lens[g0:g1]), _lib.ptr(memory_in[g0:g1]), T * row, row, 0, H,
                                                   _lib.ptr(hn[g0:g1]), 2 * H, 0, H, g1 - g0, T, I, H, ld, PAD,
                                                   _lib.ptr(ws[0]), _lib.ptr(ws[1]), nbytes, st), "ctts_lstm_biseq_f32")


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
        device = text_seq.device
        if device.type != 'cuda':
            raise _lib.HipLibraryError("Tacotron2 HIP path needs GPU tensors (no CPU fallback)")
        lib = _lib.lib()
        B, txt_T = text_seq.shape
        enc_dim = self.encoder.lstm.hidden_size * 2
        row = self.decoder._memory_in_dim
        memory = torch.zeros(B, txt_T, row, dtype=torch.float32, device=device)
        hn = torch.zeros(B, enc_dim, dtype=torch.float32, device=device)
        pred_sylps = torch.empty(B, 1, dtype=torch.float32, device=device)
        # (Encoder) text -> memory[:, :, :enc_dim], final hidden states
        self.encoder.forward_into_memory(self.embedding.weight, text_seq, text_lengths, speaker_id, memory, hn)
        # (Speaker / SylpsNet / torchMoji) per-utterance columns of the memory, pred_sylps
        keep = []

        def dev(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        sn = self.sylps_net.seq_layers
        mw = _lib.TacoMemoryWeights(
            sylps_w=dev(self.encoder.sylps_layer.linear_layer.weight), sylps_b=dev(self.encoder.sylps_layer.linear_layer.bias),
            speaker_embedding=dev(self.speaker_embedding.weight),
            syl_w0=dev(sn[0].linear_layer.weight), syl_b0=dev(sn[0].linear_layer.bias),
            syl_w2=dev(sn[2].linear_layer.weight), syl_b2=dev(sn[2].linear_layer.bias),
            syl_res_weight=dev(self.sylps_net.res_weight.reshape(1)),
            tm_w=dev(self.tm_linear.weight), tm_b=dev(self.tm_linear.bias))
        if hasattr(self, 'tm_bn'):
            mw.tm_gamma, mw.tm_beta = dev(self.tm_bn.weight), dev(self.tm_bn.bias)
            mw.tm_mean, mw.tm_var = dev(self.tm_bn.running_mean), dev(self.tm_bn.running_var)
        tmh = torchmoji_hdn.detach().to(device=device, dtype=torch.float32).contiguous()
        spk64 = speaker_id.to(torch.int64).contiguous()
        # model.py:1058 ``gt_sylps or pred_sylps``: a given syllables-per-second value [B] or [B, 1] feeds the SylpsNet instead of the
        # predicted one (the reference's ``or`` only works for one utterance; "use gt_sylps if given" is what it stands for)
        gts = None if gt_sylps is None else torch.as_tensor(gt_sylps).detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        if gts is not None and gts.numel() != B:
            raise ValueError(f"gt_sylps has {gts.numel()} values for a batch of {B}")
        with torch.cuda.device(device):
            _lib.check(lib.ctts_taco_memory_sylps_f32(C.byref(mw), _lib.ptr(hn), _lib.ptr(spk64), _lib.ptr(tmh), _lib.ptr(gts),
                                                     _lib.ptr(memory), _lib.ptr(pred_sylps), B, txt_T, enc_dim,
                                                     self.speaker_embedding_dim, sn[0].linear_layer.out_features, tmh.shape[1],
                                                     self.tm_linear.out_features, _stream(device)), "ctts_taco_memory_sylps_f32")
        pred_mel, pred_gate, alignments, hidden = self.decoder.inference(memory, memory_lengths=text_lengths, keep_masks=keep_masks,
                                                                         fixed_steps=fixed_steps, return_hidden_state=return_hidden_state)
        pred_mel_postnet = self.postnet(pred_mel) if hasattr(self, 'postnet') else pred_mel
        out = {"pred_mel_postnet": pred_mel_postnet, "pred_gate": pred_gate, "alignments": alignments,
               "pred_sylps": pred_sylps, "pred_mel": pred_mel, "encoder_outputs": memory[:, :, :enc_dim]}
        if return_hidden_state:      # (the reference computes them and drops them from its dict, model.py:1069-1079; kept here)
            out["hidden_att_contexts"] = hidden
        return out


def load_model(hparams):
    """model.py:22-33."""
    model = Tacotron2(hparams)
    if torch.cuda.is_available():
        model = model.cuda()
    # fp16_run (model.py:25-31) only swaps the attention's score_mask_value for finfo(float16).min: masked energies then give
    # exp(-65504 - max) = 0 exactly in fp32, as -inf does - the window kernels' weights past a text's length are exact zeros
    # either way.  (Half precision itself is apex's business in the reference's trainer; inference here computes in fp32.)
    return model
