"""CPU oracle for the Tacotron2-TM mel loop (BASELINE config 5).

TEST INFRASTRUCTURE ONLY (see oracle/waveglow_oracle.py header for the import rule).
numpy fp32 restatement written from SURVEY.md §8(a) "Verified restatement of rows T5-T9" plus
rows T2-T4, T10.  Reference lines followed (relative to /root/reference/CookieTTS/):
  lstm_cell            utils/model/layers.py:308-372 (eval branch: gates i,f,g,o; no dropout/zoneout)
  prenet               _2_ttm/tacotron2_tm/model.py:180-193 (no biases; dropout ALWAYS applied)
  attention_step       model.py:93-161 + LocationLayer :49-65 (window mask :131-146, softmax, bmm, new_pos)
  decoder_init         model.py:870-871 (MemoryBottleneck :319-332), :507-608 (zero states, memory_layer)
  decoder_step         model.py:668-767
  stop_step            model.py:879-904
  decoder_inference    model.py:851-916
  encoder / postnet / tacotron_inference  model.py:283-316, 218-228, 1044-1080,
                       tacotron2_ssvae/nets/SylpsNet.py:52-61, untts/model.py:310-337 (eval BatchNorm)

Parity pin: tests/golden/tacotron_*.npz = outputs of the reference's own ``Decoder.inference`` /
``Tacotron2.inference`` (tests/golden/make_golden.py) with the prenet dropout masks captured.
"""
from __future__ import annotations

import contextlib

import numpy as np

FT = np.float32        # the float type every intermediate is rounded to (the reference computes in fp32)


@contextlib.contextmanager
def precision(dtype):
    """Evaluate the same equations with every intermediate in ``dtype`` instead of fp32.  ``precision(np.float64)`` is the
    arbiter of tests/test_tacotron_long.py: on a trajectory that amplifies rounding, an fp32 implementation is judged by
    its distance to the fp64 trajectory relative to the reference's own distance to it."""
    global FT
    saved, FT = FT, dtype
    try:
        yield
    finally:
        FT = saved


def _sig(x):
    return (FT(1.0) / (FT(1.0) + np.exp(-x))).astype(FT)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    g = (x @ w_ih.T + b_ih + h @ w_hh.T + b_hh).astype(FT)
    H = h.shape[1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = (_sig(f) * c + _sig(i) * np.tanh(gg)).astype(FT)
    h2 = (_sig(o) * np.tanh(c2)).astype(FT)
    return h2, c2


def prenet(sd, x, keep1, keep2):
    """keep masks are 0/1; kept activations are scaled by 1/(1-p) = 2 (F.dropout p=0.5)."""
    w1 = sd["decoder.prenet.layers.0.linear_layer.weight"]
    w2 = sd["decoder.prenet.layers.1.linear_layer.weight"]
    a = np.maximum(x @ w1.T, 0).astype(FT) * (keep1.astype(FT) * FT(2.0))
    return np.maximum(a @ w2.T, 0).astype(FT) * (keep2.astype(FT) * FT(2.0))


def decoder_init(sd, memory_in):
    wb = sd["decoder.memory_bottleneck.bottleneck.linear_layer.weight"]
    memory = (memory_in @ wb.T).astype(FT)
    if "decoder.memory_bottleneck.bottleneck.linear_layer.bias" in sd:
        memory = memory + sd["decoder.memory_bottleneck.bottleneck.linear_layer.bias"]
    wm = sd["decoder.attention_layer.memory_layer.linear_layer.weight"]
    return memory, (memory @ wm.T).astype(FT)


def attention_step(sd, hp, att_h, memory, processed_memory, w_prev, cum, pos, lengths):
    """Returns (context [B, mem], weights [B, T], new_pos [B])."""
    B, T, _ = memory.shape
    wloc = sd["decoder.attention_layer.location_layer.location_conv.conv.weight"]       # [F, 2, K]
    wd = sd["decoder.attention_layer.location_layer.location_dense.linear_layer.weight"]  # [A, F]
    wq = sd["decoder.attention_layer.query_layer.linear_layer.weight"]                  # [A, R]
    v = sd["decoder.attention_layer.v.linear_layer.weight"][0]                          # [A]
    K = wloc.shape[2]
    padk = (K - 1) // 2
    cat = np.stack([w_prev, cum], axis=1)                                               # [B, 2, T]
    catp = np.pad(cat, ((0, 0), (0, 0), (padk, padk)))
    loc = np.zeros((B, wloc.shape[0], T), dtype=FT)
    for j in range(K):
        loc += np.einsum("fc,bct->bft", wloc[:, :, j], catp[:, :, j:j + T]).astype(FT)
    proc = (np.einsum("bft,af->bta", loc, wd) + (att_h @ wq.T)[:, None, :] + processed_memory).astype(FT)
    e = (np.tanh(proc) @ v).astype(FT)                                                 # [B, T]
    R = hp.windowed_attention_range
    off = sd["decoder.attention_layer.windowed_att_pos_offset"].reshape(-1)[0] if \
        "decoder.attention_layer.windowed_att_pos_offset" in sd else FT(0)
    cur = (pos + off).astype(FT) if off != 0 else pos
    cur = np.minimum(np.maximum(cur, FT(R)), (lengths - 1 - R).astype(FT))
    start = np.rint(np.maximum(cur - FT(R), FT(0)))                                   # round half to even
    end = start + 2 * R
    t = np.arange(T)[None, :]
    allowed = (t >= start[:, None]) & (t <= end[:, None]) & (t < lengths[:, None])
    e = np.where(allowed, e, -np.inf).astype(FT)
    m = e.max(axis=1, keepdims=True)
    p = np.exp(e - m).astype(FT)
    w = (p / p.sum(axis=1, keepdims=True)).astype(FT)
    ctx = np.einsum("bt,btd->bd", w, memory).astype(FT)
    new_pos = (w * np.arange(T, dtype=FT)[None, :]).sum(axis=1).astype(FT)
    return ctx, w, new_pos


def _lstm_params(sd, prefix):
    return (sd[prefix + ".weight_ih"], sd[prefix + ".weight_hh"], sd[prefix + ".bias_ih"], sd[prefix + ".bias_hh"])


def decoder_inference_steps(sd, hp, memory_in, lengths, keep_masks, n_steps):
    """Run exactly ``n_steps`` decoder steps.  Returns mel [B, n_mel, T], gate logits [B, T], alignments
    [B, T, txt_T].  ``keep_masks`` [n_steps, 2, B, prenet_dim] uint8."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    memory_in = np.asarray(memory_in, dtype=FT)
    lengths = np.asarray(lengths).astype(np.int64)
    B, T, _ = memory_in.shape
    memory, pm = decoder_init(sd, memory_in)
    Ra, Rd, Rd2 = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.second_decoder_rnn_dim
    att_h = np.zeros((B, Ra), FT); att_c = np.zeros((B, Ra), FT)
    dec_h = np.zeros((B, Rd), FT); dec_c = np.zeros((B, Rd), FT)
    d2_h = np.zeros((B, Rd2), FT); d2_c = np.zeros((B, Rd2), FT)
    w = np.zeros((B, T), FT); cum = np.zeros((B, T), FT)
    ctx = np.zeros((B, memory.shape[2]), FT); pos = np.zeros((B,), FT)
    x = np.zeros((B, hp.n_mel_channels), FT)
    sf = _sig(sd["decoder.exp_smoothing_factor"].reshape(-1)[0].astype(FT))
    wp, bp = sd["decoder.linear_projection.linear_layer.weight"], sd["decoder.linear_projection.linear_layer.bias"]
    wg, bg = sd["decoder.gate_layer.linear_layer.weight"], sd["decoder.gate_layer.linear_layer.bias"]
    mels, gates, aligns = [], [], []
    for i in range(n_steps):
        p = prenet(sd, x, keep_masks[i, 0], keep_masks[i, 1])
        att_h, att_c = lstm_cell(np.concatenate([p, ctx, dec_h], axis=1), att_h, att_c,
                                 *_lstm_params(sd, "decoder.attention_rnn"))
        ctx, w, new_pos = attention_step(sd, hp, att_h, memory, pm, w, cum, pos, lengths)
        pos = (pos * sf + new_pos * (FT(1.0) - sf)).astype(FT)
        cum = (cum + w).astype(FT)
        dec_h, dec_c = lstm_cell(np.concatenate([att_h, ctx], axis=1), dec_h, dec_c,
                                 *_lstm_params(sd, "decoder.decoder_rnn"))
        d2_h, d2_c = lstm_cell(dec_h, d2_h, d2_c, *_lstm_params(sd, "decoder.second_decoder_rnn"))
        d = (dec_h + d2_h).astype(FT)
        dc = np.concatenate([d, ctx], axis=1)
        gate = (dc @ wg.T + bg).astype(FT)[:, 0]
        mel = (dc @ wp.T + bp).astype(FT)
        mels.append(mel); gates.append(gate); aligns.append(w)
        x = mel
    return (np.stack(mels, axis=2), np.stack(gates, axis=1), np.stack(aligns, axis=1))


def stop_step(gate_logits, gate_threshold, gate_delay, max_decoder_steps):
    """Number of steps the reference's loop executes (model.py:879-904) given per-step gate logits
    [B, >= that many steps].  Returns n_steps (<= max_decoder_steps)."""
    B, n = gate_logits.shape
    sig_max = np.zeros(B, dtype=FT)
    break_point = max_decoder_steps
    for i in range(min(n, max_decoder_steps)):
        if i > 4:
            sig_max = np.maximum(_sig(gate_logits[:, i].astype(FT)), sig_max)
        if sig_max.min() > gate_threshold:
            break_point = min(break_point, i + gate_delay)
        if i >= break_point:
            return i + 1
    return min(n, max_decoder_steps)


# ------------------------------------------------------------------ one-shot stages (T2, T3, T10) ----
def _conv1d_same(x, w, b):
    """x [B, Cin, T], w [Cout, Cin, K] (odd K, zero 'same' padding), b [Cout]."""
    K = w.shape[2]
    p = (K - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p)))
    T = x.shape[2]
    y = np.zeros((x.shape[0], w.shape[0], T), dtype=FT)
    for j in range(K):
        y += np.einsum("oc,bct->bot", w[:, :, j], xp[:, :, j:j + T]).astype(FT)
    return (y + b[None, :, None]).astype(FT)


def _bn_eval(x, sd, prefix, eps=1e-5):
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    shape = (1, -1, 1) if x.ndim == 3 else (1, -1)
    return ((x - m.reshape(shape)) / np.sqrt(v.reshape(shape) + FT(eps)) * g.reshape(shape) + b.reshape(shape)).astype(FT)


def encoder(sd, hp, text, lengths, speaker_ids):
    """model.py:283-316 in eval mode.  Returns (outputs [B, T, 1024] zero beyond each length, pred_sylps [B, 1])."""
    emb = sd["embedding.weight"][text].transpose(0, 2, 1)                        # [B, 512, T]
    spk = sd["encoder.encoder_speaker_embedding.weight"][speaker_ids][:, :, None]
    x = np.concatenate([emb, np.repeat(spk, emb.shape[2], axis=2)], axis=1).astype(FT)
    for i in range(hp.encoder_n_convolutions):
        p = f"encoder.convolutions.{i}"
        x = _bn_eval(_conv1d_same(x, sd[p + ".0.conv.weight"], sd[p + ".0.conv.bias"]), sd, p + ".1")
        x = np.where(x > 0, x, FT(0.01) * x).astype(FT)                        # LeakyReLU(0.01)
    x = x.transpose(0, 2, 1)                                                     # [B, T, 1024]
    B, T, _ = x.shape
    H = hp.encoder_LSTM_dim // 2
    out = np.zeros((B, T, 2 * H), dtype=FT)
    hn = np.zeros((B, 2 * H), dtype=FT)
    for d, sfx in enumerate(["", "_reverse"]):
        wih, whh = sd["encoder.lstm.weight_ih_l0" + sfx], sd["encoder.lstm.weight_hh_l0" + sfx]
        bih, bhh = sd["encoder.lstm.bias_ih_l0" + sfx], sd["encoder.lstm.bias_hh_l0" + sfx]
        for b in range(B):                                                       # packed semantics: per-item walk
            h = np.zeros((1, H), FT); c = np.zeros((1, H), FT)
            order = range(int(lengths[b])) if d == 0 else range(int(lengths[b]) - 1, -1, -1)
            for t in order:
                h, c = lstm_cell(x[b:b + 1, t], h, c, wih, whh, bih, bhh)
                out[b, t, d * H:(d + 1) * H] = h[0]
            hn[b, d * H:(d + 1) * H] = h[0]
    sylps = (hn @ sd["encoder.sylps_layer.linear_layer.weight"].T + sd["encoder.sylps_layer.linear_layer.bias"]).astype(FT)
    return out, sylps


def memory_assemble(sd, hp, enc_out, pred_sylps, speaker_ids, torchmoji_hdn):
    """model.py:1051-1068 (+ SylpsNet.infer_auto, tm_bn eval, tm_linear)."""
    B, T, _ = enc_out.shape
    spk = sd["speaker_embedding.weight"][speaker_ids]
    cat = np.concatenate([pred_sylps, np.log(pred_sylps)], axis=1).astype(FT)   # [B, 2]
    h = cat @ sd["sylps_net.seq_layers.0.linear_layer.weight"].T + sd["sylps_net.seq_layers.0.linear_layer.bias"]
    h = np.where(h > 0, h, FT(0.05) * h).astype(FT)
    res = h @ sd["sylps_net.seq_layers.2.linear_layer.weight"].T + sd["sylps_net.seq_layers.2.linear_layer.bias"]
    sylzu = (cat + sd["sylps_net.res_weight"].reshape(()) * res)[:, 0:1].astype(FT)
    tm = _bn_eval(np.asarray(torchmoji_hdn, FT), sd, "tm_bn")
    tm = (tm @ sd["tm_linear.weight"].T + sd["tm_linear.bias"]).astype(FT)
    rep = lambda v: np.repeat(v[:, None, :], T, axis=1)
    return np.concatenate([enc_out, rep(spk), rep(sylzu), rep(tm)], axis=2).astype(FT)


def postnet(sd, hp, mel):
    """model.py:218-228 in eval mode."""
    n = hp.postnet_n_convolutions
    b_res = hp.postnet_residual_connections
    x_orig = mel.astype(FT).copy()
    x = mel
    for i in range(n):
        p = f"postnet.convolutions.{i}"
        y = _conv1d_same(x, sd[p + ".0.conv.weight"], sd[p + ".0.conv.bias"])
        if (bool(b_res) and i % b_res == 0) or i + 1 == n:
            x_orig = (x_orig + y).astype(FT)
            x = x_orig
        else:
            x = np.tanh(_bn_eval(y, sd, p + ".1")).astype(FT)
    return x_orig


def tacotron_inference_steps(sd, hp, text, lengths, speaker_ids, torchmoji_hdn, keep_masks, n_steps, gt_sylps=None):
    """Tacotron2.inference (model.py:1044-1080) for a fixed number of decoder steps.  ``gt_sylps`` [B]: model.py:1058
    ``gt_sylps or pred_sylps`` - the given value feeds the SylpsNet, pred_sylps is returned either way."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    enc_out, sylps = encoder(sd, hp, np.asarray(text), lengths, np.asarray(speaker_ids))
    syl_in = sylps if gt_sylps is None else np.asarray(gt_sylps, dtype=np.float32).reshape(sylps.shape)
    memory_in = memory_assemble(sd, hp, enc_out, syl_in, np.asarray(speaker_ids), torchmoji_hdn)
    mel, gate, align = decoder_inference_steps(sd, hp, memory_in, lengths, keep_masks, n_steps)
    return dict(encoder_outputs=enc_out, pred_sylps=sylps, memory_in=memory_in, pred_mel=mel,
                pred_mel_postnet=postnet(sd, hp, mel), gate_logits=gate, alignments=align)
