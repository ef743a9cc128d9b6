"""Tacotron2-TM decoder loop (config 5): oracle vs reference golden (CPU), HIP vs golden / oracle (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from cookietts_amd import synthetic
from oracle import tacotron_oracle as to

MEL_TOL = 1e-4            # BASELINE.json: mel L_inf <= 1e-4


def _golden():
    g = np.load(os.path.join(GOLDEN, "tacotron_decoder.npz"))
    hp = synthetic.tacotron_hparams()
    shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
    return g, hp, synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes), shapes


def test_oracle_matches_reference_decoder_golden():
    g, hp, sd, _ = _golden()
    mel, gate, align = to.decoder_inference_steps(sd, hp, g["memory_in"], g["lengths"], g["masks"], g["masks"].shape[0])
    assert mel.shape == g["mel"].shape == (2, 80, 14)
    assert np.abs(mel - g["mel"]).max() < 1e-6
    assert np.abs(1 / (1 + np.exp(-gate)) - g["gate_sigmoid"]).max() < 1e-6
    assert np.abs(align - g["alignments"]).max() < 1e-6
    assert ((g["alignments"] > 0).sum(axis=2) == 33).all()        # +-16 window (model.py:131-146)
    assert np.allclose(g["alignments"].sum(axis=2), 1.0, atol=1e-5)


def test_host_module_tree_matches_reference_state_dict():
    from cookietts_amd.tacotron2 import Tacotron2
    _, hp, sd, shapes = _golden()
    m = Tacotron2(hp)
    own = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert own == shapes                                           # same keys, same shapes as the reference's
    m.load_state_dict(synthetic.to_torch(sd))
    assert synthetic.tacotron_state_dict(hp, seed=1234).keys() == sd.keys()


def test_stop_rule_matches_reference_semantics():
    """model.py:879-904: max of sigmoid(gate) tracked only for i > 4; stop gate_delay steps after ALL items crossed."""
    from cookietts_amd.tacotron2 import stop_step
    gates = np.full((2, 40), -5.0, dtype=np.float32)
    gates[0, 2] = 9.0                      # ignored: i <= 4
    gates[0, 10:] = 9.0
    gates[1, 17] = 9.0                     # second item crosses at step 17 -> break_point = 17 + delay
    for delay in (0, 3):
        want = to.stop_step(gates, 0.5, delay, 1000)
        assert want == 17 + delay + 1
        n, state = None, None
        for c0 in range(0, 40, 7):         # fed in blocks, like the device loop
            n, state = stop_step(torch.from_numpy(gates[:, c0:c0 + 7]), 0.5, delay, 1000, state)
            if n is not None:
                break
        assert n == want
    assert to.stop_step(np.full((1, 12), -9.0, np.float32), 0.5, 0, 10) == 10      # max_decoder_steps cap


def test_unsupported_hparams_fail_loudly():
    from cookietts_amd.tacotron2 import Tacotron2
    with pytest.raises(NotImplementedError):
        Tacotron2(synthetic.tacotron_hparams(attention_type=1))
    with pytest.raises(NotImplementedError):
        Tacotron2(synthetic.tacotron_hparams(windowed_attention_range=0))


def _model():
    from cookietts_amd.tacotron2 import Tacotron2
    g, hp, sd, _ = _golden()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval(), g, hp, sd


@pytest.mark.gpu
def test_hip_decoder_matches_reference_golden(hip_lib_path):
    m, g, hp, sd = _model()
    n = g["masks"].shape[0]
    mel, gate, align, _ = m.decoder.inference(torch.from_numpy(g["memory_in"]).cuda(),
                                              torch.from_numpy(g["lengths"]).cuda(),
                                              keep_masks=g["masks"], fixed_steps=n)
    mel, gate, align = mel.cpu().numpy(), gate.cpu().numpy(), align.cpu().numpy()
    print("mel Linf vs reference:", np.abs(mel - g["mel"]).max())
    assert np.abs(mel - g["mel"]).max() < MEL_TOL
    assert np.abs(gate - g["gate_sigmoid"]).max() < MEL_TOL
    assert np.abs(align - g["alignments"]).max() < MEL_TOL
    assert ((align > 0).sum(axis=2) <= 33).all() and np.allclose(align.sum(axis=2), 1.0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,lens", [(1, 20, [20]), (3, 45, [45, 17, 33]), (4, 200, [200, 195, 150, 100])])
def test_hip_decoder_matches_oracle_shapes(hip_lib_path, B, T, lens):
    """Short texts (window clamps: len-17 < 16), odd batch (padded to 4 internally), config-5 lengths."""
    m, g, hp, sd = _model()
    rng = np.random.default_rng(B * 1000 + T)
    memory_in = (rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)
    lengths = np.array(lens, dtype=np.int64)
    n = 9
    masks = synthetic.prenet_dropout_masks(n, B, seed=T)
    ref_mel, ref_gate, ref_align = to.decoder_inference_steps(sd, hp, memory_in, lengths, masks, n)
    mel, gate, align, _ = m.decoder.inference(torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda(),
                                              keep_masks=masks, fixed_steps=n)
    assert np.abs(mel.cpu().numpy() - ref_mel).max() < MEL_TOL
    assert np.abs(align.cpu().numpy() - ref_align).max() < MEL_TOL
    assert np.abs(gate.cpu().numpy() - 1 / (1 + np.exp(-ref_gate))).max() < MEL_TOL


@pytest.mark.gpu
def test_hip_decoder_stop_rule_and_chunking(hip_lib_path):
    """Free-running inference: the device loop is cut at the step the reference's rule would stop at, and
    chunked execution (stop check every 32 steps) equals one fixed run of the same length."""
    m, g, hp, sd = _model()
    with torch.no_grad():
        m.decoder.gate_layer.linear_layer.bias.fill_(3.0)     # every gate > threshold -> stops at i = 5 + delay
    m.decoder._invalidate()
    m.decoder.gate_delay = 4
    m.decoder.max_decoder_steps = 100
    mem = torch.from_numpy(g["memory_in"]).cuda()
    lens = torch.from_numpy(g["lengths"]).cuda()
    masks = synthetic.prenet_dropout_masks(100, 2, seed=3)
    mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=masks)
    assert mel.shape[2] == 5 + 4 + 1 and gate.shape == (2, 10) and align.shape == (2, 10, 60)
    m.decoder.gate_threshold = 2.0                             # never stops: runs to max_decoder_steps in chunks
    m.decoder.max_decoder_steps = 70
    a = m.decoder.inference(mem, lens, keep_masks=masks)
    b = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=70)
    assert a[0].shape[2] == 70 and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


@pytest.mark.gpu
def test_device_stop_rule_matches_reference_semantics(hip_lib_path):
    """ctts_taco_stop_rule_f32 fed block by block == the per-step rule of model.py:879-904 (oracle restatement), on
    crafted gate logits: early crossings ignored for i <= 4, the slowest utterance decides, gate_delay, the
    max_decoder_steps cap, stickiness of the verdict, a batch wider than the workgroup."""
    import ctypes as C
    from cookietts_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    cases = []
    g = np.full((2, 40), -5.0, np.float32); g[0, 2] = 9.0; g[0, 10:] = 9.0; g[1, 17] = 9.0
    cases += [(g, 0.5, 0, 1000), (g, 0.5, 3, 1000), (g, 0.5, 30, 40)]
    cases += [(np.full((1, 12), -9.0, np.float32), 0.5, 0, 12)]                           # never crosses: cap
    big = rng.normal(-3.0, 1.0, (300, 90)).astype(np.float32)                             # 300 utterances > 256 threads
    big[np.arange(300), rng.integers(6, 60, 300)] = 6.0
    cases += [(big, 0.9, 2, 90)]
    for gates, thr, delay, max_steps in cases:
        B, n_all = gates.shape
        want = to.stop_step(gates, thr, delay, max_steps)
        gd = torch.from_numpy(gates).cuda()
        state = torch.empty(lib.ctts_taco_stop_state_bytes(B) // 4, dtype=torch.float32, device="cuda")
        _lib.check(lib.ctts_taco_stop_reset(_lib.ptr(state), B, max_steps, None), "reset")
        got = None
        for c0 in range(0, n_all, 7):                       # blocks of 7 steps, like the device loop's blocks of 32
            n = min(7, n_all - c0)
            _lib.check(lib.ctts_taco_stop_rule_f32(_lib.ptr(gd), B, n_all, c0, n, thr, delay, _lib.ptr(state), None), "rule")
            v = int(state[B + 1:B + 2].view(torch.int32).item())
            if got is None and v >= 0:
                got = v
            assert got is None or v == got                   # sticky once set
        assert (got if got is not None else max_steps) == want, (want, got, thr, delay)


@pytest.mark.gpu
def test_full_tacotron_inference_contract(hip_lib_path):
    """Tacotron2.inference drop-in contract (model.py:1044-1080): dict keys and shapes."""
    m, g, hp, sd = _model()
    B, T = 2, 30
    rng = np.random.default_rng(0)
    text = torch.from_numpy(rng.integers(1, 179, size=(B, T))).cuda()
    lens = torch.tensor([30, 22]).cuda()
    spk = torch.tensor([0, 1]).cuda()
    tm = torch.from_numpy(rng.standard_normal((B, 2304)).astype(np.float32)).cuda()
    out = m.inference(text, lens, spk, tm, fixed_steps=12)
    assert out["pred_mel_postnet"].shape == (B, 80, 12) and out["pred_gate"].shape == (B, 12)
    assert out["alignments"].shape == (B, 12, T) and out["pred_sylps"].shape == (B, 1)
    assert all(torch.isfinite(out[k]).all() for k in ("pred_mel_postnet", "pred_gate", "alignments"))


@pytest.mark.gpu
def test_full_model_matches_reference_golden(hip_lib_path):
    """Tacotron2.inference end to end (embedding, encoder convs+BN+LeakyReLU, packed BiLSTM, memory assembly,
    decoder loop, postnet) against the reference's own outputs."""
    m, _, hp, sd = _model()
    g = np.load(os.path.join(GOLDEN, "tacotron_full.npz"))
    n = g["masks"].shape[0]
    out = m.inference(torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["lengths"]).cuda(),
                      torch.from_numpy(g["speakers"]).cuda(), torch.from_numpy(g["torchmoji"]).cuda(),
                      keep_masks=g["masks"], fixed_steps=n)
    enc = out["encoder_outputs"].cpu().numpy()
    print("encoder Linf:", np.abs(enc - g["encoder_outputs"]).max(),
          " postnet mel Linf:", np.abs(out["pred_mel_postnet"].cpu().numpy() - g["pred_mel_postnet"]).max())
    assert np.abs(enc - g["encoder_outputs"]).max() < MEL_TOL
    assert (enc[1, 22:] == 0).all() and (enc[2, 9:] == 0).all()            # pad_packed_sequence zeros
    assert np.abs(out["pred_sylps"].cpu().numpy() - g["pred_sylps"]).max() < MEL_TOL
    assert np.abs(out["pred_mel_postnet"].cpu().numpy() - g["pred_mel_postnet"]).max() < MEL_TOL
    assert np.abs(out["pred_gate"].cpu().numpy() - g["pred_gate"]).max() < MEL_TOL
    assert np.abs(out["alignments"].cpu().numpy() - g["alignments"]).max() < MEL_TOL


def test_oracle_full_model_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "tacotron_full.npz"))
    _, hp, sd, _ = _golden()
    o = to.tacotron_inference_steps(sd, hp, g["text"], g["lengths"], g["speakers"], g["torchmoji"], g["masks"],
                                    g["masks"].shape[0])
    assert np.abs(o["encoder_outputs"] - g["encoder_outputs"]).max() < 1e-6
    assert np.abs(o["pred_sylps"] - g["pred_sylps"]).max() < 1e-6
    assert np.abs(o["pred_mel_postnet"] - g["pred_mel_postnet"]).max() < 1e-5
    assert np.abs(o["alignments"] - g["alignments"]).max() < 1e-6


@pytest.mark.gpu
def test_batches_larger_than_a_device_group_run_in_lockstep(hip_lib_path, monkeypatch):
    """A workspace of the persistent / VALU forms holds 4 utterances; batches beyond what one workspace takes run as groups
    that advance together (one stop-rule evaluation over the whole batch, model.py:898-904).  7 utterances = groups of 4 + 3
    here because the batched form (one workspace up to 256 rows: tests/test_tacotron_batched.py) is switched off."""
    from cookietts_amd import tacotron2
    monkeypatch.setattr(tacotron2, "BATCHED_FROM", 1 << 30)
    m, g, hp, sd = _model()
    B, T, n = 7, 37, 14
    rng = np.random.default_rng(77)
    memory_in = (rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)
    lengths = np.array([37, 30, 21, 37, 18, 25, 33], dtype=np.int64)
    masks = synthetic.prenet_dropout_masks(n, B, seed=8)
    mem, lens = torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda()
    mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
    ref_mel, ref_gate, ref_align = to.decoder_inference_steps(sd, hp, memory_in, lengths, masks, n)
    assert np.abs(mel.cpu().numpy() - ref_mel).max() < MEL_TOL
    assert np.abs(align.cpu().numpy() - ref_align).max() < MEL_TOL
    for g0, g1 in ((0, 4), (4, 7)):                       # each group alone reproduces its rows bit for bit
        one = m.decoder.inference(mem[g0:g1].contiguous(), lens[g0:g1], keep_masks=np.ascontiguousarray(masks[:, :, g0:g1]),
                                  fixed_steps=n)
        assert torch.equal(one[0], mel[g0:g1]) and torch.equal(one[2], align[g0:g1])
    # free-running: every item past the threshold -> the batch stops together, 5 utterances through the whole model
    text = torch.from_numpy(rng.integers(1, 179, size=(5, 30))).cuda()
    tl = torch.tensor([30, 22, 17, 30, 9]).cuda()
    tm = torch.from_numpy(rng.standard_normal((5, 2304)).astype(np.float32)).cuda()
    spk = torch.arange(5).cuda()
    out = m.inference(text, tl, spk, tm, fixed_steps=6)
    assert out["pred_mel_postnet"].shape == (5, 80, 6) and torch.isfinite(out["pred_mel_postnet"]).all()
    part = m.inference(text[4:], tl[4:], spk[4:], tm[4:], keep_masks=None, fixed_steps=6)
    assert np.abs(out["encoder_outputs"][4].cpu().numpy() - part["encoder_outputs"][0].cpu().numpy()).max() < 1e-6


@pytest.mark.gpu
def test_packed_weights_follow_load_state_dict_on_the_parent(hip_lib_path):
    """Decoder / Encoder / Postnet keep packed blobs; Tacotron2.load_state_dict (the parent) must invalidate them."""
    m, _, hp, sd = _model()
    g = np.load(os.path.join(GOLDEN, "tacotron_full.npz"))
    args = (torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["lengths"]).cuda(),
            torch.from_numpy(g["speakers"]).cuda(), torch.from_numpy(g["torchmoji"]).cuda())
    n = g["masks"].shape[0]
    shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
    other = synthetic.tacotron_state_dict(hp, seed=4321, shapes=shapes)
    m.load_state_dict(synthetic.to_torch(other))
    first = m.inference(*args, keep_masks=g["masks"], fixed_steps=n)["pred_mel_postnet"].cpu().numpy()
    assert np.abs(first - g["pred_mel_postnet"]).max() > 1e-2                 # different weights, different mel
    m.load_state_dict(synthetic.to_torch(sd))                                 # back to the golden's weights
    again = m.inference(*args, keep_masks=g["masks"], fixed_steps=n)
    assert np.abs(again["pred_mel_postnet"].cpu().numpy() - g["pred_mel_postnet"]).max() < MEL_TOL
    assert np.abs(again["encoder_outputs"].cpu().numpy() - g["encoder_outputs"]).max() < MEL_TOL


def test_persistent_decoder_size_query(hip_lib_path):
    """ctts_taco_decoder_persistent_bytes answers on the host: built for the repo-default decoder shape only."""
    import ctypes as C
    from cookietts_amd import _lib
    from cookietts_amd.tacotron2 import Tacotron2
    lib = _lib.lib()
    cfg = Tacotron2(synthetic.tacotron_hparams()).decoder.c_config()
    nb = lib.ctts_taco_decoder_persistent_bytes(C.byref(cfg), 4, 200)
    # granules: 2 parities x 4 batch rows x (256 + 1280 + 192 + 512 + 768 + 768 + 256) x 8 bytes + control words
    assert nb >= 2 * 4 * 4032 * 8 and nb % 8 == 0
    assert lib.ctts_taco_decoder_persistent_bytes(C.byref(cfg), 5, 200) == 0          # batch > 4: per-launch form
    assert lib.ctts_taco_decoder_persistent_bytes(C.byref(cfg), 4, 5000) == 0         # text longer than the LDS staging
    other = type(cfg).from_buffer_copy(cfg)
    other.attention_rnn_dim = 1024
    assert lib.ctts_taco_decoder_persistent_bytes(C.byref(other), 4, 200) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("other", ["batched", "valu"])
def test_persistent_decoder_equals_per_launch_decoder(hip_lib_path, tuning, other):
    """The persistent kernel (one launch per block of steps) and the per-launch forms - the batched MFMA form (seven launches per
    step: what ctts_taco_decoder_steps_f32 runs by default) and the VALU kernels (six launches, CTTS_TACO_VALU) - implement the
    same arithmetic in different summation orders: same outputs to fp32 rounding, over a free-running decode that crosses
    several launch blocks, with ragged lengths, and for a batch of 1 (padded rows of the exchange stay zero)."""
    if other == "valu":
        tuning.set("CTTS_TACO_VALU")
    m, g, hp, sd = _model()
    rng = np.random.default_rng(11)
    for B, T, n in ((4, 200, 150), (1, 33, 40), (3, 64, 70)):
        mem = torch.from_numpy((rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)).cuda()
        lens = torch.from_numpy(rng.integers(T // 2, T + 1, B).astype(np.int64)).cuda()
        lens[0] = T
        masks = synthetic.prenet_dropout_masks(n, B, seed=B)
        m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, n           # runs all n steps, in blocks of 32
        m.decoder.use_persistent = True
        a = m.decoder.inference(mem, lens, keep_masks=masks)
        m.decoder.use_persistent = False
        b = m.decoder.inference(mem, lens, keep_masks=masks)
        m.decoder.use_persistent = True
        assert a[0].shape == b[0].shape == (B, 80, n)
        print(f"B={B} T={T}: mel {float((a[0] - b[0]).abs().max()):.2e} align {float((a[2] - b[2]).abs().max()):.2e}")
        assert (a[0] - b[0]).abs().max() < MEL_TOL and (a[1] - b[1]).abs().max() < MEL_TOL
        assert (a[2] - b[2]).abs().max() < MEL_TOL
    assert m.decoder.persistent_state == "ok"                                      # the persistent path really ran


@pytest.mark.gpu
def test_persistent_decoder_abort_is_detected_and_falls_back(hip_lib_path):
    """The control words are the LAST 64 bytes of the exchange buffer on both sides of the ABI: a non-zero word 0 there
    (what a bounded wait that gave up leaves behind; the kernel then returns at entry without touching anything) must
    surface as HipLibraryError with the recorded (workgroup, phase, step) - not as a zero-filled mel - and, on the first
    persistent launch of a process, as a warning plus the per-launch decoder's result."""
    import ctypes as C
    import warnings
    from cookietts_amd import _lib
    from cookietts_amd.tacotron2 import PERSIST_CTL_WORDS
    m, g, hp, sd = _model()
    dec = m.decoder
    rng = np.random.default_rng(5)
    B, T, n = 2, 48, 40
    mem = torch.from_numpy((rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)).cuda()
    lens = torch.tensor([T, T - 9]).cuda()
    masks = synthetic.prenet_dropout_masks(n, B, seed=3)
    dec.gate_threshold, dec.max_decoder_steps = 2.0, n
    dec.use_persistent = False
    want = dec.inference(mem, lens, keep_masks=masks)
    dec.use_persistent = True
    good = dec.inference(mem, lens, keep_masks=masks)
    assert dec.persistent_state == "ok"
    assert (good[0] - want[0]).abs().max() < MEL_TOL
    (key, bufs), = dec._xchg.items()
    nb = _lib.lib().ctts_taco_decoder_persistent_bytes(C.byref(dec.c_config()), B, T)
    assert bufs[0].numel() * 8 == nb and nb % 64 == 0

    def poison(xb):
        ctl = xb[-PERSIST_CTL_WORDS:].view(torch.int32)
        ctl[0], ctl[1], ctl[2], ctl[3] = 1, 7, 3, 5
    poison(bufs[0])
    with pytest.raises(_lib.HipLibraryError, match="workgroup 7, phase 3, step 5"):
        dec.inference(mem, lens, keep_masks=masks)
    assert dec._xchg == {}                                   # sticky words dropped with the buffer: the next call is clean
    again = dec.inference(mem, lens, keep_masks=masks)
    assert torch.equal(again[0], good[0])
    # first-launch probe of THIS decoder: same poison -> warning, per-launch result, this decoder's persistent form off for
    # its next PERSIST_REPROBE_AFTER calls - and a second model in the same process is not touched by it
    from cookietts_amd import tacotron2
    other, _, _, _ = _model()
    dec.reprobe_persistent()
    assert dec.persistent_state == "unprobed" and dec._xchg == {}
    dec.inference(mem, lens, keep_masks=masks)                       # allocates the exchange buffer, probes fine
    dec._persist = "unprobed"
    (key, bufs), = dec._xchg.items()
    poison(bufs[0])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fb = dec.inference(mem, lens, keep_masks=masks)
    assert any("per-launch form" in str(x.message) and "workgroup 7" in str(x.message) for x in w)
    assert dec.persistent_state == "disabled" and dec._xchg == {}
    assert torch.equal(fb[0], want[0]) and torch.equal(fb[2], want[2])
    assert other.decoder.persistent_state == "unprobed"
    other.decoder.gate_threshold, other.decoder.max_decoder_steps = 2.0, n
    og = other.decoder.inference(mem, lens, keep_masks=masks)
    assert other.decoder.persistent_state == "ok" and other.decoder._xchg and torch.equal(og[0], good[0])
    for _ in range(tacotron2.PERSIST_REPROBE_AFTER):                  # the per-launch form, quietly
        fb = dec.inference(mem, lens, keep_masks=masks)
        assert dec.persistent_state == "disabled" and torch.equal(fb[0], want[0])
    back = dec.inference(mem, lens, keep_masks=masks)                # re-probed by itself
    assert dec.persistent_state == "ok" and dec._xchg and torch.equal(back[0], good[0])
    dec.use_persistent = False
    assert dec.persistent_state == "off"


@pytest.mark.gpu
def test_bidirectional_lstm_launch_equals_two_one_direction_runs(hip_lib_path):
    """ctts_lstm_biseq_f32 (both directions of the encoder BiLSTM in one launch per time step) against two
    ctts_lstm_seq_f32 runs: the same kernels' arithmetic, so bit-equal outputs and final states, ragged lengths."""
    import ctypes as C
    from cookietts_amd import _lib
    from cookietts_amd.tacotron2 import PAD, _ld_for
    m, g, hp, sd = _model()
    enc = m.encoder
    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    convs, packs = enc._ops(dev)
    I, H = enc.lstm.input_size, enc.lstm.hidden_size
    B, T = 3, 77
    ld = _ld_for(T)
    rng = np.random.default_rng(77)
    x = torch.zeros(B, I, ld, device=dev)
    x[:, :, PAD:PAD + T] = torch.from_numpy(rng.standard_normal((B, I, T)).astype(np.float32)).to(dev)
    lens = torch.tensor([77, 40, 1], dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nbytes = lib.ctts_lstm_seq_workspace_bytes(H, B, ld)
    row = 2 * H + 5
    outs = []
    for both in (False, True):
        out = torch.zeros(B, T, row, device=dev)
        hn = torch.zeros(B, 2 * H, device=dev)
        ws = torch.empty(2, nbytes // 4, device=dev)
        if both:
            _lib.check(lib.ctts_lstm_biseq_f32(_lib.ptr(packs[0]), _lib.ptr(packs[1]), _lib.ptr(x), _lib.ptr(lens), _lib.ptr(out),
                                               T * row, row, 0, H, _lib.ptr(hn), 2 * H, 0, H, B, T, I, H, ld, PAD, _lib.ptr(ws[0]),
                                               _lib.ptr(ws[1]), nbytes, st), "ctts_lstm_biseq_f32")
        else:
            for d in range(2):
                _lib.check(lib.ctts_lstm_seq_f32(_lib.ptr(packs[d]), _lib.ptr(x), _lib.ptr(lens), d, _lib.ptr(out), T * row, row,
                                                 d * H, _lib.ptr(hn), 2 * H, d * H, B, T, I, H, ld, PAD, _lib.ptr(ws[d]), nbytes,
                                                 st), "ctts_lstm_seq_f32")
        torch.cuda.synchronize()
        outs.append((out, hn))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0][0, :, :2 * H].abs().max()) > 0 and float(outs[0][0][1, 40:].abs().max()) == 0.0
    assert lib.ctts_lstm_biseq_f32(_lib.ptr(packs[0]), _lib.ptr(packs[1]), _lib.ptr(x), _lib.ptr(lens), _lib.ptr(outs[0][0]),
                                   T * row, row, 0, H, _lib.ptr(outs[0][1]), 2 * H, 0, H, B, T, I, H, ld, PAD, _lib.ptr(ws[0]),
                                   _lib.ptr(ws[0]), nbytes, st) != 0             # one workspace for both directions: refused


# ---- a model whose hparams are not the repo defaults ---------------------------------------------
# The server builds the model from checkpoint['hparams'] (text2speech.py:299-316): every width roughly halved here
# (synthetic.TACOTRON_SMALL_OVERRIDES); golden = the reference's own Tacotron2.inference of that model.
def _small():
    g = np.load(os.path.join(GOLDEN, "tacotron_small.npz"))
    hp = synthetic.tacotron_hparams(**synthetic.TACOTRON_SMALL_OVERRIDES)
    shapes = json.load(open(os.path.join(GOLDEN, "tacotron_small_state_shapes.json")))
    return g, hp, synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes), shapes


def test_non_default_hparams_build_the_reference_module_tree_and_match_the_oracle():
    from cookietts_amd.tacotron2 import Tacotron2
    g, hp, sd, shapes = _small()
    m = Tacotron2(hp)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes   # keys / shapes of the REFERENCE's module for this hp
    m.load_state_dict(synthetic.to_torch(sd))
    assert m.decoder.c_config().attention_rnn_dim == 640 and m.decoder.c_config().prenet_dim == 128
    o = to.tacotron_inference_steps(sd, hp, g["text"], g["lengths"], g["speakers"], g["torchmoji"], g["masks"],
                                    g["masks"].shape[0])
    assert np.abs(o["encoder_outputs"] - g["encoder_outputs"]).max() < 1e-6
    assert np.abs(o["pred_mel_postnet"] - g["pred_mel_postnet"]).max() < 1e-5
    assert np.abs(o["alignments"] - g["alignments"]).max() < 1e-6
    assert ((g["alignments"] > 0).sum(axis=2) <= 17).all()                 # +-8 window here


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["batched", "valu"])
def test_hip_non_default_hparams_match_reference_golden(hip_lib_path, tuning, form):
    """Loads from a reference-format state dict and runs on the per-launch decoder (the persistent form is built for the
    default widths only and reports 0 bytes for this shape)."""
    from cookietts_amd import _lib
    from cookietts_amd.tacotron2 import Tacotron2
    import ctypes as C
    if form == "valu":
        tuning.set("CTTS_TACO_VALU")
    g, hp, sd, _ = _small()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(sd))
    m = m.cuda().eval()
    cfg = m.decoder.c_config()
    assert _lib.lib().ctts_taco_decoder_persistent_bytes(C.byref(cfg), 3, 40) == 0      # persistent kernel: default widths only
    assert _lib.lib().ctts_taco_decoder_max_batch(C.byref(cfg)) == 256                  # ... the batched form takes this shape
    n = g["masks"].shape[0]
    out = m.inference(torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["lengths"]).cuda(),
                      torch.from_numpy(g["speakers"]).cuda(), torch.from_numpy(g["torchmoji"]).cuda(),
                      keep_masks=g["masks"], fixed_steps=n)
    enc = out["encoder_outputs"].cpu().numpy()
    errs = {k: float(np.abs(out[k].cpu().numpy() - g[k]).max()) for k in ("pred_sylps", "pred_mel_postnet", "pred_gate", "alignments")}
    print("non-default hparams: encoder Linf", np.abs(enc - g["encoder_outputs"]).max(), errs)
    assert np.abs(enc - g["encoder_outputs"]).max() < MEL_TOL
    assert all(e < MEL_TOL for e in errs.values()), errs


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["persistent", "batched"])
def test_a_nan_with_the_sentinels_bit_pattern_flows_through_as_nan(hip_lib_path, form):
    """The persistent decoder's vectors travel as self-flagging 4-byte values whose "not yet" pattern, 0xFFFFFFFF, is itself a
    quiet NaN.  AMD NaN propagation keeps sign and payload of an input NaN, so a weight with exactly that pattern (a 0xFF-filled
    buffer) would publish the sentinel: every consumer would spin into its timeout and the launch would abort.  The publishers
    canonicalise (pd_bits): the run completes and the frames are NaN, as on the per-launch form."""
    m, g, hp, sd = _model()
    with torch.no_grad():
        m.decoder.attention_rnn.bias_ih[5] = torch.tensor([0xFFFFFFFF - (1 << 32)], dtype=torch.int32).view(torch.float32)[0]
    m.decoder._invalidate()
    m.decoder.use_persistent = form == "persistent"
    n = 40
    mem = torch.from_numpy(g["memory_in"]).cuda()
    lens = torch.from_numpy(g["lengths"]).cuda()
    mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=synthetic.prenet_dropout_masks(n, 2, seed=5), fixed_steps=n)
    assert mel.shape == (2, 80, n) and torch.isnan(mel[:, :, -1]).all()      # it ran all steps; the NaN reached every frame
    if form == "persistent":
        assert m.decoder.persistent_state == "ok" and m.decoder._xchg      # ... on the persistent kernel, without an abort


@pytest.mark.gpu
def test_gt_sylps_and_hidden_states_follow_the_reference_semantics(hip_lib_path):
    """model.py:1044, 1058: a given ``gt_sylps`` feeds the SylpsNet instead of the predicted value (pred_sylps is returned either
    way); model.py:762, 888-889: ``return_hidden_state`` records [dec_h + d2_h | attention context] per step - the vector the
    gate layer and the mel projection read, so the recorded states reproduce the call's own gate logits and mel frames."""
    m, g, hp, sd = _model()
    f = np.load(os.path.join(GOLDEN, "tacotron_full.npz"))
    args = [torch.from_numpy(f[k]).cuda() for k in ("text", "lengths", "speakers", "torchmoji")]
    n = f["masks"].shape[0]
    base = m.inference(*args, keep_masks=f["masks"], fixed_steps=n)
    same = m.inference(*args, gt_sylps=base["pred_sylps"], keep_masks=f["masks"], fixed_steps=n)
    assert torch.equal(same["pred_mel_postnet"], base["pred_mel_postnet"])                # the predicted value given back: no change
    other = m.inference(*args, gt_sylps=base["pred_sylps"][:, 0] * 1.5, keep_masks=f["masks"], fixed_steps=n, return_hidden_state=True)
    assert torch.equal(other["pred_sylps"], base["pred_sylps"])
    assert float((other["pred_mel_postnet"] - base["pred_mel_postnet"]).abs().max()) > 1e-4
    # oracle with the same override
    o = to.tacotron_inference_steps(sd, hp, f["text"], f["lengths"], f["speakers"], f["torchmoji"], f["masks"], n,
                                    gt_sylps=(base["pred_sylps"][:, 0] * 1.5).cpu().numpy())
    assert np.abs(other["pred_mel_postnet"].cpu().numpy() - o["pred_mel_postnet"]).max() < MEL_TOL
    h = other["hidden_att_contexts"]
    dec = m.decoder
    assert h.shape == (3, dec.second_decoder_rnn_dim + dec.memory_dim, n)
    gate_w, gate_b = dec.gate_layer.linear_layer.weight.detach(), dec.gate_layer.linear_layer.bias.detach()
    logits = torch.einsum("bdt,od->bt", h, gate_w) + gate_b
    assert float((torch.sigmoid(logits) - other["pred_gate"]).abs().max()) < 1e-5
    pw, pb = dec.linear_projection.linear_layer.weight.detach(), dec.linear_projection.linear_layer.bias.detach()
    mel = torch.einsum("bdt,od->bot", h, pw) + pb[None, :, None]
    assert float((mel - other["pred_mel"]).abs().max()) < 1e-4
    with pytest.raises(ValueError):
        m.inference(*args, gt_sylps=torch.ones(2), keep_masks=f["masks"], fixed_steps=n)
