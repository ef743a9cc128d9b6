"""The batched MFMA form of the decoder step (csrc/tacotron_batched.h): the batch sizes the reference's server decodes in one
``Decoder.inference`` call (_5_infer/t2s_server/text2speech.py:418-424, 537, 554: up to 256 rows) - against the reference's own
outputs where a golden exists (B = 8 ragged: ``tacotron_batch8.npz``; the B = 4 long-horizon and stop-rule goldens run this form
as the third ``form`` of tests/test_tacotron_long.py / test_tacotron_stop.py: it is what ctts_taco_decoder_steps_f32 runs at any batch, CTTS_TACO_VALU selects the older VALU kernels) and against the
oracle at other shapes."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from cookietts_amd import synthetic
from oracle import tacotron_oracle as to

MEL_TOL = 1e-4            # BASELINE.json: mel L_inf <= 1e-4


def _model(hp=None, sd=None):
    from cookietts_amd import Tacotron2
    hp = hp or synthetic.tacotron_hparams()
    if sd is None:
        shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
        sd = synthetic.tacotron_state_dict(hp, seed=11, shapes=shapes)
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval(), hp, sd


def _inputs(hp, B, T, lens, n, seed):
    rng = np.random.default_rng(seed)
    memory_in = (rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)
    return memory_in, np.asarray(lens, dtype=np.int64), synthetic.prenet_dropout_masks(n, B, hp.prenet_dim, seed=seed + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,n", [(5, 37, 9), (16, 60, 9), (33, 45, 7), (64, 40, 5), (70, 40, 5)])
def test_batched_decoder_matches_oracle(hip_lib_path, monkeypatch, B, T, n):
    """Column-tile shapes 16 / 32 / 64 / 128 items (NT = 1, 2, 4 with K split over two workgroups, 4 x 2), ragged lengths incl. texts
    shorter than the window.
    (B = 5 is below the size from which the host picks this form by itself: the line is moved for the test.)"""
    import cookietts_amd.tacotron2 as t2
    monkeypatch.setattr(t2, "BATCHED_FROM", 5)
    m, hp, sd = _model()
    rng = np.random.default_rng(B)
    lens = [T] + [int(x) for x in rng.integers(18, T + 1, size=B - 1)]
    memory_in, lengths, masks = _inputs(hp, B, T, lens, n, seed=100 + B)
    ref_mel, ref_gate, ref_align = to.decoder_inference_steps(sd, hp, memory_in, lengths, masks, n)
    mel, gate, align, _ = m.decoder.inference(torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda(),
                                              keep_masks=masks, fixed_steps=n)
    assert not m.decoder._xchg or all(x is None for x in next(iter(m.decoder._xchg.values())))     # one workspace, no persistent launch
    e = (np.abs(mel.cpu().numpy() - ref_mel).max(), np.abs(align.cpu().numpy() - ref_align).max(),
         np.abs(gate.cpu().numpy() - 1 / (1 + np.exp(-ref_gate))).max())
    print(f"batched B={B}: L_inf mel {e[0]:.2e} alignments {e[1]:.2e} gate {e[2]:.2e}")
    assert max(e) < MEL_TOL


@pytest.mark.gpu
def test_batched_rows_do_not_depend_on_the_rest_of_the_batch(hip_lib_path):
    """An item's frames are a function of its own memory and masks only: the first six rows of a 7-item call equal a 6-item
    call bit for bit (same column-tile shape), and the groups-of-4 persistent form within rounding."""
    import cookietts_amd.tacotron2 as t2
    t2_from = t2.BATCHED_FROM
    t2.BATCHED_FROM = 5
    m, hp, sd = _model()
    B, T, n = 7, 37, 14
    memory_in, lengths, masks = _inputs(hp, B, T, [37, 30, 21, 37, 18, 25, 33], n, seed=77)
    mem, lens = torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda()
    full = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
    six = m.decoder.inference(mem[:6].contiguous(), lens[:6], keep_masks=np.ascontiguousarray(masks[:, :, :6]), fixed_steps=n)
    assert torch.equal(six[0], full[0][:6]) and torch.equal(six[2], full[2][:6]) and torch.equal(six[1], full[1][:6])
    old = t2_from
    try:
        t2.BATCHED_FROM = 1 << 30                       # groups of MAX_GROUP in lockstep on the persistent / per-launch forms
        m.decoder._ws, m.decoder._xchg = {}, {}
        grouped = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
    finally:
        t2.BATCHED_FROM = old
        m.decoder._ws, m.decoder._xchg = {}, {}
    d = float((grouped[0] - full[0]).abs().max())
    print(f"batched vs groups of 4: mel L_inf {d:.2e}")
    assert d < 2e-5


def _batch8():
    g = np.load(os.path.join(GOLDEN, "tacotron_batch8.npz"))
    hp = synthetic.tacotron_hparams()
    shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
    sd = synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes, attention_drive=tuple(float(x) for x in g["attention_drive"]))
    n = int(g["n_steps"])
    return g, hp, sd, synthetic.prenet_dropout_masks(n, 8, hp.prenet_dim, seed=int(g["mask_seed"])), n


def test_oracle_matches_the_eight_utterance_reference_golden():
    """tests/golden/make_golden.py tacotron_batch8: the reference's Tacotron2.inference on eight ragged utterances (lengths 64 ..
    12: two texts shorter than the 33-token window), 48 forced steps."""
    g, hp, sd, masks, n = _batch8()
    o = to.tacotron_inference_steps(sd, hp, g["text"], g["lengths"], g["speakers"], g["torchmoji"], masks, n)
    e = (np.abs(o["alignments"] - g["alignments"]).max(), np.abs(o["pred_mel"] - g["decoder_mel"]).max(),
         np.abs(o["pred_mel_postnet"] - g["pred_mel_postnet"]).max(), np.abs(1 / (1 + np.exp(-o["gate_logits"])) - g["pred_gate"]).max())
    print("oracle vs reference, batch 8:", e)
    assert e[0] < 5e-5 and max(e[1:]) < 1e-5            # (the peaked recipe's weights: same bound as tacotron_long_peaked)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["batched", "groups"])
def test_hip_matches_the_eight_utterance_reference_golden(hip_lib_path, monkeypatch, form):
    """The same golden through Tacotron2.inference on the GPU: eight utterances as ONE batched-form decoder call, and as the
    host runs them by default (two groups of four on the persistent kernel, in lockstep)."""
    import cookietts_amd.tacotron2 as t2
    monkeypatch.setattr(t2, "BATCHED_FROM", 5 if form == "batched" else 9)
    g, hp, sd, masks, n = _batch8()
    m, _, _ = _model(hp, sd)
    out = m.inference(torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["lengths"]).cuda(), torch.from_numpy(g["speakers"]).cuda(),
                      torch.from_numpy(g["torchmoji"]).cuda(), keep_masks=masks, fixed_steps=n)
    assert len(next(iter(m.decoder._ws.values()))) == (1 if form == "batched" else 2)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    e = (np.abs(o["alignments"] - g["alignments"]).max(), np.abs(o["pred_mel"] - g["decoder_mel"]).max(),
         np.abs(o["pred_mel_postnet"] - g["pred_mel_postnet"]).max(), np.abs(o["pred_gate"] - g["pred_gate"]).max())
    print(f"HIP {form} vs reference, batch 8: L_inf alignments {e[0]:.2e} decoder mel {e[1]:.2e} postnet mel {e[2]:.2e} gate {e[3]:.2e}")
    assert max(e) < MEL_TOL
    for b, L in enumerate(g["lengths"]):
        assert (o["alignments"][b, :, L:] == 0).all()


@pytest.mark.gpu
def test_batched_form_is_bit_identical_run_to_run(hip_lib_path):
    """Sixteen rows, 120 steps, same memory and dropout masks, three runs: every output bit for bit (fixed summation orders: the
    K slices of a workgroup, the EARLY sums of the pipelined step's launches in slot order)."""
    m, hp, sd = _model()
    B, T, n = 16, 90, 120
    rng = np.random.default_rng(16)
    memory_in, lengths, masks = _inputs(hp, B, T, [T] + [int(x) for x in rng.integers(40, T + 1, size=B - 1)], n, seed=160)
    mem, lens = torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda()
    ref = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
    assert len(next(iter(m.decoder._ws.values()))) == 1 and torch.isfinite(ref[0]).all()
    for _ in range(2):
        out = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
        assert all(torch.equal(a, b) for a, b in zip(ref[:3], out[:3]))


@pytest.mark.gpu
def test_sixteen_rows_stop_together_where_the_reference_stops_its_four(hip_lib_path):
    """The stop-rule golden's four utterances four times over = 16 rows in one batched-form call: the rule waits for the LAST row
    (model.py:898-904: ``sig_max_gates.min() > gate_threshold``), rows are independent, so T_mel is the golden's in every case and
    every copy of an utterance gets the golden's frames."""
    from test_tacotron_stop import _load, _model as _stop_model
    g, hp, sd, masks, cases = _load()
    m = _stop_model(sd, hp)
    rep = lambda a: np.concatenate([a] * 4, axis=0)                        # noqa: E731
    order = np.argsort(-rep(g["lengths"]), kind="stable")                  # pack_padded_sequence wants the lengths sorted
    args = [torch.from_numpy(rep(g[k])[order]).cuda() for k in ("text", "lengths", "speakers", "torchmoji")]
    masks16 = np.ascontiguousarray(np.concatenate([masks] * 4, axis=2)[:, :, order])
    src = (order % 4)
    for name in ("delay0", "delay3", "thr_hi", "cap_in_delay", "never"):
        thr, delay, cap, T = cases[name]
        m.decoder.gate_delay, m.decoder.max_decoder_steps, m.decoder.gate_threshold = int(delay), int(cap), float(thr)
        out = m.inference(*args, keep_masks=masks16)
        assert len(next(iter(m.decoder._ws.values()))) == 1
        o = out["pred_mel_postnet"].cpu().numpy()
        assert o.shape == (16, 80, T), (name, o.shape, T)
        e = np.abs(o - g[f"{name}_pred_mel_postnet"][src]).max()
        print(f"16 rows / {name}: T_mel {T}, postnet mel L_inf vs the reference's four {e:.2e}")
        assert e < MEL_TOL


@pytest.mark.gpu
def test_more_rows_than_one_workspace_takes_run_as_lockstep_groups_of_256(hip_lib_path):
    """260 rows = a group of 256 and one of 4 (one workspace each, the same block of steps for both, one stop-rule evaluation):
    every row equals the same row decoded in a batch of its own kind."""
    m, hp, sd = _model()
    B, T, n = 260, 24, 4
    rng = np.random.default_rng(260)
    memory_in, lengths, masks = _inputs(hp, B, T, [T] + [int(x) for x in rng.integers(18, T + 1, size=B - 1)], n, seed=2600)
    mem, lens = torch.from_numpy(memory_in).cuda(), torch.from_numpy(lengths).cuda()
    full = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=n)
    assert len(next(iter(m.decoder._ws.values()))) == 2 and torch.isfinite(full[0]).all()
    head = m.decoder.inference(mem[:256].contiguous(), lens[:256], keep_masks=np.ascontiguousarray(masks[:, :, :256]), fixed_steps=n)
    assert torch.equal(head[0], full[0][:256]) and torch.equal(head[2], full[2][:256])
    ref_mel, _, ref_align = to.decoder_inference_steps(sd, hp, memory_in[256:], lengths[256:], np.ascontiguousarray(masks[:, :, 256:]), n)
    assert np.abs(full[0][256:].cpu().numpy() - ref_mel).max() < MEL_TOL and np.abs(full[2][256:].cpu().numpy() - ref_align).max() < MEL_TOL
