"""The stop rule of Decoder.inference (model.py:879-904) against goldens in which the REFERENCE'S OWN rule ends the loop
(tests/golden/make_golden.py tacotron_stop): full-size model, B=4, lengths [72,64,48,56], gate logits that cross the
threshold at steps 23 / 40 / 61 / 52 per item - item 0 also on steps 1-3, which the rule ignores (``if i > 4``), and on
steps 23-24 only until step 40 (the rule keeps the running max).  Nine cases set ``gate_threshold`` / ``gate_delay`` /
``max_decoder_steps`` on the decoder the way the server does (text2speech.py:410-412,457): T_mel 62 / 64 / 65 / 66 / 72
straddle the device rule's 32-step blocks, two cases hit the step cap (before the crossing; inside the delay), one never
crosses.  T_mel - integer work - must be bit-exact; mel / gate / alignments within the 1e-4 of BASELINE.json."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from cookietts_amd import synthetic
from oracle import tacotron_oracle as to

MEL_TOL = 1e-4
EXPECT_T = {"delay0": 62, "delay2": 64, "delay3": 65, "delay4": 66, "delay10": 72, "thr_hi": 90, "cap_before": 50,
            "cap_in_delay": 64, "never": 80}


def _load():
    g = np.load(os.path.join(GOLDEN, "tacotron_stop.npz"))
    hp = synthetic.tacotron_hparams()
    shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
    sd = synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes,
                                       attention_drive=tuple(float(x) for x in g["attention_drive"]),
                                       stop_drive=(float(g["stop_rate"]), float(g["stop_sharp"]), g["stop_times"].tolist()))
    sd["decoder.gate_layer.linear_layer.weight"] = g["gate_w"]
    sd["decoder.gate_layer.linear_layer.bias"] = g["gate_b"]
    cases = {str(n): (float(p[0]), int(p[1]), int(p[2]), int(t))
             for n, p, t in zip(g["case_names"], g["case_params"], g["T_mel"])}
    masks = synthetic.prenet_dropout_masks(200, len(g["lengths"]), hp.prenet_dim, seed=int(g["mask_seed"]))
    return g, hp, sd, masks, cases


def test_golden_holds_what_the_docstring_says():
    g, hp, sd, masks, cases = _load()
    assert {k: v[3] for k, v in cases.items()} == EXPECT_T
    sg = g["pred_gate"]
    assert sg.shape == (4, 90) and g["decoder_mel"].shape == (4, 80, 90) and g["alignments"].shape == (4, 90, 72)
    assert [int(np.argmax(sg[b, 5:] > 0.5) + 5) for b in range(4)] == [23, 40, 61, 52]
    assert (sg[0, 1:4] > 0.5).all() and (sg[0, 25:40] < 0.5).all()           # the two traps
    assert np.abs(sg - 0.5).min() > 0.2                                        # no crossing rests on rounding
    for name, (thr, delay, cap, T) in cases.items():
        assert g[f"{name}_pred_mel_postnet"].shape == (4, 80, T)


def test_oracle_stop_step_matches_the_reference_runs():
    """The oracle's loop (decoder steps + stop_step, model.py:879-904) gives the reference's T_mel in every case, and its
    outputs over the longest run match the reference's."""
    g, hp, sd, masks, cases = _load()
    o = to.tacotron_inference_steps(sd, hp, g["text"], g["lengths"], g["speakers"], g["torchmoji"], masks, 200)
    for name, (thr, delay, cap, T) in cases.items():
        assert to.stop_step(o["gate_logits"], thr, delay, cap) == T, name
    n = g["pred_gate"].shape[1]
    sig = 1 / (1 + np.exp(-o["gate_logits"][:, :n].astype(np.float64)))
    e = (np.abs(sig - g["pred_gate"]).max(), np.abs(o["pred_mel"][:, :, :n] - g["decoder_mel"]).max(),
         np.abs(o["alignments"][:, :n] - g["alignments"]).max())
    print(f"oracle vs reference over {n} steps: gate {e[0]:.2e} mel {e[1]:.2e} alignments {e[2]:.2e}")
    assert max(e) < 1e-5
    for name, (thr, delay, cap, T) in cases.items():
        post = to.postnet(sd, hp, o["pred_mel"][:, :, :T])
        assert np.abs(post - g[f"{name}_pred_mel_postnet"]).max() < 1e-5, name


def _model(sd, hp):
    from cookietts_amd.tacotron2 import Tacotron2
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["persistent", "per_launch", "batched"])
def test_hip_stop_rule_ends_the_loop_where_the_reference_does(hip_lib_path, tuning, form):
    """Tacotron2.inference with the stop rule live (no ``fixed_steps``): T_mel bit-exact in all nine cases, outputs within
    1e-4 of the reference's, all three forms of the decoder (batched = the MFMA form of batch > 4, forced here at B = 4)."""
    g, hp, sd, masks, cases = _load()
    m = _model(sd, hp)
    m.decoder.use_persistent = form == "persistent"
    if form == "per_launch":
        tuning.set("CTTS_TACO_VALU")
    args = [torch.from_numpy(g[k]).cuda() for k in ("text", "lengths", "speakers", "torchmoji")]
    for name, (thr, delay, cap, T) in cases.items():
        m.decoder.gate_delay = int(delay)                   # text2speech.py:410
        m.decoder.max_decoder_steps = int(cap)              # :411
        m.decoder.gate_threshold = float(thr)               # :412
        out = m.inference(*args, keep_masks=masks)
        o = {k: v.cpu().numpy() for k, v in out.items()}
        assert o["pred_mel_postnet"].shape == (4, 80, T), (name, o["pred_mel_postnet"].shape, T)
        assert o["pred_gate"].shape == (4, T) and o["alignments"].shape == (4, T, 72) and o["pred_mel"].shape == (4, 80, T)
        e = (np.abs(o["pred_mel"] - g["decoder_mel"][:, :, :T]).max(), np.abs(o["pred_gate"] - g["pred_gate"][:, :T]).max(),
             np.abs(o["alignments"] - g["alignments"][:, :T]).max(),
             np.abs(o["pred_mel_postnet"] - g[f"{name}_pred_mel_postnet"]).max())
        print(f"{form} / {name}: T_mel {T}; L_inf decoder mel {e[0]:.2e} gate {e[1]:.2e} alignments {e[2]:.2e} "
              f"postnet mel {e[3]:.2e}")
        assert max(e) < MEL_TOL, (name, e)
    assert bool(m.decoder._xchg) == (form == "persistent")


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [1, 7, 32, 61, 62, 63, 200])
def test_hip_stop_rule_is_independent_of_the_block_size(hip_lib_path, chunk, monkeypatch):
    """The device rule folds a block of gate logits at a time; the step it stops at must not depend on where the block
    boundaries fall relative to the crossing (61), the stop (61 + delay) or the ignored steps (0-4)."""
    from cookietts_amd import tacotron2
    g, hp, sd, masks, cases = _load()
    m = _model(sd, hp)
    monkeypatch.setattr(tacotron2, "_CHUNK", chunk)
    args = [torch.from_numpy(g[k]).cuda() for k in ("text", "lengths", "speakers", "torchmoji")]
    for name in ("delay0", "delay3", "cap_in_delay"):
        thr, delay, cap, T = cases[name]
        m.decoder.gate_delay, m.decoder.max_decoder_steps, m.decoder.gate_threshold = delay, cap, thr
        out = m.inference(*args, keep_masks=masks)
        assert out["pred_mel"].shape[2] == T, (name, chunk, out["pred_mel"].shape[2])
        assert np.abs(out["pred_mel"].cpu().numpy() - g["decoder_mel"][:, :, :T]).max() < MEL_TOL
