#!/usr/bin/env python3
"""Localise a difference between the two forms of the decoder loop: run n steps with each, then compare every state array of
the decoder workspace (layout = dec_carve of csrc/tacotron_plan.h) and the per-row mel error."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_tacotron as tt  # noqa: E402

m, g, hp, sd = tt._model()
mem = torch.from_numpy(g["memory_in"]).cuda()
lens = torch.from_numpy(g["lengths"]).cuda()
B, T = mem.shape[0], mem.shape[1]
NB = 1 if B <= 1 else 2 if B <= 2 else 4
A, Ra, Rd, Dm, P = 192, 1280, 768, 512, 256


def carve():
    o = 0
    out = {}

    def take(name, n):
        nonlocal o
        out[name] = (o, n)
        o = (o + n + 63) // 64 * 64
    take("memory", NB * T * Dm); take("pm", NB * T * A)
    take("att_h0", NB * Ra); take("att_h1", NB * Ra); take("att_c", NB * Ra)
    take("dec_h0", NB * Rd); take("dec_h1", NB * Rd); take("dec_c", NB * Rd)
    take("d2_h0", NB * Rd); take("d2_h1", NB * Rd); take("d2_c", NB * Rd)
    take("w", NB * T); take("cum", NB * T); take("ctx", NB * Dm); take("pos", NB); take("prenet", NB * P)
    return out


lay = carve()
for n in (1, 2):
    snap = {}
    for form in ("persistent", "per_launch"):
        m.decoder.use_persistent = form == "persistent"
        mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=g["masks"], fixed_steps=n)
        torch.cuda.synchronize()
        ws = m.decoder._ws[(mem.device, B, T)][0].cpu().numpy()
        snap[form] = ({k: ws[o:o + c].copy() for k, (o, c) in lay.items()}, mel.cpu().numpy())
    print(f"---- after {n} step(s): max |persistent - per_launch| per state array (and its argmax)")
    fin = n & 1
    for k in lay:
        if k in ("memory", "pm"):
            continue
        if k.endswith("0") or k.endswith("1"):
            if int(k[-1]) != fin:
                continue
        d = np.abs(snap["persistent"][0][k] - snap["per_launch"][0][k])
        bad = np.nonzero(d > 1e-5)[0]
        print(f"{k:8s} max {d.max():.3e} at {int(d.argmax())}  (> 1e-5: {len(bad)} of {len(d)}; first {bad[:12].tolist()})")
    dm = np.abs(snap["persistent"][1] - snap["per_launch"][1])
    print("mel diff per step:", dm.max(axis=(0, 1)))
