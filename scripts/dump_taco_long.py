#!/usr/bin/env python3
"""Dump the HIP decoder's outputs on the long Tacotron goldens (both forms) to gpurun_out/ for offline comparison against
the fp64 trajectory of oracle/tacotron_oracle.py (the arbiter is too slow to be re-run in every GPU call)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_tacotron_long as tl  # noqa: E402

out_dir = os.path.join(REPO, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
for name in sys.argv[1:] or ["long_sharp"]:
    g, hp, sd, masks, n = tl._load(name)
    m = tl._model(sd, hp)
    for form in ("persistent", "per_launch"):
        m.decoder.use_persistent = form == "persistent"
        out = m.inference(torch.from_numpy(g["text"]).cuda(), torch.from_numpy(g["lengths"]).cuda(),
                          torch.from_numpy(g["speakers"]).cuda(), torch.from_numpy(g["torchmoji"]).cuda(),
                          keep_masks=masks, fixed_steps=n)
        np.savez_compressed(os.path.join(out_dir, f"hip_taco_{name}_{form}.npz"),
                            **{k: v.cpu().numpy() for k, v in out.items() if k != "encoder_outputs"})
        print("dumped", name, form)
