"""Batched decoder under one environment knob against the default: outputs compared bit for bit (mel, gate, alignments) and
us/step, over batch sizes, ragged text lengths.   python scripts/debug/taco_knob_ab.py KNOB [B ...] [--steps N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import Tacotron2, _lib, synthetic  # noqa: E402

argv = sys.argv[1:]
steps = 256
if "--steps" in argv:
    i = argv.index("--steps")
    steps = int(argv[i + 1])
    del argv[i:i + 2]
knob = argv[0]
batches = [int(a) for a in argv[1:]] or [9, 16, 32, 64]
hp = synthetic.tacotron_hparams()
m = Tacotron2(hp)
m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)), strict=False)
m = m.cuda().eval()
T = 200


def run(B, on):
    if on:
        os.environ[knob] = "1"
    else:
        os.environ.pop(knob, None)
    _lib.tuning_reload()
    rng = np.random.default_rng(B)
    mem = torch.from_numpy((rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)).cuda()
    lens = torch.from_numpy(rng.integers(40, T + 1, size=B)).cuda()
    lens[0] = T
    keep = (torch.rand(steps, 2, B, hp.prenet_dim, device="cuda", generator=torch.Generator("cuda").manual_seed(B)) < 0.5).to(torch.uint8)
    out = m.decoder.inference(mem, lens, keep_masks=keep, fixed_steps=steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.decoder.inference(mem, lens, keep_masks=keep, fixed_steps=steps)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / 3 / steps * 1e6


for B in batches:
    a, ta = run(B, False)
    b, tb = run(B, True)
    a2, ta2 = run(B, False)
    same = all(torch.equal(x, y) for x, y in zip(a[:3], b[:3]))
    print(f"B={B:4d}: default {ta:7.2f} / {ta2:7.2f} us/step   {knob} {tb:7.2f} us/step   identical={same} finite={bool(torch.isfinite(a[0]).all())}", flush=True)
