"""Tacotron2.inference end to end (embedding, encoder, memory, decoder with `steps` forced steps, postnet) over a list of batch
sizes: ms per call and the encoder's share.  usage: python scripts/debug/taco_e2e_time.py [B ...] [--steps N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import Tacotron2, synthetic  # noqa: E402


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    argv = sys.argv[1:]
    steps = 400
    if "--steps" in argv:
        i = argv.index("--steps")
        steps = int(argv[i + 1])
        del argv[i:i + 2]
    hp = synthetic.tacotron_hparams()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)))
    m = m.cuda().eval()
    T = 200
    for B in [int(a) for a in argv] or [4, 16, 64, 256]:
        rng = np.random.default_rng(B)
        text = torch.from_numpy(rng.integers(1, 179, size=(B, T))).cuda()
        lens = torch.from_numpy(np.sort(rng.integers(100, T + 1, B))[::-1].copy()).cuda()
        lens[0] = T
        spk = (torch.arange(B) % 64).cuda()
        tm = torch.from_numpy(rng.standard_normal((B, 2304)).astype(np.float32)).cuda()
        full = timed(lambda: m.inference(text, lens, spk, tm, fixed_steps=steps))
        short = timed(lambda: m.inference(text, lens, spk, tm, fixed_steps=1))
        print(f"B={B:4d}: {full * 1e3:9.2f} ms for {steps} steps ({(full - short) / (steps - 1) * 1e6:7.2f} us/step), "
              f"everything but the decoder loop {short * 1e3:8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
