"""Decoder step time of the default Tacotron2 decoder over a list of batch sizes: fixed-step Decoder.inference, text of 200
symbols (config 5's shape), whichever form the host picks (persistent <= 4, batched above; CTTS_TACO_BATCHED_FROM moves the line).
usage: python scripts/debug/taco_batch_time.py [B ...] [--steps N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import Tacotron2, synthetic  # noqa: E402


def main():
    argv = sys.argv[1:]
    steps = 256
    if "--steps" in argv:
        i = argv.index("--steps")
        steps = int(argv[i + 1])
        del argv[i:i + 2]
    small = "--small" in argv                 # the non-default-widths golden shape (synthetic.TACOTRON_SMALL_OVERRIDES)
    if small:
        argv.remove("--small")
    args = argv
    batches = [int(a) for a in args] or [1, 4, 5, 8, 16, 32, 64, 128, 256]
    hp = synthetic.tacotron_hparams(**(synthetic.TACOTRON_SMALL_OVERRIDES if small else {}))
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)), strict=False)
    m = m.cuda().eval()
    T = 200
    for B in batches:
        rng = np.random.default_rng(B)
        mem = torch.from_numpy((rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)).cuda()
        lens = torch.full((B,), T, dtype=torch.int64).cuda()
        keep = (torch.rand(steps, 2, B, hp.prenet_dim, device="cuda") < 0.5).to(torch.uint8)
        for _ in range(2):
            m.decoder.inference(mem, lens, keep_masks=keep, fixed_steps=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = m.decoder.inference(mem, lens, keep_masks=keep, fixed_steps=steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        groups = len(next(iter(m.decoder._ws.values())))
        print(f"B={B:4d}: {dt * 1e6 / steps:8.2f} us/step  {B * steps / dt / 1e3:9.1f} k frames/s  ({groups} workspace(s), finite={bool(torch.isfinite(out[0]).all())})",
              flush=True)


if __name__ == "__main__":
    main()
