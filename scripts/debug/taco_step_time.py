"""Decoder step time of the persistent Tacotron decoder (config 5: B=4, 200 symbols, 900 forced steps) under a list of settings of
one environment knob, in one process: python scripts/debug/taco_step_time.py CTTS_TACO_NO_FUSE unset 1
(profiles/r5_37 used it with an experimental knob that was not kept)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import _lib, synthetic  # noqa: E402
from cookietts_amd.tacotron2 import Tacotron2  # noqa: E402


def main():
    knob, values = sys.argv[1], sys.argv[2:]
    hp = synthetic.tacotron_hparams()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)))
    m = m.cuda().eval()
    B, T, steps = int(os.environ.get("TACO_B", "4")), int(os.environ.get("TACO_T", "200")), 900      # (TACO_B / TACO_T: other batch sizes / text lengths)
    rng = np.random.default_rng(1234)
    lens = torch.tensor([T, max(T - 5, 1), max(3 * T // 4, 1), max(T // 2, 1)][:B]).cuda()
    mem = torch.from_numpy((rng.standard_normal((B, T, 1313)) * 0.5).astype(np.float32)).cuda()
    for v in values:
        if v == "unset":
            os.environ.pop(knob, None)
        else:
            os.environ[knob] = v
        _lib.tuning_reload()
        out = m.decoder.inference(mem, lens, fixed_steps=steps)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            out = m.decoder.inference(mem, lens, fixed_steps=steps)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{knob}={v}: {best / steps * 1e6:.2f} us/step (best of 5), decoder form {m.decoder.persistent_state}", flush=True)


if __name__ == "__main__":
    main()
