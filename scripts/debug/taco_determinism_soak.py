"""Run-to-run determinism of the persistent Tacotron decoder: the same memory and the same dropout keep-masks N times - every output
(mel, gate, alignments) must be bit-identical.  A stale read in the exchange protocol (a value of step s - 2 taken for step s) would show
up here as a difference: python scripts/debug/taco_determinism_soak.py [runs] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import synthetic  # noqa: E402
from cookietts_amd.tacotron2 import Tacotron2  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    hp = synthetic.tacotron_hparams()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)))
    m = m.cuda().eval()
    bad = 0
    for B, T in ((4, 200), (1, 60), (3, 333)):
        rng = np.random.default_rng(B * 1000 + T)
        lens = torch.tensor([T, max(T - 5, 1), max(3 * T // 4, 1), max(T // 2, 1)][:B]).cuda()
        mem = torch.from_numpy((rng.standard_normal((B, T, 1313)) * 0.5).astype(np.float32)).cuda()
        keep = torch.from_numpy((rng.random((steps, 2, B, m.decoder.prenet_dim)) < 0.5).astype(np.uint8)).cuda()
        ref = None
        for r in range(runs):
            out = m.decoder.inference(mem, lens, keep_masks=keep, fixed_steps=steps)
            torch.cuda.synchronize()
            cur = [o.clone() for o in out[:3]]
            if ref is None:
                ref = cur
            elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                bad += 1
                print(f"B={B} T={T}: run {r} differs from run 0 (max |d mel| = {float((ref[0] - cur[0]).abs().max()):.3e})", flush=True)
        print(f"B={B} T={T}: {runs} runs of {steps} steps, decoder form {m.decoder.persistent_state}, finite: {bool(torch.isfinite(ref[0]).all())}", flush=True)
    print("all runs bit-identical" if bad == 0 else f"{bad} runs differed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
