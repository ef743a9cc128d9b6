"""Half-mode WaveGlow: a batch as ONE call against its two halves on two streams (two model copies = two workspaces).
Does running the halves side by side fill the idle CUs of the last round of each launch?   python scripts/debug/two_stream_halves.py [dtype] [B ...]"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cookietts_amd import WaveGlow, synthetic  # noqa: E402

dtype = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "f16"]
batches = [int(x) for x in sys.argv[1:] if x.isdigit()] or [2, 3, 4, 6, 8, 12, 16]
cfg = synthetic.WAVEGLOW_CONFIGS["full"]
m = WaveGlow(**cfg)
m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=3)))
m = m.cuda().eval()
m.set_compute_dtype(dtype)
m2 = copy.deepcopy(m)
m2.set_compute_dtype(dtype)
F = 900
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


for B in batches:
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=5)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=5)).cuda() * 0.6
    h = (B + 1) // 2
    ma, za, mb, zb = mel[:h].contiguous(), z[:h].contiguous(), mel[h:].contiguous(), z[h:].contiguous()

    def one():
        return m.infer_from_noise(mel, z)

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = m.infer_from_noise(ma, za)
        with torch.cuda.stream(s2):
            b = m2.infer_from_noise(mb, zb)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return torch.cat([a, b])

    reps = max(3, 40 // B)
    o1, t1 = timed(one, reps)
    o2, t2 = timed(two, reps)
    print(f"B={B:3d}: one call {t1:8.2f} ms   two halves on two streams {t2:8.2f} ms   identical={bool(torch.equal(o1, o2))}", flush=True)
