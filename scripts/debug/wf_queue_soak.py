"""Soak of the WaveFlow row queue's release protocol: small shapes (consumers poll while producers finish, so a result that
is flagged before it is visible WOULD be read) and the full size, many calls each, every call compared bit for bit with the
per-layer launches."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from cookietts_amd import WaveFlow, _lib, synthetic  # noqa: E402

cfg = synthetic.WAVEFLOW_CONFIGS["full"]
m = WaveFlow(**cfg)
m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=78)))
m = m.cuda().eval()


def knobs(**kv):
    for k in ("CTTS_WF_NO_ROW_QUEUE", "CTTS_F32_NO_SPLITK", "CTTS_WF_NO_REGION_SPLIT", "CTTS_WF_ROW_QUEUE_MIN"):
        os.environ.pop(k, None)
    os.environ.update(kv)
    _lib.tuning_reload()


bad = 0
# (B, F, calls, split-K body?)  The split-K body is compared with the per-layer split-K shape, which the library only takes
# up to 128 blocks of 128 x 256: sizes chosen below that.
for B, F, calls, splitk in [(1, 40, 60, False), (1, 130, 40, False), (2, 77, 40, False), (3, 200, 30, False), (5, 333, 20, False),
                            (8, 900, 12, False), (1, 40, 60, True), (1, 900, 20, True), (2, 500, 30, True), (3, 300, 30, True)]:
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F + 1, seed=B * 7 + F)).cuda()
    z = (torch.randn(B, F * 256, generator=torch.Generator().manual_seed(F)) * 0.6).cuda()
    shape = {} if splitk else {"CTTS_F32_NO_SPLITK": "1"}
    knobs(CTTS_WF_NO_ROW_QUEUE="1", CTTS_WF_NO_REGION_SPLIT="1", **shape)
    ref, _ = m.inverse(z, mel, return_CPU=False)
    assert bool(_lib.lib().ctts_last_gemm_loop() & 32) == splitk
    knobs(CTTS_WF_ROW_QUEUE_MIN="1", **shape)
    t0 = time.time()
    n_bad = 0
    for _ in range(calls):
        got, _ = m.inverse(z, mel, return_CPU=False)
        code = _lib.lib().ctts_last_gemm_loop()
        assert code & 64 and bool(code & 32) == splitk, code
        n_bad += 0 if torch.equal(got, ref) else 1
    torch.cuda.synchronize()
    print(f"B={B} F={F} body={'split-K' if splitk else '128x128'}: {calls} calls, {n_bad} differ from the per-layer launches, "
          f"{(time.time() - t0) / calls * 1e3:.1f} ms per call", flush=True)
    bad += n_bad
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
