"""Timing of the WaveFlow row queue at config 4 under the CTTS_WF_QUEUE_DEBUG bits (children: one setting per process)."""
import os
import subprocess
import sys
import time

CHILD = r'''
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from cookietts_amd import synthetic, WaveFlow, _lib
cfg = synthetic.WAVEFLOW_CONFIGS["full"]
m = WaveFlow(**cfg); m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=78))); m = m.cuda().eval()
B, F = int(sys.argv[1]), int(sys.argv[2])
mel = torch.from_numpy(synthetic.synthetic_mel(B, F + 1, seed=3)).cuda()
z = (torch.randn(B, F * 256, generator=torch.Generator().manual_seed(3)) * 0.6).cuda()
a, _ = m.inverse(z, mel, return_CPU=False); torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.time(); a2, _ = m.inverse(z, mel, return_CPU=False); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
print("ms", " ".join("%.1f" % t for t in ts), "loop", _lib.lib().ctts_last_gemm_loop(), "deterministic", bool(torch.equal(a, a2)), flush=True)
'''
cases = [tuple(int(x) for x in c.split(":")) for c in sys.argv[1].split(",")]
for B, F, dbg in cases:
    # debug >= 0: queue forced, CTTS_WF_QUEUE_DEBUG = debug (+ 1024: CTTS_F32_NO_SPLITK, i.e. the 128 x 128 body at every size);
    # -1: queue off; -2: library default
    env = dict(os.environ)
    if dbg >= 0:
        env["CTTS_WF_ROW_QUEUE_MIN"] = "1"
        env["CTTS_WF_QUEUE_DEBUG"] = str(dbg & 1023)
        if dbg & 1024:
            env["CTTS_F32_NO_SPLITK"] = "1"
    elif dbg == -1:
        env["CTTS_WF_NO_ROW_QUEUE"] = "1"
    try:
        p = subprocess.run([sys.executable, "-c", CHILD, str(B), str(F)], env=env, capture_output=True, text=True, timeout=100)
        out = p.stdout.strip() + (" rc=%d " % p.returncode) + (p.stderr[-300:] if p.returncode else "")
    except subprocess.TimeoutExpired as e:
        out = "TIMEOUT"
    print(f"B={B} F={F} debug={dbg}: {out}", flush=True)
    if out == "TIMEOUT":
        break
