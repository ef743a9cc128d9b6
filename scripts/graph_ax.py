"""Is the batch-1 ax WaveGlow call launch-bound?  Same call eagerly and replayed from a captured HIP graph."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cookietts_amd import synthetic
from cookietts_amd.waveglow_ax import WaveGlow
cfg = synthetic.WAVEGLOW_AX_CONFIGS["notebook"]
m = WaveGlow(**cfg); m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=1234))); m = m.cuda().eval()
B, F = 1, 468
mel = torch.from_numpy(synthetic.synthetic_mel(B, F, cfg["n_mel_channels"])).cuda()
ids = torch.zeros(B, dtype=torch.int64).cuda()
call = lambda: m.infer(mel, speaker_ids=ids, sigma=1.0, return_CPU=False)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager ms/call", round(timed(call), 2))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): call()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        out = call()
    print("graph ms/call", round(timed(g.replay), 2))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
