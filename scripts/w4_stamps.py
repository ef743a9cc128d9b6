"""In-kernel clocks of the bf16 four-wave conv-GEMM (block 1000 of the last in-layer launch): CTTS_BF16_W4_DEBUG=3."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTTS_BF16_W4"] = "1"
os.environ["CTTS_BF16_W4_DEBUG"] = sys.argv[1] if len(sys.argv) > 1 else "4"
from cookietts_amd import synthetic, _lib
from cookietts_amd.waveglow import WaveGlow
cfg = synthetic.WAVEGLOW_CONFIGS["full"]
m = WaveGlow(**cfg)
m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=1)))
m = m.cuda().eval(); m.set_compute_dtype(torch.bfloat16)
B, F = 8, 900
mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
for _ in range(2):
    m.infer_from_noise(mel, z); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib = ctypes.CDLL(_lib.lib()._name)
print("rc", lib.ctts_debug_w4_stamps(out))
loop, ticks, epi, pro, nch = [out[i] for i in range(5)]
print(f"nch {nch}: loop {loop} cycles = {loop / max(nch,1):.0f} per chunk; {ticks / 100:.2f} us -> clock {loop / (ticks / 100):.0f} MHz; "
      f"epilogue {epi} cycles; prologue {pro} cycles")
