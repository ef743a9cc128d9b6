"""upsample + squeeze (glow.py:318-324) on the MFMA kernel vs the VALU kernel: max difference and time per launch.
   python scripts/debug/upsample_ab.py [config] [B ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cookietts_amd import WaveGlow, _lib, synthetic  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "full"
batches = [int(x) for x in sys.argv[1:] if x.isdigit()] or [1, 8, 32]
cfg = synthetic.WAVEGLOW_CONFIGS[cfg_name]
m = WaveGlow(**cfg)
m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=3)))
m = m.cuda().eval()
blob, _ = m._ensure_packed(torch.device("cuda", 0))
lib = _lib.lib()
c = m.c_config()


def run(B, F, no_mfma, reps):
    if no_mfma:
        os.environ["CTTS_UP_NO_MFMA"] = "1"
    else:
        os.environ.pop("CTTS_UP_NO_MFMA", None)
    lib.ctts_tuning_reload()
    geo = _lib.WaveGlowGeometry()
    _lib.check(lib.ctts_waveglow_geometry_for(C.byref(c), F, C.byref(geo)), "geometry")
    spect = torch.zeros(B, cfg["n_mel_channels"] * cfg["n_group"], geo.ld, device="cuda")
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=4)).cuda()
    call = lambda: _lib.check(lib.ctts_upsample_squeeze_f32(C.byref(c), _lib.ptr(blob), _lib.ptr(mel), _lib.ptr(spect), B, F, None), "up")
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return spect, e0.elapsed_time(e1) / reps * 1e3


for B in batches:
    for F in (19, 333, 900):
        a, ta = run(B, F, False, 20)
        b, tb = run(B, F, True, 20)
        d = float((a - b).abs().max())
        s = float(b.abs().max())
        gf = 2.0 * cfg["n_mel_channels"] ** 2 * cfg["win_length"] * F * B / 1e9
        print(f"B={B:3d} F={F:4d}: mfma {ta:8.1f} us ({gf / ta * 1e-3 * 1e3:6.1f} TFLOP/s)   valu {tb:8.1f} us   max |diff| {d:.3e} (max |spect| {s:.2f})", flush=True)
