"""Phase timeline of the persistent decoder from in-kernel s_memrealtime stamps (config 5 shapes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cookietts_amd import synthetic, _lib
from cookietts_amd.tacotron2 import Tacotron2
hp = synthetic.tacotron_hparams()
m = Tacotron2(hp); m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234))); m = m.cuda().eval()
B, T, steps = 4, 200, 64
rng = np.random.default_rng(1)
mem = torch.from_numpy((rng.standard_normal((B, T, 1313)) * 0.5).astype(np.float32)).cuda()
lens = torch.tensor([200, 195, 150, 100]).cuda()
m.decoder.inference(mem, lens, fixed_steps=steps)            # warm
stamps = torch.zeros(256 * 64 * 16, dtype=torch.int64, device="cuda")
_lib.lib().ctts_taco_decoder_persistent_debug(_lib.ptr(stamps))
m.decoder.inference(mem, lens, fixed_steps=steps)
torch.cuda.synchronize()
_lib.lib().ctts_taco_decoder_persistent_debug(None)
s = stamps.cpu().numpy().reshape(256, 64, 16).astype(np.float64) / 100.0      # us
names = ["A:fresh att+publish", "wait att_h", "B:q+early dec(att)", "wait ctx", "C:fresh dec+early", "wait dec_h",
         "D:fresh d2+early", "wait d2_h", "E:proj+early d2hh", "wait h1", "F:W2+early att hh", "wait p"]
for wg in (0, 19, 100, 251):
    d = np.diff(s[wg, 8:56, :13], axis=1)              # steady-state steps
    step_us = (s[wg, 9:56, 0] - s[wg, 8:55, 0]).mean()
    print(f"wg {wg}: step {step_us:.1f} us | " + " | ".join(f"{n} {v:.1f}" for n, v in zip(names, d.mean(axis=0))))
for wg in (252, 255):
    d = np.diff(s[wg, 8:56, :7], axis=1).mean(axis=0)
    print(f"attention wg {wg}: " + " | ".join(f"{n} {v:.1f}" for n, v in zip(
        ["pre (bursts + loc conv)", "wait q", "energies", "softmax", "ctx", "publish + w/cum"], d)))
