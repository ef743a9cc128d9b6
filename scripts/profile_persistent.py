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
SLOTS = 24
stamps = torch.zeros(256 * 64 * SLOTS, dtype=torch.int64, device="cuda")
_lib.lib().ctts_taco_decoder_persistent_debug(_lib.ptr(stamps))
m.decoder.inference(mem, lens, fixed_steps=steps)
torch.cuda.synchronize()
_lib.lib().ctts_taco_decoder_persistent_debug(None)
s = stamps.cpu().numpy().reshape(256, 64, SLOTS).astype(np.float64) / 100.0      # us
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

for wg in (252,):
    e = s[wg, 8:56, :]
    print(f"attention wg {wg} energies split: q gathered -> tanh loop done {np.mean(e[:, 18] - e[:, 2]):.2f} | wave reductions "
          f"{np.mean(e[:, 19] - e[:, 18]):.2f} | store + barrier {np.mean(e[:, 3] - e[:, 19]):.2f} us")

# Cross-workgroup view of every exchange (s_memrealtime is one chip-wide 100 MHz counter): when did the LAST publisher
# publish, how far apart were the publishers, and how long after the last publish had the FIRST / LAST consumer its copy.
# (publish stamps: 1 att_h, 17 q, attention 5 ctx, 13 dec_h, 14 d2_h, 15 h1, 16 p; gather-done stamps: 2, attn 2, 4, 6, 8, 10, 12)
L, Aw = slice(0, 252), slice(252, 256)
steps_ss = slice(8, 56)
def col(wgs, k):
    x = s[wgs, steps_ss, k]
    return np.where(x > 0, x, np.nan)
rows = [("att_h (5120 granules)", col(L, 1), col(L, 2)), ("q (768, to the 4 attention wgs)", col(slice(0, 192), 17), col(Aw, 2)),
        ("ctx (2048)", col(Aw, 5), col(L, 4)), ("dec_h (3072)", col(L, 13), col(L, 6)), ("d2_h (3072)", col(L, 14), col(L, 8)),
        ("h1 (1024)", col(L, 15), col(L, 10)), ("p (1024)", col(L, 16), col(L, 12))]
print("exchange: publisher spread (last - first publish) | first consumer done after last publish | last consumer done after last publish  [us, mean over steps]")
for name, pub, done in rows:
    last_pub, first_pub = np.nanmax(pub, axis=0), np.nanmin(pub, axis=0)
    print(f"  {name:34s} {np.nanmean(last_pub - first_pub):5.2f} | {np.nanmean(np.nanmin(done, axis=0) - last_pub):5.2f} | {np.nanmean(np.nanmax(done, axis=0) - last_pub):5.2f}")
print("compute between a gather and the next publish (mean over workgroups and steps): "
      f"A {np.nanmean(col(L, 1) - col(L, 0)):.2f} | q {np.nanmean(col(slice(0, 192), 17) - col(slice(0, 192), 2)):.2f} | "
      f"attention post {np.nanmean(col(Aw, 5) - col(Aw, 2)):.2f} | C {np.nanmean(col(L, 13) - col(L, 4)):.2f} | "
      f"D {np.nanmean(col(L, 14) - col(L, 6)):.2f} | h1 {np.nanmean(col(L, 15) - col(L, 8)):.2f} | p {np.nanmean(col(L, 16) - col(L, 10)):.2f}")

# per-wave att_h publish times relative to wave 0's (slots 20 + wave), and the latest wave of the whole chip relative to the last wave-0 publish
pw = np.stack([col(L, 20 + w) for w in range(4)])                      # [wave][wg][step]
print("att_h publish of wave w minus wave 0 of the same workgroup (mean / max over workgroups and steps): " +
      " | ".join(f"w{w} {np.nanmean(pw[w] - pw[0]):.2f} / {np.nanmax(pw[w] - pw[0]):.2f}" for w in range(1, 4)))
last_any = np.nanmax(pw, axis=(0, 1))
print(f"last att_h publish of ANY wave after the last wave-0 publish: {np.nanmean(last_any - np.nanmax(col(L, 1), axis=0)):.2f} us; "
      f"first consumer done after the last publish of any wave: {np.nanmean(np.nanmin(col(L, 2), axis=0) - last_any):.2f} us")
for name, sl_ in (("class 0 (wg 0-19)", slice(0, 20)), ("class 1 (20-51)", slice(20, 52)), ("class 2 (52-95)", slice(52, 96)), ("class 3 (96-251)", slice(96, 252))):
    print(f"  {name}: " + " | ".join(f"w{w} mean {np.nanmean((pw[w] - pw[0])[sl_]):.2f} max {np.nanmax((pw[w] - pw[0])[sl_]):.2f}" for w in range(1, 4)))
late = pw[3] - pw[0]
print("  wave 3 lateness by step (mean over workgroups), steps 8..23: " + " ".join(f"{v:.2f}" for v in np.nanmean(late, axis=0)[:16]))
worst = np.unravel_index(np.nanargmax(late), late.shape)
print(f"  worst: wg {worst[0]}, steady-state step index {worst[1]}; workgroups with wave-3 lateness > 1 us anywhere: {np.where(np.nanmax(late, axis=1) > 1.0)[0].tolist()[:40]}")
