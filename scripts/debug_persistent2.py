"""Compare the decoder workspace state after k steps: persistent vs per-launch path."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cookietts_amd import synthetic
from cookietts_amd.tacotron2 import Tacotron2

GOLDEN = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(GOLDEN, "tacotron_decoder.npz"))
hp = synthetic.tacotron_hparams()
shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
sd = synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes)
m = Tacotron2(hp); m.load_state_dict(synthetic.to_torch(sd)); m = m.cuda().eval()
mem, lens = torch.from_numpy(g["memory_in"]).cuda(), torch.from_numpy(g["lengths"]).cuda()
B, T = mem.shape[0], mem.shape[1]
NB = 1 if B <= 1 else 2 if B <= 2 else 4
al = lambda n: (n + 63) // 64 * 64
names = [("memory", NB * T * 512), ("pm", NB * T * 192), ("att_h0", NB * 1280), ("att_h1", NB * 1280), ("att_c", NB * 1280),
         ("dec_h0", NB * 768), ("dec_h1", NB * 768), ("dec_c", NB * 768), ("d2_h0", NB * 768), ("d2_h1", NB * 768),
         ("d2_c", NB * 768), ("w", NB * T), ("cum", NB * T), ("ctx", NB * 512), ("pos", NB), ("prenet", NB * 256)]
def carve(ws):
    o, out = 0, {}
    for n, sz in names:
        out[n] = ws[o:o + sz].cpu().numpy().copy(); o += al(sz)
    return out
for k in (1, 2, 3):
    st = {}
    for name, env in (("persistent", None), ("launches", "1")):
        if env: os.environ["CTTS_TACO_NO_PERSIST"] = env
        else: os.environ.pop("CTTS_TACO_NO_PERSIST", None)
        m.decoder.max_decoder_steps = 14
        mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=g["masks"], fixed_steps=None if False else k) if False else \
            m.decoder.inference(mem, lens, keep_masks=g["masks"][:14], fixed_steps=k)
        torch.cuda.synchronize()
        ws = list(m.decoder._ws.values())[0][0]
        st[name] = carve(ws)
    cur = k & 1
    keys = [f"att_h{cur}", "att_c", f"dec_h{cur}", "dec_c", f"d2_h{cur}", "d2_c", "w", "cum", "ctx", "pos", "prenet"]
    print(f"after {k} step(s): " + "  ".join(f"{n}:{np.abs(st['persistent'][n] - st['launches'][n]).max():.1e}" for n in keys))
