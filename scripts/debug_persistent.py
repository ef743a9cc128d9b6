"""Step-by-step comparison of the persistent decoder against the per-launch decoder and the oracle."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cookietts_amd import synthetic
from cookietts_amd.tacotron2 import Tacotron2

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(GOLDEN, "tacotron_decoder.npz"))
hp = synthetic.tacotron_hparams()
shapes = json.load(open(os.path.join(GOLDEN, "tacotron_state_shapes.json")))
sd = synthetic.tacotron_state_dict(hp, seed=int(g["seed"]), shapes=shapes)
m = Tacotron2(hp); m.load_state_dict(synthetic.to_torch(sd)); m = m.cuda().eval()
mem, lens = torch.from_numpy(g["memory_in"]).cuda(), torch.from_numpy(g["lengths"]).cuda()
n = g["masks"].shape[0]
outs = {}
for name, env in (("persistent", None), ("launches", "1")):
    if env: os.environ["CTTS_TACO_NO_PERSIST"] = env
    else: os.environ.pop("CTTS_TACO_NO_PERSIST", None)
    if name == "persistent" and len(sys.argv) > 1: os.environ["CTTS_TACO_CHUNK"] = sys.argv[1]
    else: os.environ.pop("CTTS_TACO_CHUNK", None)
    mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=g["masks"], fixed_steps=n)
    outs[name] = (mel.cpu().numpy(), gate.cpu().numpy(), align.cpu().numpy())
ref = (g["mel"], g["gate_sigmoid"], g["alignments"])
for s in range(min(n, 6)):
    row = []
    for name in ("persistent", "launches"):
        o = outs[name]
        row.append(f"{name}: mel {np.abs(o[0][:, :, s] - ref[0][:, :, s]).max():.2e} gate {np.abs(o[1][:, s] - ref[1][:, s]).max():.2e} "
                   f"align {np.abs(o[2][:, s] - ref[2][:, s]).max():.2e}")
    print(f"step {s}: " + " | ".join(row))
d = np.abs(outs["persistent"][0] - outs["launches"][0])
print("mel |persistent - launches| per channel at step 1, b=0:", np.array2string(d[0, :, 1], precision=1, max_line_width=250))
print("gate diff per step b=0:", np.array2string(np.abs(outs["persistent"][1] - outs["launches"][1])[0], precision=1))
