"""bench.py at a list of batch sizes (no rows, no CPU baseline): python scripts/debug/bench_batches.py f16 1 2 4
(FRAMES=<n> in the environment: mel frames per utterance, default 900)"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
dtype = sys.argv[1]
for b in sys.argv[2:]:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", b, "--frames", os.environ.get("FRAMES", "900"), "--no-rows", "--cpu-frames", "0", "--steps", "10", "--warmup", "3"]
    if dtype != "f32":
        cmd += ["--dtype", dtype]
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    print(f"{dtype} batch {b} x {os.environ.get('FRAMES', '900')} frames: {d['ms_per_step']:.2f} ms/step = {d['rtf']:.0f}x real time, dominant kernel at {d['roofline']['frac']:.3f} of its roofline", flush=True)
