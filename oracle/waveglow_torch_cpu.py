"""Multi-threaded CPU restatement of WaveGlow inference (``glow.py`` topology) on torch CPU ops.

TEST INFRASTRUCTURE ONLY - this is the ``cpu_baseline`` harness BASELINE.md §4 / SURVEY.md §8(d)
describe: the same algorithm as ``oracle/waveglow_oracle.py`` (numpy), but expressed with
``F.conv_transpose1d`` / ``F.conv1d`` so that the host's cores are actually used the way the
reference's CPU path uses them (oneDNN convolutions, intra-op thread pool).  It contains no
reference code: it is written from the equations of SURVEY.md §8(a) "Verified restatement of rows
G2-G8".  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it; the product path
(``cookietts_amd``) never does.

Parity pin: ``tests/test_oracle_golden.py`` checks it against the reference's own outputs in
``tests/golden/waveglow_*.npz`` (same bound as the numpy oracle).

Reference lines each step follows (relative to /root/reference/CookieTTS/_4_mtw/waveglow):
  upsample + trim + unfold     glow.py:318-324
  WN stack                     glow.py:188-222, gate glow.py:34-41
  coupling inverse / inv 1x1   glow.py:337-340, :85-99
  early outputs / un-squeeze   glow.py:342-349
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn.functional as F


def physical_cores() -> int:
    """Physical cores this process may run on (affinity- and SMT-aware)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    phys = None
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        pass
    if not phys:
        phys = allowed
    return max(1, min(int(phys), int(allowed)))


def fold(sd):
    """numpy state dict (weight_g / weight_v pairs) -> dict of torch fp32 effective weights."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_v"):
            v = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            g = torch.from_numpy(np.ascontiguousarray(sd[k[:-2] + "_g"], dtype=np.float32))
            norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            out[k[:-2]] = v * (g.view_as(norm) / norm)
        elif not k.endswith(".weight_g"):
            out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


@torch.no_grad()
def waveglow_infer(w, cfg, mel, z_scaled):
    """w = fold(state_dict); mel [B, n_mel, F]; z_scaled [B, n_group, L] (sigma applied) -> wave [B, F*hop]."""
    mel = torch.as_tensor(mel, dtype=torch.float32)
    z = torch.as_tensor(z_scaled, dtype=torch.float32)
    G, hop = cfg["n_group"], cfg["hop_length"]
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    n_flows, every, esize = cfg["n_flows"], cfg["n_early_every"], cfg["n_early_size"]
    B, _, Fr = mel.shape
    y = F.conv_transpose1d(mel, w["upsample.weight"], w["upsample.bias"], stride=hop)[:, :, :Fr * hop]
    L = Fr * hop // G
    spect = y.reshape(B, -1, L, G).permute(0, 1, 3, 2).reshape(B, -1, L)
    n_early = sum(1 for k in range(1, n_flows) if k % every == 0)
    lo = n_early * esize
    audio = z[:, lo:, :]
    for k in reversed(range(n_flows)):
        p = f"WN.{k}"
        h = audio.shape[1] // 2
        a0, a1 = audio[:, :h], audio[:, h:]
        x = F.conv1d(a0, w[p + ".start.weight"], w[p + ".start.bias"])
        c = spect
        j = 0
        while f"{p}.cond_layers.{j}.bias" in w:
            c = F.conv1d(c, w[f"{p}.cond_layers.{j}.weight"], w[f"{p}.cond_layers.{j}.bias"])
            j += 1
        out = None
        for i in range(n_layers):
            d = 2 ** i
            u = F.conv1d(x, w[f"{p}.in_layers.{i}.weight"], w[f"{p}.in_layers.{i}.bias"], dilation=d, padding=d)
            u = u + c[:, 2 * C * i:2 * C * (i + 1)]
            act = torch.tanh(u[:, :C]) * torch.sigmoid(u[:, C:])
            r = F.conv1d(act, w[f"{p}.res_skip_layers.{i}.weight"], w[f"{p}.res_skip_layers.{i}.bias"])
            if i < n_layers - 1:
                x = x + r[:, :C]
                skip = r[:, C:]
            else:
                skip = r
            out = skip if out is None else out + skip
        e = F.conv1d(out, w[p + ".end.weight"], w[p + ".end.bias"])
        b, log_s = e[:, :h], e[:, h:]
        a1 = (a1 - b) / torch.exp(log_s)
        audio = torch.cat([a0, a1], dim=1)
        w_inv = torch.linalg.inv(w[f"convinv.{k}.conv.weight"][:, :, 0])
        audio = torch.matmul(w_inv, audio)
        if k % every == 0 and k > 0:
            lo -= esize
            audio = torch.cat([z[:, lo:lo + esize, :], audio], dim=1)
    assert lo == 0
    return audio.permute(0, 2, 1).reshape(B, -1).numpy()


def timed_baseline(sd, cfg, frames, seed, budget_s=40.0, max_runs=5, probe_frames=48):
    """Thread-count probe on a short utterance (also the warm-up: thread pool + oneDNN primitive creation), then
    best-of-N full utterances at the best count.

    oneDNN's batch-1 conv1d does not scale to every core of a large host: the probe times ``probe_frames`` frames at
    {physical cores, /2, /4, /8, 8, 4} threads and keeps the fastest, so the baseline is the best this CPU can do, not a
    thread-oversubscribed figure.  N is bounded by ``budget_s`` of wall time (at least one timed run always happens).
    Returns dict(value samples/s, cores = threads used, runs, best_s, samples, probe).
    """
    from cookietts_amd import synthetic
    phys = physical_cores()
    w = fold(sd)
    G, hop = cfg["n_group"], cfg["hop_length"]

    def inputs(fr):
        mel = synthetic.synthetic_mel(1, fr, seed=seed)
        z = synthetic.synthetic_noise(1, G, fr * hop // G, seed=seed) * np.float32(0.6)
        return mel, z

    pf = min(probe_frames, frames)
    pm, pz = inputs(pf)
    torch.set_num_threads(phys)
    waveglow_infer(w, cfg, *inputs(min(8, frames)))            # first-touch warm-up, untimed
    probe = {}
    for n in sorted({phys, max(phys // 2, 1), max(phys // 4, 1), max(phys // 8, 1), min(8, phys), min(4, phys)}, reverse=True):
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        waveglow_infer(w, cfg, pm, pz)
        probe[n] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    mel, z = inputs(frames)
    best, runs, t_start = None, 0, time.perf_counter()
    while runs < max_runs and (runs == 0 or time.perf_counter() - t_start + (best or 0) < budget_s):
        t0 = time.perf_counter()
        wave = waveglow_infer(w, cfg, mel, z)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        runs += 1
    return {"value": wave.size / best, "cores": cores, "threads_set": torch.get_num_threads(), "physical": phys,
            "runs": runs, "best_s": best, "samples": int(wave.size), "frames": frames,
            "probe": {str(k): round(v, 3) for k, v in probe.items()}}


# ---- aggregate figure: several single-utterance instances side by side -------------------------------------------
# One utterance on `threads` cores leaves the rest of a large host idle while the GPU row runs a batch of 8.  The
# aggregate figure runs `instances` fresh child processes at once (each: its own weights, warm-up, then ONE timed
# utterance, all started together on a file barrier) and counts all their samples over the span from the first start
# to the last end.  The children never touch a GPU (HIP_VISIBLE_DEVICES is emptied, nothing of the HIP library loads).
def _child_main(argv):
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--frames", type=int, required=True)
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--sync-dir", required=True)
    ap.add_argument("--index", type=int, required=True)
    a = ap.parse_args(argv)
    from cookietts_amd import synthetic
    cfg = synthetic.WAVEGLOW_CONFIGS[a.config]
    torch.set_num_threads(a.threads)
    w = fold(synthetic.waveglow_state_dict(cfg, seed=a.seed))
    G, hop = cfg["n_group"], cfg["hop_length"]

    def inputs(fr, sd_):
        return (synthetic.synthetic_mel(1, fr, seed=sd_),
                synthetic.synthetic_noise(1, G, fr * hop // G, seed=sd_) * np.float32(0.6))
    waveglow_infer(w, cfg, *inputs(min(8, a.frames), a.seed))                 # warm-up, untimed
    mel, z = inputs(a.frames, a.seed + a.index)
    open(os.path.join(a.sync_dir, f"ready_{a.index}"), "w").close()
    go = os.path.join(a.sync_dir, "go")
    t_wait = time.time()
    while not os.path.exists(go):
        if time.time() - t_wait > 600:
            raise SystemExit("no go file")
        time.sleep(0.01)
    t0 = time.time()
    wave = waveglow_infer(w, cfg, mel, z)
    t1 = time.time()
    print(json.dumps({"index": a.index, "start": t0, "end": t1, "samples": int(wave.size), "finite": bool(np.isfinite(wave).all())}),
          flush=True)


def timed_aggregate(config_key, frames, seed, threads, instances, timeout_s=900.0):
    """`instances` concurrent child processes x `threads` threads, one `frames`-frame utterance each.
    Returns dict(value = all samples / (last end - first start), instances, threads, span_s, per_instance_s)."""
    import json
    import subprocess
    import sys
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    env["PYTHONPATH"] = repo + os.pathsep + env.get("PYTHONPATH", "")
    with tempfile.TemporaryDirectory(prefix="ctts_cpu_") as d:
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.waveglow_torch_cpu", "--config", config_key, "--frames", str(frames),
                                   "--threads", str(threads), "--seed", str(seed), "--sync-dir", d, "--index", str(i)],
                                  cwd=repo, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for i in range(instances)]
        try:
            t0 = time.time()
            while not all(os.path.exists(os.path.join(d, f"ready_{i}")) for i in range(instances)):
                for p in procs:
                    if p.poll() is not None:
                        raise RuntimeError(f"CPU-baseline child exited early ({p.returncode}): {p.stderr.read()[-2000:]}")
                if time.time() - t0 > timeout_s:
                    raise RuntimeError("CPU-baseline children did not get ready in time")
                time.sleep(0.05)
            open(os.path.join(d, "go"), "w").close()
            outs = []
            for p in procs:
                out, err = p.communicate(timeout=timeout_s)
                if p.returncode != 0:
                    raise RuntimeError(f"CPU-baseline child failed ({p.returncode}): {err[-2000:]}")
                outs.append(json.loads(out.strip().splitlines()[-1]))
        finally:
            for p in procs:           # exact children only
                if p.poll() is None:
                    p.kill()
    assert all(o["finite"] for o in outs)
    span = max(o["end"] for o in outs) - min(o["start"] for o in outs)
    total = sum(o["samples"] for o in outs)
    return {"value": total / span, "instances": instances, "threads": threads, "span_s": span, "samples": total,
            "per_instance_s": [round(o["end"] - o["start"], 2) for o in sorted(outs, key=lambda o: o["index"])]}


if __name__ == "__main__":
    import sys
    _child_main(sys.argv[1:])
