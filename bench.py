#!/usr/bin/env python3
"""Headline benchmark: WaveGlow inference throughput (audio samples/s at 22.05 kHz).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path - ``WaveGlow.infer`` (glow.py:314-350 semantics) of
BASELINE.json config 2: full WaveGlow (12 flows, 512 WN channels, 8 groups), fp32, on a batch
of 8 synthetic 80x900 mels already resident in HBM, random-init weights from the deterministic
recipe in cookietts_amd/synthetic.py.  For N > 1 (launched by torch.distributed.run, one rank
per GPU over RCCL) every rank runs its own batch of 8 (utterance-batch sharding, weak scaling,
no data-path collective); the timed region is bracketed by barrier + synchronize and the MAX
over ranks is used.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     - the dominant kernel (WN in-layer conv-GEMM: dilated conv + cond + gate),
                 algorithmic FLOPs per launch / mean launch time measured with HIP events
                 inside the library on the launch stream, vs the fp32 MFMA peak.
  cpu_baseline - the numpy CPU oracle (a port, test infrastructure) timed on this box's host
                 cores on a bounded sample of the same model (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "audio samples/sec (22.05kHz) WaveGlow infer, 80×900 mel, 1/2/4/8 GPU; real-time factor"
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
# HBM bytes per in-layer launch from the committed PMC passes (profiles/r1_17_pmc_traffic.json:
# 2 x FETCH_SIZE (gfx950 half-count correction, calibrated on flow_tail) + WRITE_SIZE), config 2 shapes only
TRAFFIC = {}
try:
    with open(os.path.join(REPO, "profiles", "r1_17_pmc_traffic.json")) as _f:
        TRAFFIC = {k: v.get("hbm_bytes_per_launch") for k, v in json.load(_f).items()}
except Exception:
    pass


def cpu_baseline(cfg, sd, frames, seed):
    """Oracle (numpy) on host cores: one utterance of `frames` mel frames through the full model."""
    from oracle import waveglow_oracle as wo
    from cookietts_amd import synthetic
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    folded = wo.fold_state_dict(sd)
    mel = synthetic.synthetic_mel(1, frames, seed=seed)
    z = synthetic.synthetic_noise(1, cfg["n_group"], frames * cfg["hop_length"] // cfg["n_group"], seed=seed)
    z = z * np.float32(0.6)
    t0 = time.perf_counter()
    wave = wo.waveglow_infer(folded, cfg, mel, z)
    dt = time.perf_counter() - t0
    return {"value": wave.size / dt, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle, same 12x512 model, 1 utterance x {frames} mel frames "
                      f"({wave.size} samples) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU (config 2: 8)")
    ap.add_argument("--frames", type=int, default=900)
    ap.add_argument("--config", default="full", help="key of cookietts_amd.synthetic.WAVEGLOW_CONFIGS")
    ap.add_argument("--cpu-frames", type=int, default=128, help="mel frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = BASELINE config 2 (default, the headline); bf16 = config 3 (use --batch 32)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cookietts_amd import WaveGlow, _lib, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    seed = 1234
    cfg = synthetic.WAVEGLOW_CONFIGS[args.config]
    sd = synthetic.waveglow_state_dict(cfg, seed=seed)          # every rank: full replica, same weights
    model = WaveGlow(**cfg)
    model.load_state_dict(synthetic.to_torch(sd))
    model = model.to(device).eval()
    if args.dtype == "bf16":
        import torch as _t
        model.set_compute_dtype(_t.bfloat16)
    B, F = args.batch, args.frames
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=seed + rank)).to(device)   # resident in HBM
    T = F * cfg["hop_length"]

    def step():
        return model.infer(mel, sigma=0.6)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    lib = _lib.lib()
    for _ in range(args.warmup):
        out = step()
    fence()
    timing = not args.no_kernel_timing
    if timing:
        lib.ctts_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.ctts_profile_enable(0)
    assert out.shape == (B, T) and bool(torch.isfinite(out).all())

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        samples = world * B * T * args.steps
        value = samples / elapsed
        wn = cfg["WN_config"]
        C, n_layers = wn["n_channels"], wn["n_layers"]
        L = T // cfg["n_group"]
        roofline = None
        if timing:
            n = ctypes.c_int64()
            ms = ctypes.c_double()
            _lib.check(lib.ctts_profile_collect(_lib.PROF_WN_IN, ctypes.byref(n), ctypes.byref(ms)), "profile")
            # algorithmic MACs per time step of ONE in-layer launch: dilated conv
            # C*2C*3 plus this layer's slice of the conditioning projection 256*2C
            mac = 3 * C * 2 * C + 256 * 2 * C
            flop_per_launch = 2.0 * mac * B * L
            mean_s = ms.value / max(n.value, 1) * 1e-3
            achieved = flop_per_launch / mean_s / 1e12
            peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
            kname = "conv_gemm_f32_kernel<GATE>" if args.dtype == "f32" else "conv_gemm_bf16_pp_kernel<GATE>"
            roofline = {"kernel": kname + " (WN in-layer: dilated conv + cond + tanh*sigmoid)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": TRAFFIC.get(args.dtype),
                        "launches": int(n.value), "mean_launch_ms": round(mean_s * 1e3, 4),
                        "flop_per_launch": flop_per_launch}
            if args.dtype == "bf16":
                # In bf16 the in-layer GEMM stays MFMA-bound (1430 FLOP/B vs a ridge of ~312); the memory-bound WN
                # kernel is the res/skip GEMM (K = C): 2*C*2C FLOP against act read + x and skip-sum read-modify-write
                # = 5*C*2 B per time step (205 FLOP/B).  Report it against HBM.
                n2 = ctypes.c_int64()
                ms2 = ctypes.c_double()
                _lib.check(lib.ctts_profile_collect(_lib.PROF_WN_RS, ctypes.byref(n2), ctypes.byref(ms2)), "profile")
                mean2 = ms2.value / max(n2.value, 1) * 1e-3
                bytes2 = float(5 * C * 2) * B * L
                roofline["res_skip_hbm"] = {
                    "kernel": "conv_gemm_bf16_pp_kernel<SPLIT> (WN res/skip 1x1 + residual/skip accumulate)", "bound": "hbm",
                    "achieved": round(bytes2 / mean2 / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(bytes2 / mean2 / 8e12, 4), "launches": int(n2.value),
                    "mean_launch_ms": round(mean2 * 1e3, 4), "bytes_per_launch": bytes2}
                # whole step against HBM with SURVEY 8d's per-layer-kernel byte count (7*C*2 B per step per layer)
                step_bytes = 7.0 * C * 2 * n_layers * cfg["n_flows"] * B * L
                roofline["step_hbm_algorithmic"] = {"bytes_per_step": step_bytes, "achieved": round(step_bytes * args.steps / elapsed / 1e9, 1),
                                                    "peak": 8000.0, "unit": "GB/s",
                                                    "frac": round(step_bytes * args.steps / elapsed / 8e12, 4)}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu = cpu_baseline(cfg, sd, args.cpu_frames, seed)
        line = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rtf": value / 22050.0,
            "config": {"workload": f"WaveGlow {args.config} ({cfg['n_flows']} flows, {C} WN ch, "
                                   f"{cfg['n_group']} groups, {n_layers} layers) {'fp32' if args.dtype == 'f32' else 'bf16-MFMA'} infer, batch {B} x (80x{F}) "
                                   f"mel per GPU, sigma 0.6, random-init weights",
                       "batch_per_gpu": B, "frames": F, "samples_per_step": world * B * T,
                       "parallelism": f"utterance-batch shard x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
